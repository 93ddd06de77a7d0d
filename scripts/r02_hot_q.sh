set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r02_hot_q; mkdir -p $out
timeout -k 10 300 python bench.py --workload cfg5b_50M_anchor_ec --steps 6 --warmup 2 --no-pmc --no-cpu-baseline --no-extras > $out/bench.json 2> $out/bench.err || { grep -v amdgpu.ids $out/bench.err | tail; exit 1; }
python -c "import json; d=json.load(open('$out/bench.json')); print('hot', round(d['value']), 'Mreads/s kernel_ms', round(d['roofline']['kernel_ms'],4), 'frac', round(d['roofline']['frac'],3))"
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$out/prof -o hot -- python3 $GRAFT_REPO_ROOT/bench.py --workload cfg5b_50M_anchor_ec --steps 3 --warmup 1 --no-pmc --no-cpu-baseline --no-extras > $GRAFT_REPO_ROOT/$out/prof.log 2>&1
echo profiled
