set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r02_hot_diag2; mkdir -p $out
export F2Q_TRACE=1
timeout -k 10 200 python bench.py --workload cfg5b_50M_anchor_ec --steps 5 --warmup 2 --no-pmc --no-cpu-baseline --no-extras > $out/a.json 2> $out/a.err || { grep -v amdgpu.ids $out/a.err | tail -12; exit 1; }
python -c "import json; d=json.load(open('$out/a.json')); print('hot', round(d['value']), 'Mreads/s kernel_ms', round(d['roofline']['kernel_ms'],4), 'frac', round(d['roofline']['frac'],3))"
