set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r02_hot; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "ec_ or anchor or golden or fuzz" > $out/pytest.txt 2>&1 || { tail -40 $out/pytest.txt; exit 1; }
tail -1 $out/pytest.txt
for v in hot nohot; do
  if [ $v = nohot ]; then export F2Q_NO_HOT=1; else unset F2Q_NO_HOT; fi
  F2Q_TRACE=1 timeout -k 10 300 python bench.py --workload cfg5b_50M_anchor_ec --steps 5 --warmup 2 --no-pmc --no-cpu-baseline --no-extras > $out/bench_$v.json 2> $out/bench_$v.err || { grep -v amdgpu.ids $out/bench_$v.err | tail; exit 1; }
  python -c "import json; d=json.load(open('$out/bench_$v.json')); print('$v', round(d['value']), 'Mreads/s kernel_ms', round(d['roofline']['kernel_ms'],4), 'frac', round(d['roofline']['frac'],3))"
done
unset F2Q_NO_HOT
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$out/prof -o hot -- python3 $GRAFT_REPO_ROOT/bench.py --workload cfg5b_50M_anchor_ec --steps 3 --warmup 1 --no-pmc --no-cpu-baseline --no-extras > $GRAFT_REPO_ROOT/$out/prof.log 2>&1
echo profiled
