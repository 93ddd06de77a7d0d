# round 2: the library-in-LDS kernel -- parity tests, then A/B against the pigeonhole kernel
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02_lt
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "lds or golden or device_synth or config3 or invariants or geometry" > gpurun_out/r02_lt/pytest.txt 2>&1 || { tail -30 gpurun_out/r02_lt/pytest.txt; exit 1; }
tail -3 gpurun_out/r02_lt/pytest.txt
for v in lt nolt; do
  if [ $v = nolt ]; then export F2Q_NO_LT=1; else unset F2Q_NO_LT; fi
  for wl in cfg3_50M_10k_m1 cfg2_10M_1k_m0; do
    timeout -k 10 200 python bench.py --workload $wl --no-pmc --no-cpu-baseline --no-extras > gpurun_out/r02_lt/bench_${v}_$wl.json 2> gpurun_out/r02_lt/bench_${v}_$wl.err
    python -c "import json,sys; d=json.load(open('gpurun_out/r02_lt/bench_${v}_$wl.json')); print('$v $wl', round(d['value']), 'Mreads/s kernel_ms', round(d['roofline']['kernel_ms'],4), 'frac', round(d['roofline']['frac'],3))"
  done
done
