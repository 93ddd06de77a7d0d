set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r02_pair; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "config4 or device_synth or golden or range_histogram or fuzz_kernels" > $out/pytest.txt 2>&1 || { tail -40 $out/pytest.txt; exit 1; }
tail -1 $out/pytest.txt
for w in cfg4_50M_100k_m1 cfg4_50M_100k_m1; do
  timeout -k 10 300 python bench.py --workload $w --steps 10 --warmup 2 --no-pmc --no-cpu-baseline --no-extras > $out/x.json 2> $out/x.err
  python -c "import json; d=json.load(open('$out/x.json')); print('$w', 'kernel_ms', round(d['roofline']['kernel_ms'],4), 'frac', round(d['roofline']['frac'],3))"
done
