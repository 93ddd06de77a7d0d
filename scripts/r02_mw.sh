set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r02_mw; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "multi_window or two_window or golden" > $out/pytest.txt 2>&1 || { tail -30 $out/pytest.txt; exit 1; }
tail -1 $out/pytest.txt
for v in packed general; do
  if [ $v = general ]; then export F2Q_FORCE_GENERAL=1; steps=2; else unset F2Q_FORCE_GENERAL; steps=10; fi
  timeout -k 10 300 python bench.py --workload cfg3_2win_50M_10k_m1 --steps $steps --no-pmc --no-cpu-baseline --no-extras > $out/bench_$v.json 2> $out/bench_$v.err
  python -c "import json; d=json.load(open('$out/bench_$v.json')); print('$v', round(d['value']), 'Mreads/s kernel_ms', round(d['roofline']['kernel_ms'],4), 'frac', round(d['roofline']['frac'],3), 'general reads', d['config']['general_path_reads_per_gpu'])"
done
