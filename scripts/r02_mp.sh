set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r02_mp; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "multi_pair or golden or fuzz or anchor" > $out/pytest.txt 2>&1 || { tail -40 $out/pytest.txt; exit 1; }
tail -1 $out/pytest.txt
for v in packed general; do
  if [ $v = general ]; then export F2Q_FORCE_GENERAL=1; steps=2; else unset F2Q_FORCE_GENERAL; steps=6; fi
  timeout -k 10 400 python bench.py --workload cfg5c_2pair_50M_10k_m1 --steps $steps --warmup 1 --no-pmc --no-cpu-baseline --no-extras > $out/bench_$v.json 2> $out/bench_$v.err || { grep -v amdgpu.ids $out/bench_$v.err | tail -5; exit 1; }
  python -c "import json; d=json.load(open('$out/bench_$v.json')); print('$v', round(d['value']), 'Mreads/s kernel_ms', round(d['roofline']['kernel_ms'],3), 'frac', round(d['roofline']['frac'],3), d['verify']['stats'])"
done
unset F2Q_FORCE_GENERAL
export F2Q_FORCE_GENERAL=1
for w in cfg3_2win_50M_10k_m1 cfg3_50M_10k_m1; do
  timeout -k 10 300 python bench.py --workload $w --steps 3 --warmup 1 --no-pmc --no-cpu-baseline --no-extras > $out/bench_general_$w.json 2> $out/bench_general_$w.err
  python -c "import json; d=json.load(open('$out/bench_general_$w.json')); print('$w general', round(d['value']), 'Mreads/s ms/step', round(d['ms_per_step'],2))"
done
