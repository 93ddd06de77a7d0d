set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r02_gk; mkdir -p $out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $out/pytest.txt 2>&1 || { tail -30 $out/pytest.txt; exit 1; }
tail -1 $out/pytest.txt
export F2Q_FORCE_GENERAL=1
for w in cfg3_2win_50M_10k_m1 cfg3_50M_10k_m1; do
  timeout -k 10 300 python bench.py --workload $w --steps 3 --warmup 1 --no-pmc --no-cpu-baseline --no-extras > $out/bench_general_$w.json 2> $out/bench_general_$w.err
  python -c "import json; d=json.load(open('$out/bench_general_$w.json')); print('$w general', round(d['value']), 'Mreads/s ms/step', round(d['ms_per_step'],2), 'general reads', d['config']['general_path_reads_per_gpu'])"
done
