set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r02_pair; mkdir -p $out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $out/pytest.txt 2>&1 || { tail -40 $out/pytest.txt; exit 1; }
tail -1 $out/pytest.txt
for w in cfg4_50M_100k_m1 cfg3_2win_50M_10k_m1 cfg3_50M_10k_m1; do
  timeout -k 10 300 python bench.py --workload $w --steps 10 --warmup 2 --no-pmc --no-cpu-baseline --no-extras > $out/x.json 2> $out/x.err
  python -c "import json; d=json.load(open('$out/x.json')); print('$w', 'kernel_ms', round(d['roofline']['kernel_ms'],4), 'frac', round(d['roofline']['frac'],3))"
done
F2Q_NO_LT=1 timeout -k 10 300 python bench.py --workload cfg5a_50M_10k_anchor_m1 --steps 10 --warmup 2 --no-pmc --no-cpu-baseline --no-extras > $out/x.json 2> $out/x.err
python -c "import json; d=json.load(open('$out/x.json')); print('cfg5a NO_LT', 'kernel_ms', round(d['roofline']['kernel_ms'],4))"
