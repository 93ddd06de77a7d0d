set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r02_cfg4; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "config4 or device_synth or lds_table" > $out/pytest.txt 2>&1 || { tail -20 $out/pytest.txt; exit 1; }
tail -1 $out/pytest.txt
run() { name=$1; wl=$2; shift; shift; timeout -k 10 300 env "$@" python bench.py --workload $wl --steps 10 --no-pmc --no-cpu-baseline --no-extras > $out/$name.json 2> $out/$name.err; python -c "import json; d=json.load(open('$out/$name.json')); print('$name', round(d['value']), 'Mreads/s kernel_ms', round(d['roofline']['kernel_ms'],3), 'frac', round(d['roofline']['frac'],3))"; }
run cfg4_gt cfg4_50M_100k_m1 F2Q_X=1
run cfg4_old cfg4_50M_100k_m1 F2Q_NO_LT=1
cd /tmp; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/stats -- python $GRAFT_REPO_ROOT/bench.py --pmc-child --workload cfg4_50M_100k_m1 > /dev/null 2> $GRAFT_REPO_ROOT/$out/stats.err; cd $GRAFT_REPO_ROOT
python - <<PY
import csv,glob,os
f=sorted(glob.glob('$out/stats/**/*kernel_stats.csv',recursive=True), key=os.path.getmtime)[-1]
for r in list(csv.DictReader(open(f)))[:5]: print(r['Name'][:60], r['Calls'], r['AverageNs'])
PY
