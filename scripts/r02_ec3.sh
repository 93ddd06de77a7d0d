set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r02_ec3; mkdir -p $out
run() { name=$1; shift; timeout -k 10 300 python bench.py --workload cfg5b_50M_anchor_ec --steps 3 --no-pmc --no-cpu-baseline --no-extras "$@" > $out/$name.json 2> $out/$name.err; python -c "import json; d=json.load(open('$out/$name.json')); print('$name', round(d['value']), 'Mreads/s kernel_ms', round(d['roofline']['kernel_ms'],3))"; }
run pn0 --p-n 0
run pn0005 --p-n 0.005
run pn0001 --p-n 0.001
cd /tmp; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/stats -- python $GRAFT_REPO_ROOT/bench.py --pmc-child --workload cfg5b_50M_anchor_ec > /dev/null 2> $GRAFT_REPO_ROOT/$out/stats.err; cd $GRAFT_REPO_ROOT
f=$(find $out/stats -name "*kernel_stats.csv" | head -1); python - <<PY
import csv
for r in csv.DictReader(open("$f")): print(r['Name'][:50], r['Calls'], r['AverageNs'])
PY
