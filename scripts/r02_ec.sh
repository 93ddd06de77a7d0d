set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r02_ec; mkdir -p $out
run() { name=$1; shift; timeout -k 10 300 env "$@" python bench.py --workload cfg5b_50M_anchor_ec --steps 5 --no-pmc --no-cpu-baseline --no-extras $EXTRA > $out/$name.json 2> $out/$name.err; python -c "import json; d=json.load(open('$out/$name.json')); print('$name', round(d['value']), 'Mreads/s kernel_ms', round(d['roofline']['kernel_ms'],3), 'ms_per_step', round(d['ms_per_step'],3))"; }
run step8M F2Q_X=1
run step4M F2Q_EC_STEP=4194304
run step2M F2Q_EC_STEP=2097152
run step1M F2Q_EC_STEP=1048576
run step16M F2Q_EC_STEP=16777216
EXTRA="--p-n 0" run step8M_noN F2Q_X=1
EXTRA="--p-n 0" run step2M_noN F2Q_EC_STEP=2097152
cd /tmp; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/stats -- python $GRAFT_REPO_ROOT/bench.py --pmc-child --workload cfg5b_50M_anchor_ec > /dev/null 2> $GRAFT_REPO_ROOT/$out/stats.err; cd $GRAFT_REPO_ROOT
f=$(find $out/stats -name "*kernel_stats.csv" | head -1); cut -d, -f1-4,6-7 $f | cut -c1-150 | head -12
