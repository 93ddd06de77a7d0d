set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r02_ec4; mkdir -p $out
run() { name=$1; shift; timeout -k 10 300 env "$@" python bench.py --workload cfg5b_50M_anchor_ec --steps 4 --no-pmc --no-cpu-baseline --no-extras > $out/$name.json 2> $out/$name.err; python -c "import json; d=json.load(open('$out/$name.json')); print('$name', round(d['value']), 'Mreads/s kernel_ms', round(d['roofline']['kernel_ms'],3), 'ms_per_step', round(d['ms_per_step'],3))"; }
run cached_8M F2Q_X=1
run cached_16M F2Q_EC_STEP=16777216
run cached_32M F2Q_EC_STEP=33554432
run cached_64M F2Q_EC_STEP=67108864
run old_8M F2Q_NO_LT=1
run old_64M F2Q_NO_LT=1 F2Q_EC_STEP=67108864
