set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r02_pre; mkdir -p $out
run() { name=$1; shift; timeout -k 10 300 env "$@" python bench.py --workload cfg4_50M_100k_m1 --steps 10 --no-pmc --no-cpu-baseline --no-extras > $out/$name.json 2> $out/$name.err || { tail -5 $out/$name.err; exit 1; }; python -c "import json; d=json.load(open('$out/$name.json')); print('$name', round(d['value']), 'Mreads/s kernel_ms', round(d['roofline']['kernel_ms'],3), 'frac', round(d['roofline']['frac'],3))"; }
run pre_g1 F2Q_V2_GRID=1
run nopre_g1 F2Q_NO_PRE=1 F2Q_V2_GRID=1
run pre_g2 F2Q_V2_GRID=2
run nopre_g2 F2Q_NO_PRE=1 F2Q_V2_GRID=2
