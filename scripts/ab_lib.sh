# usage: ab_lib.sh <tag> <alt .so> <workload...>   the same bench line with the tree's library and with another build of it (A/B)
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; alt=$2; shift 2; out=gpurun_out/$tag; mkdir -p $out
for wl in "$@"; do for v in base alt; do
  if [ $v = alt ]; then export F2Q_LIB_PATH=$GRAFT_REPO_ROOT/$alt; else unset F2Q_LIB_PATH; fi
  timeout -k 10 300 python bench.py --workload $wl --steps 20 --no-pmc --no-cpu-baseline --no-extras > $out/bench_${wl}_$v.json 2> $out/bench_${wl}_$v.err
  python -c "import json; d=json.load(open('$out/bench_${wl}_$v.json')); r=d['roofline']; print('$v $wl', round(d['value']), 'Mreads/s  kernel_ms', round(r['kernel_ms'],4), 'frac', round(r['frac'],3))"
done; done
