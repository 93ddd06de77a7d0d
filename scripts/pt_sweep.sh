# usage: pt_sweep.sh <tag> <chunk sizes...>   config 4 per GPU with different F2Q_PT_CHUNK (reads per scatter/count round)
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift; out=gpurun_out/$tag; mkdir -p $out
for ch in "$@"; do
  F2Q_PT_CHUNK=$ch timeout -k 10 300 python bench.py --workload cfg4_50M_100k_m1 --steps 20 --no-pmc --no-cpu-baseline --no-extras > $out/bench_$ch.json 2> $out/bench_$ch.err
  python -c "import json; d=json.load(open('$out/bench_$ch.json')); r=d['roofline']; print('chunk $ch', round(d['value']), 'Mreads/s  ms/step', round(d['ms_per_step'],4), ' kernel_ms', round(r['kernel_ms'],4), 'frac', round(r['frac'],3))"
done
