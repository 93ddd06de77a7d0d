# config 4's kernel (tables in L2 / Infinity Cache) against the library size: does the time follow the table footprint?
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r02_gsweep; mkdir -p $out
for g in 25000 35000 50000 70000 100000 200000 400000; do
  timeout -k 10 300 python bench.py --workload cfg4_50M_100k_m1 --guides $g --steps 10 --warmup 2 --no-extras --no-cpu-baseline > $out/g$g.json 2> $out/g$g.err || { tail -5 $out/g$g.err; exit 1; }
  python - <<PY
import json
d=json.load(open('$out/g$g.json')); r=d['roofline']
print($g, 'ms/step', round(d['ms_per_step'],3), 'kernel', r['kernel'], round(r['kernel_ms'],3), 'traffic B/read', r['traffic_bytes_per_read'] and round(r['traffic_bytes_per_read'],1))
PY
done
