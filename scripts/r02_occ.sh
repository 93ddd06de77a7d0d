# config 4's kernel at one and two workgroups per CU: latency-bound (time doubles) or throughput-bound (time stays)?
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r02_occ; mkdir -p $out
for m in 1 2 3; do
  F2Q_V2_GRID=$m timeout -k 10 300 python bench.py --workload cfg4_50M_100k_m1 --steps 10 --warmup 2 --no-extras --no-cpu-baseline --no-pmc > $out/m$m.json 2> $out/m$m.err || { tail -5 $out/m$m.err; exit 1; }
  python -c "import json; d=json.load(open('$out/m$m.json')); print('wgs/CU', $m, 'ms/step', round(d['ms_per_step'],3), 'kernel_ms', round(d['roofline']['kernel_ms'],3))"
done
