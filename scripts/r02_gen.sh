cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/r02_gen; mkdir -p $out
export F2Q_NO_HOT=1
for g in 64 6 2; do
  export F2Q_GEN_GRID=$g
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/prof_$g -o gen -- python3 $GRAFT_REPO_ROOT/bench.py --workload cfg5b_50M_anchor_ec --steps 2 --warmup 1 --no-pmc --no-cpu-baseline --no-extras > $out/prof_$g.log 2>&1
  echo "grid x$g done"
done
