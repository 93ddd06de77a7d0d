set -e
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/r02_hot_prof; mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out/prof -o hot -- python3 $GRAFT_REPO_ROOT/bench.py --workload cfg5b_50M_anchor_ec --steps 3 --warmup 1 --no-pmc --no-cpu-baseline --no-extras > $out/prof.log 2>&1
f=$(find $out/prof -name "*kernel_stats.csv" | head -1); head -16 $f | cut -c1-160
tail -2 $out/prof.log | cut -c1-400
