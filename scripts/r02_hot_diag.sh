set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r02_hot_diag; mkdir -p $out
export F2Q_TRACE=1 F2Q_TRACE_SYNC=1
timeout -k 10 200 python bench.py --workload cfg5b_50M_anchor_ec --reads 6000000 --steps 2 --warmup 1 --no-pmc --no-cpu-baseline --no-extras > $out/small.json 2> $out/small.err || { tail -30 $out/small.err; exit 1; }
echo small ok; grep -c "sync" $out/small.err
timeout -k 10 300 python bench.py --workload cfg5b_50M_anchor_ec --steps 2 --warmup 1 --no-pmc --no-cpu-baseline --no-extras > $out/full.json 2> $out/full.err || { tail -30 $out/full.err; exit 1; }
echo full ok
