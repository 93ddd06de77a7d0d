set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r02_ingest2; mkdir -p $out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $out/pytest.txt 2>&1 || { tail -40 $out/pytest.txt; exit 1; }
tail -1 $out/pytest.txt
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/stats -- python3 $GRAFT_REPO_ROOT/scripts/host_rate.py > $GRAFT_REPO_ROOT/$out/host_rate.txt 2> $GRAFT_REPO_ROOT/$out/stats.err
cd $GRAFT_REPO_ROOT && cat $out/host_rate.txt
f=$(find $out/stats -name "*kernel_stats.csv" | head -1); test -n "$f" && head -8 $f | cut -c1-150
