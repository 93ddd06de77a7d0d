set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r02_bgzf; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_cli.py -m gpu -x -q > $out/pytest.txt 2>&1 || { tail -40 $out/pytest.txt; exit 1; }
tail -1 $out/pytest.txt
timeout -k 10 700 bash scripts/rehearse_ranks.sh > $out/rehearse.txt 2>&1 || true
tail -22 gpurun_out/rehearse/progress.txt
