cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r02_hot_t; mkdir -p $out
F2Q_TRACE=1 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "ec_hot" -s > $out/pytest.txt 2>&1; echo rc=$?
grep -v amdgpu.ids $out/pytest.txt | tail -25
