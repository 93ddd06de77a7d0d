# round 2, first GPU call: fresh binary through the whole GPU suite, the new bench line, rank rehearsals
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02_first
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/r02_first/smoke.txt 2>&1
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r02_first/pytest_gpu.txt 2>&1
tail -3 gpurun_out/r02_first/pytest_gpu.txt
timeout -k 10 600 python bench.py > gpurun_out/r02_first/bench_default.json 2> gpurun_out/r02_first/bench_default.err
cat gpurun_out/r02_first/bench_default.json
# one rank through the RCCL branch
timeout -k 10 300 python bench.py --force-dist --no-pmc --no-cpu-baseline --no-extras > gpurun_out/r02_first/bench_nccl_1rank.json 2> gpurun_out/r02_first/bench_nccl_1rank.err
cat gpurun_out/r02_first/bench_nccl_1rank.json
# two ranks started by bench.py itself, sharing the one GPU: the strong-scaling split (reduced: 50M reads in total), gloo reduction
timeout -k 10 300 python bench.py --gpus 2 --dist-backend gloo --reads 50000000 --steps 5 > gpurun_out/r02_first/bench_2rank_gloo.json 2> gpurun_out/r02_first/bench_2rank_gloo.err
cat gpurun_out/r02_first/bench_2rank_gloo.json
# the same over nccl: RCCL refuses two ranks on one device (expected to fail on this 1-GPU box; the record says how)
timeout -k 10 200 python bench.py --gpus 2 --reads 50000000 --steps 5 > gpurun_out/r02_first/bench_2rank_nccl.json 2> gpurun_out/r02_first/bench_2rank_nccl.err || echo "2-rank nccl on one GPU: exit $?" | tee -a gpurun_out/r02_first/bench_2rank_nccl.json
tail -5 gpurun_out/r02_first/bench_2rank_nccl.err
