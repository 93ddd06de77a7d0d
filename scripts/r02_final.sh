# round 2, last GPU call: fresh binary through smoke + the whole GPU suite, the default bench line (timed), rank rehearsals
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r02_final; mkdir -p $out
python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('smoke ok')" > $out/smoke.txt 2>&1 || { tail -20 $out/smoke.txt; exit 1; }
tail -1 $out/smoke.txt
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.txt 2>&1 || { tail -30 $out/pytest_gpu.txt; exit 1; }
tail -1 $out/pytest_gpu.txt
t0=$(date +%s)
timeout -k 10 900 python bench.py > $out/bench_default.json 2> $out/bench_default.err || { grep -v amdgpu.ids $out/bench_default.err | tail; exit 1; }
t1=$(date +%s); echo "default bench.py: $((t1 - t0)) s wall" | tee $out/bench_default_wall.txt
cat $out/bench_default.json
timeout -k 10 300 python bench.py --force-dist --no-pmc --no-cpu-baseline --no-extras > $out/bench_nccl_1rank.json 2> $out/bench_nccl_1rank.err
python -c "import json; d=json.load(open('$out/bench_nccl_1rank.json')); print('1 rank through RCCL:', round(d['value']), d['config'].get('collective'))"
timeout -k 10 400 python bench.py --gpus 2 --dist-backend gloo --reads 50000000 --steps 5 > $out/bench_2rank_gloo.json 2> $out/bench_2rank_gloo.err
python -c "import json; d=json.load(open('$out/bench_2rank_gloo.json')); print('2 ranks (gloo, one GPU):', round(d['value']), d['scaling'], d['config']['workload'], d['verify'].get('all_ranks_equal'))"
timeout -k 10 400 python bench.py --gpus 4 --dist-backend gloo --reads 40000000 --steps 3 > $out/bench_4rank_gloo.json 2> $out/bench_4rank_gloo.err
python -c "import json; d=json.load(open('$out/bench_4rank_gloo.json')); print('4 ranks (gloo, one GPU):', round(d['value']), d['scaling'], d['n_gpus'], d['config']['workload'], d['verify'])"
