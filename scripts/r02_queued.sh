set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r02_queued; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "queued or full_size_invariants or library_is_built" > $out/pytest.txt 2>&1 || { tail -30 $out/pytest.txt; exit 1; }
tail -1 $out/pytest.txt
for wl in cfg3_50M_10k_m1 cfg2_10M_1k_m0 cfg4_50M_100k_m1 cfg5a_50M_10k_anchor_m1 cfg5b_50M_anchor_ec; do
  timeout -k 10 300 python bench.py --workload $wl --steps 10 --warmup 2 --no-extras --no-cpu-baseline --no-pmc > $out/$wl.json 2> $out/$wl.err || { tail -5 $out/$wl.err; exit 1; }
  python -c "import json; d=json.load(open('$out/$wl.json')); r=d['roofline']; print('$wl', round(d['value']), 'Mreads/s ms/step', round(d['ms_per_step'],4), 'kernel_ms', round(r['kernel_ms'],4), 'frac', round(r['frac'],3), d['verify'].get('identity_reads_eq_sum_of_outcomes'))"
done
timeout -k 10 300 python bench.py --force-dist --no-pmc --no-cpu-baseline --no-extras > $out/nccl1.json 2> $out/nccl1.err
python -c "import json; d=json.load(open('$out/nccl1.json')); print('1 rank through RCCL:', round(d['value']), d['ms_per_step'], d['verify'])"
