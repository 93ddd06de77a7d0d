set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r02_hot2; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "ec_ or extract_count or anchor or golden or fuzz or config5" > $out/pytest.txt 2>&1 || { tail -40 $out/pytest.txt; exit 1; }
tail -1 $out/pytest.txt
for w in cfg5b_50M_anchor_ec cfg3b_50M_fixed_ec; do
timeout -k 10 300 python bench.py --workload $w --steps 6 --warmup 2 --no-pmc --no-cpu-baseline --no-extras > $out/bench_$w.json 2> $out/bench_$w.err || { grep -v amdgpu.ids $out/bench_$w.err | tail -5; exit 1; }
python -c "import json; d=json.load(open('$out/bench_$w.json')); print('$w', round(d['value']), 'Mreads/s kernel_ms', round(d['roofline']['kernel_ms'],3), 'frac', round(d['roofline']['frac'],3), d['verify']['stats'])"
done
