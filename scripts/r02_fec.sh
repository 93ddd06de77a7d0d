set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r02_fec; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "ec_ or extract_count or golden or fuzz" > $out/pytest.txt 2>&1 || { tail -40 $out/pytest.txt; exit 1; }
tail -1 $out/pytest.txt
for v in hot nohot; do
if [ $v = nohot ]; then export F2Q_NO_HOT=1; else unset F2Q_NO_HOT; fi
timeout -k 10 300 python bench.py --workload cfg3b_50M_fixed_ec --steps 5 --warmup 2 --no-pmc --no-cpu-baseline --no-extras > $out/bench_$v.json 2> $out/bench_$v.err || { grep -v amdgpu.ids $out/bench_$v.err | tail -5; exit 1; }
python -c "import json; d=json.load(open('$out/bench_$v.json')); print('fixed EC $v', round(d['value']), 'Mreads/s kernel_ms', round(d['roofline']['kernel_ms'],3), 'frac', round(d['roofline']['frac'],3), d['verify']['stats'])"
done
