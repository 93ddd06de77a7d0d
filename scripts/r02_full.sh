set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r02_full; mkdir -p $out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $out/pytest.txt 2>&1 || { tail -40 $out/pytest.txt; exit 1; }
tail -1 $out/pytest.txt
timeout -k 10 400 python bench.py --workload cfg5b_50M_anchor_ec --no-cpu-baseline --no-extras > $out/bench_cfg5b.json 2> $out/bench_cfg5b.err || { grep -v amdgpu.ids $out/bench_cfg5b.err | tail; exit 1; }
python -c "import json; d=json.load(open('$out/bench_cfg5b.json')); r=d['roofline']; print('cfg5b', round(d['value']), 'Mreads/s kernel_ms', round(r['kernel_ms'],4), 'frac', round(r['frac'],3), 'traffic B/read', r.get('traffic_bytes_per_read'), d['verify'])"
