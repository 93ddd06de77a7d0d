# k_count_fixed4<false,5,2,PRE>: rows prefetched into LDS -- parity (config 4 tests) and A/B against F2Q_NO_PRE=1
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r02_pre; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "config4 or range_histogram" > $out/pytest.txt 2>&1 || { tail -30 $out/pytest.txt; exit 1; }
tail -1 $out/pytest.txt
run() { name=$1; shift; timeout -k 10 300 env "$@" python bench.py --workload cfg4_50M_100k_m1 --steps 10 --no-pmc --no-cpu-baseline --no-extras > $out/$name.json 2> $out/$name.err || { tail -5 $out/$name.err; exit 1; }; python -c "import json; d=json.load(open('$out/$name.json')); print('$name', round(d['value']), 'Mreads/s kernel_ms', round(d['roofline']['kernel_ms'],3), 'frac', round(d['roofline']['frac'],3), d['verify'].get('identity_reads_eq_sum_of_outcomes'), d['verify']['stats'])"; }
run pre F2Q_X=1
run nopre F2Q_NO_PRE=1
