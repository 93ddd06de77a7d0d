set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r02_ec2; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cli.py -m gpu -x -q -k "ec or EC or extract or anchor or config5 or fuzz or golden" > $out/pytest.txt 2>&1 || { tail -30 $out/pytest.txt; exit 1; }
tail -1 $out/pytest.txt
for wl in cfg5b_50M_anchor_ec cfg5a_50M_10k_anchor_m1; do
timeout -k 10 300 python bench.py --workload $wl --steps 5 --no-pmc --no-cpu-baseline --no-extras > $out/$wl.json 2> $out/$wl.err; python -c "import json; d=json.load(open('$out/$wl.json')); print('$wl', round(d['value']), 'Mreads/s kernel_ms', round(d['roofline']['kernel_ms'],3), 'ms_per_step', round(d['ms_per_step'],3), 'frac', round(d['roofline']['frac'],3), 'general', d['config']['general_path_reads_per_gpu'])"
done
cd /tmp; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/stats -- python $GRAFT_REPO_ROOT/bench.py --pmc-child --workload cfg5b_50M_anchor_ec > /dev/null 2> $GRAFT_REPO_ROOT/$out/stats.err; cd $GRAFT_REPO_ROOT
f=$(find $out/stats -name "*kernel_stats.csv" | head -1); cut -d, -f1-4 $f | cut -c1-60,150-250 | head -8
