# usage: round_profile.sh <round tag, e.g. r03>   the round's evidence in ONE gpurun call, written to gpurun_out/<tag>_profiles/
#   (copy what is to be judged into profiles/):
#   per workload: the bench line, rocprofv3 kernel statistics and the PMC counters (HBM bytes, L2, SQ) of the torch-free
#   child; the ingest path (host text -> tiles) under rocprofv3; the default bench line; the .gz reader rates.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; out=gpurun_out/${tag}_profiles; mkdir -p $out
WLS="cfg3_50M_10k_m1 cfg2_10M_1k_m0 cfg4_50M_100k_m1 cfg5a_50M_10k_anchor_m1 cfg5b_50M_anchor_ec cfg3b_50M_fixed_ec cfg3_2win_50M_10k_m1 cfg5c_2pair_50M_10k_m1"
bash scripts/gpu_round.sh ${tag}_profiles - "$WLS" stats pmc > $out/round.txt 2>&1 || { tail -30 $out/round.txt; exit 1; }
grep "Mreads/s" $out/round.txt
# the ingest path: FASTQ text in host memory -> f2q_count_block (k_nl_count, k_scan_*, k_line_starts, k_classify, k_pack + counting)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/ingest -- python scripts/host_rate.py > $out/ingest_host_rate.txt 2> $out/ingest.err
cp $(find $out/ingest -name "*kernel_stats.csv" | head -1) $out/ingest_kernel_stats.csv
cut -d, -f1-4 $out/ingest_kernel_stats.csv | head -14
timeout -k 10 600 python bench.py > $out/bench_default.json 2> $out/bench_default.err
python -c "import json; d=json.load(open('$out/bench_default.json')); print('default', round(d['value']), d['roofline']['frac'], d['end_to_end'], d.get('strong_scaling_n1'))"
