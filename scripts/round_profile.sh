# collects the round's rocprofv3 evidence: kernel stats + PMC (traffic) for the bench workload and config 5a
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for wl in cfg3_50M_10k_m1 cfg5a_50M_10k_anchor_m1; do
  out=gpurun_out/final_$wl
  mkdir -p $out
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --workload $wl > $out/bench_under_rocprof.json 2> $out/stats.err
  cp $(find $out/stats -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
  bash scripts/pmc.sh final_${wl}_pmc --workload $wl > $out/pmc.txt 2>&1
  cp gpurun_out/final_${wl}_pmc/summary.json $out/pmc_summary.json
  timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --workload $wl > $out/bench.json 2>/dev/null
  cat $out/kernel_stats.csv | cut -c1-160
done
# kernel stats only for the other BASELINE configurations
for wl in cfg2_10M_1k_m0 cfg4_50M_100k_m1 cfg5b_50M_anchor_ec; do
  out=gpurun_out/final_$wl
  mkdir -p $out
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --workload $wl > $out/bench_under_rocprof.json 2> $out/stats.err
  cp $(find $out/stats -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
  timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --workload $wl > $out/bench.json 2>/dev/null
  cat $out/kernel_stats.csv | cut -c1-160
done
F2Q_TRACE=1 timeout -k 10 300 python scripts/file_rate.py > gpurun_out/final_file_ingest.txt 2>&1 || true
grep -v amdgpu.ids gpurun_out/final_file_ingest.txt
timeout -k 10 600 python bench.py > gpurun_out/final_bench_default.json 2>/dev/null
cat gpurun_out/final_bench_default.json
