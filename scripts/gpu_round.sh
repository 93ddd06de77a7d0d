# usage: gpu_round.sh <tag> "<pytest -k expression or ->" "<workloads, blank separated, or ->" [stats] [pmc]
#   one gpurun call: a parity subset, then a short bench line per workload (kernel time, frac), then optionally the
#   rocprofv3 kernel statistics (stats) and the HBM / L2 / SQ counters (pmc) of each workload's torch-free child.
#   Environment variables reach the library (F2Q_NO_PT=1, F2Q_PT_CHUNK=...) -- prefix the call with them.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; expr=$2; wls=$3; shift 3 || true
out=gpurun_out/$tag; mkdir -p $out
if [ "$expr" != "-" ]; then
  timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "$expr" > $out/pytest.txt 2>&1 || { tail -40 $out/pytest.txt; exit 1; }
  tail -1 $out/pytest.txt
fi
[ "$wls" = "-" ] && wls=""
for wl in $wls; do
  timeout -k 10 300 python bench.py --workload $wl --steps 20 --no-pmc --no-cpu-baseline --no-extras > $out/bench_$wl.json 2> $out/bench_$wl.err || { tail -20 $out/bench_$wl.err; exit 1; }
  python -c "import json; d=json.load(open('$out/bench_$wl.json')); r=d['roofline']; print('$tag $wl', round(d['value']), 'Mreads/s  ms/step', round(d['ms_per_step'],4), ' kernel_ms', round(r['kernel_ms'],4), 'frac', round(r['frac'],3))"
done
for what in "$@"; do
  for wl in $wls; do
    if [ "$what" = stats ]; then
      timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$wl -- python bench.py --pmc-child --workload $wl > /dev/null 2> $out/stats_$wl.err
      f=$(find $out/stats_$wl -name "*kernel_stats.csv" | head -1)
      cp $f $out/${wl}_kernel_stats.csv
      cut -d, -f1-4,6-7 $f | sed 's/(f2q::[^"]*)//' | head -9
    elif [ "$what" = pmc ]; then
      run() { timeout -k 10 300 rocprofv3 --pmc $1 --kernel-trace --output-format csv -d $out/pmc_$2_$wl -- python bench.py --pmc-child --workload $wl > /dev/null 2> $out/pmc_$2_$wl.err; }
      run "FETCH_SIZE" fetch
      run "WRITE_SIZE" write
      run "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum" tcc
      run "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" sq
      run "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" sq2
      python - > $out/${wl}_pmc.txt <<PY
import csv, glob, collections
for d in ('fetch', 'write', 'tcc', 'sq', 'sq2'):
    fs = glob.glob('$out/pmc_' + d + '_$wl/**/*counter_collection.csv', recursive=True)
    if not fs:
        print(d, 'no csv'); continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        k = r['Kernel_Name'].split('(')[0]
        if k.startswith('void '): k = k[5:]
        if k.startswith('f2q::') or k.startswith('k_'): agg[(k, r['Counter_Name'])].append(float(r['Counter_Value']))
    for (k, c), v in sorted(agg.items()):
        print(f"{k[:48]:48s} {c:24s} launches {len(v):4d}  mean {sum(v) / len(v):16.1f}")
PY
      cat $out/${wl}_pmc.txt
    fi
  done
done
