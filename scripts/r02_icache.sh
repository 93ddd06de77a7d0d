# instruction-cache counters of the fixed-offset kernels (config 3: library in LDS; config 4: tables in L2)
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r02_icache; mkdir -p $out
rocprofv3 --list-avail > $out/avail.txt 2>&1 || true
grep -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_INST_LEVEL[A-Z_]*\|SQ_WAIT_IFETCH[A-Z_]*\|SQ_INSTS_[A-Z_]*" $out/avail.txt | sort -u | tr '\n' ' '; echo
for wl in cfg4_50M_100k_m1 cfg3_50M_10k_m1; do
  ( cd /tmp && timeout -k 10 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVES --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/$wl -- python $GRAFT_REPO_ROOT/bench.py --pmc-child --workload $wl > /dev/null 2> $GRAFT_REPO_ROOT/$out/$wl.err ) || { tail -5 $out/$wl.err; echo "pmc $wl failed"; }
  python - <<PY
import csv,glob,collections
fs=glob.glob('$out/$wl/**/*counter_collection.csv', recursive=True)
agg=collections.defaultdict(list)
for r in csv.DictReader(open(fs[0])):
    k=r['Kernel_Name'].split('(')[0]
    if 'synth' in k or k.startswith('__amd'): continue
    agg[(k,r['Counter_Name'])].append(float(r['Counter_Value']))
for (k,c),v in sorted(agg.items()): print('$wl', k[:50], c, round(sum(v)/len(v)))
PY
done
