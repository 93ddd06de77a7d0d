set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r02_an_$1; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "anchor or config5 or fuzz or golden" > $out/pytest.txt 2>&1 || { tail -30 $out/pytest.txt; exit 1; }
tail -1 $out/pytest.txt
for v in lt nolt; do
  if [ $v = nolt ]; then export F2Q_NO_LT=1; else unset F2Q_NO_LT; fi
  timeout -k 10 300 python bench.py --workload cfg5a_50M_10k_anchor_m1 --steps 10 --no-pmc --no-cpu-baseline --no-extras > $out/$v.json 2> $out/$v.err; python -c "import json; d=json.load(open('$out/$v.json')); print('$v', round(d['value']), 'Mreads/s kernel_ms', round(d['roofline']['kernel_ms'],3), 'frac', round(d['roofline']['frac'],3))"
done
unset F2Q_NO_LT
if [ "$2" = pmc ]; then
  run() { timeout -k 10 300 rocprofv3 --pmc $1 --kernel-trace --output-format csv -d $out/$2 -- python bench.py --pmc-child --workload cfg5a_50M_10k_anchor_m1 > /dev/null 2> $out/$2.err; }
  run "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" sq
  python - <<PY
import csv,glob,collections
fs=glob.glob('$out/sq/**/*counter_collection.csv', recursive=True)
agg=collections.defaultdict(list)
for r in csv.DictReader(open(fs[0])):
    k=r['Kernel_Name'].split('(')[0]
    if 'k_count' in k: agg[(k,r['Counter_Name'])].append(float(r['Counter_Value']))
for (k,c),v in sorted(agg.items()): print(k[:44],c,round(sum(v)/len(v)/1e6,3),'M')
PY
fi
