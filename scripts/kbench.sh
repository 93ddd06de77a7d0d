# usage: kbench.sh <tag> [pmc]   quick parity subset + kernel time of the fixed-offset workloads (+ SQ/TCC counters)
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; out=gpurun_out/kb_$tag; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "lds or device_synth or config3" > $out/pytest.txt 2>&1 || { tail -30 $out/pytest.txt; exit 1; }
tail -1 $out/pytest.txt
for wl in cfg3_50M_10k_m1 cfg2_10M_1k_m0; do
  timeout -k 10 200 python bench.py --workload $wl --steps 20 --no-pmc --no-cpu-baseline --no-extras > $out/bench_$wl.json 2> $out/bench_$wl.err
  python -c "import json; d=json.load(open('$out/bench_$wl.json')); print('$tag $wl', round(d['value']), 'Mreads/s kernel_ms', round(d['roofline']['kernel_ms'],4), 'frac', round(d['roofline']['frac'],3))"
done
if [ "$2" = pmc ]; then
  run() { timeout -k 10 300 rocprofv3 --pmc $1 --kernel-trace --output-format csv -d $out/$2 -- python bench.py --pmc-child --workload cfg3_50M_10k_m1 > /dev/null 2> $out/$2.err; }
  run "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" sq
  run "SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU" sq2
  run "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum" tcc
  python - <<PY
import csv,glob,collections
for d in ('sq','sq2','tcc'):
    fs=glob.glob('$out/'+d+'/**/*counter_collection.csv', recursive=True)
    if not fs: print(d,'no csv'); continue
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        k=r['Kernel_Name'].split('(')[0]
        if 'k_count' in k: agg[(k,r['Counter_Name'])].append(float(r['Counter_Value']))
    for (k,c),v in sorted(agg.items()): print(k[:40],c,round(sum(v)/len(v)/1e6,3),'M')
PY
fi
