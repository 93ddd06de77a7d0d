# round 2 evidence: per workload the bench line (with live PMC traffic) and the rocprofv3 kernel stats of the same command
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r02_prof; mkdir -p $out
for wl in cfg3_50M_10k_m1 cfg2_10M_1k_m0 cfg4_50M_100k_m1 cfg5a_50M_10k_anchor_m1 cfg5b_50M_anchor_ec cfg3_2win_50M_10k_m1 cfg5c_2pair_50M_10k_m1 cfg3b_50M_fixed_ec; do
  timeout -k 10 400 python bench.py --workload $wl --steps 10 --warmup 2 --no-extras > $out/${wl}_bench.json 2> $out/${wl}_bench.err || echo "bench $wl failed"
  ( cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/${wl}_stats -- python $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 10 --warmup 2 --no-pmc --no-cpu-baseline --no-extras > $GRAFT_REPO_ROOT/$out/${wl}_bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$out/${wl}_stats.err ) || echo "stats $wl failed"
  f=$(ls -t $out/${wl}_stats/*/*kernel_stats.csv | head -1); cp $f $out/${wl}_kernel_stats.csv
  python - <<PY
import json,csv
d=json.load(open('$out/${wl}_bench.json')); r=d['roofline']
print('$wl', round(d['value']), 'Mreads/s kernel_ms', round(r['kernel_ms'],4), 'frac', round(r['frac'],3), 'traffic B/read', r['traffic_bytes_per_read'] and round(r['traffic_bytes_per_read'],1), 'cpu', d.get('cpu_baseline',{}).get('value'))
for row in list(csv.DictReader(open('$out/${wl}_kernel_stats.csv')))[:3]: print('    ', row['Name'][:70], row['Calls'], row['AverageNs'])
PY
done
# SQ / TCC counters of the dominant kernels (separate passes)
for wl in cfg3_50M_10k_m1 cfg4_50M_100k_m1 cfg5a_50M_10k_anchor_m1 cfg5b_50M_anchor_ec; do
  for grp in "sq:SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "sq2:SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU" "tcc:TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum" "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
    g=${grp%%:*}; ctr=${grp#*:}
    ( cd /tmp && timeout -k 10 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/pmc_${wl}_$g -- python $GRAFT_REPO_ROOT/bench.py --pmc-child --workload $wl > /dev/null 2> $GRAFT_REPO_ROOT/$out/pmc_${wl}_$g.err ) || echo "pmc $wl $g failed"
  done
  python - <<PY
import csv,glob,collections,json
out={}
for g in ('sq','sq2','tcc','fetch','write'):
    fs=glob.glob('$out/pmc_${wl}_'+g+'/**/*counter_collection.csv', recursive=True)
    if not fs: continue
    agg=collections.defaultdict(list); dur=collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        k=r['Kernel_Name'].split('(')[0]
        if 'synth' in k or k.startswith('__amd'): continue
        agg[(k,r['Counter_Name'])].append(float(r['Counter_Value'])); dur[k].append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
    for (k,c),v in agg.items():
        out.setdefault(k,{})[c]=sum(v)/len(v); out[k]['duration_ns_'+g]=sum(dur[k])/len(dur[k])
json.dump(out,open('$out/${wl}_pmc.json','w'),indent=1)
print('$wl pmc kernels:', list(out))
PY
done
