# device-ingest kernels (text -> tiles): kernel stats and HBM traffic of f2q_count_block on 4M reads of FASTQ text
set -e
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/r02_ingest; mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python $GRAFT_REPO_ROOT/scripts/host_rate.py > $out/host_rate.txt 2> $out/stats.err
for ctr in FETCH_SIZE WRITE_SIZE; do timeout -k 10 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/pmc_$ctr -- python $GRAFT_REPO_ROOT/scripts/host_rate.py > /dev/null 2> $out/pmc_$ctr.err; done
cd $GRAFT_REPO_ROOT
python - <<PY
import csv,glob,os,collections,json
f=sorted(glob.glob('$out/stats/**/*kernel_stats.csv',recursive=True), key=os.path.getmtime)[-1]
rows=list(csv.DictReader(open(f)))
open('$out/kernel_stats.csv','w').write(open(f).read())
for r in rows[:10]: print(r['Name'][:60], r['Calls'], r['AverageNs'])
pm={}
for c in ('FETCH_SIZE','WRITE_SIZE'):
    fs=glob.glob('$out/pmc_'+c+'/**/*counter_collection.csv',recursive=True)
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        k=r['Kernel_Name'].split('(')[0]
        if k.startswith('k_') or 'k_count' in k: agg[k].append(float(r['Counter_Value']))
    for k,v in agg.items(): pm.setdefault(k,{})[c+'_KiB_mean']=sum(v)/len(v); pm[k]['launches']=len(v)
json.dump(pm,open('$out/pmc.json','w'),indent=1)
for k,v in pm.items(): print(k[:40], v)
cat_ = open('$out/host_rate.txt').read(); print(cat_)
PY
