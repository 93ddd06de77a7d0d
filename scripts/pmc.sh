# usage: pmc.sh <name> <bench args>  -> SQ + TCC counters of the counting kernels (separate passes)
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
name=$1; shift
mkdir -p gpurun_out/$name
run() { timeout -k 10 300 rocprofv3 --pmc $1 --kernel-trace --output-format csv -d gpurun_out/$name/$2 -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline "${@:3}" > gpurun_out/$name/$2.json 2> gpurun_out/$name/$2.err; }
run "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" sq "$@"
run "SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS" sq2 "$@"
run "FETCH_SIZE" fetch "$@"
run "WRITE_SIZE" write "$@"
run "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum" tcc "$@"
run "GRBM_GUI_ACTIVE" grbm "$@"
python - <<PY
import csv,glob,collections,json
out={}
for d in ('sq','sq2','fetch','write','tcc','grbm'):
    fs=glob.glob(f'gpurun_out/$name/{d}/*/*counter_collection.csv')
    if not fs: continue
    agg=collections.defaultdict(list); dur=collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        k=r['Kernel_Name'].split('(')[0]
        agg[(k,r['Counter_Name'])].append(float(r['Counter_Value']))
        dur[k].append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
    for (k,c),v in sorted(agg.items()):
        if k.startswith('__amd') or 'synth' in k: continue
        out.setdefault(k,{})[c]=sum(v)/len(v)
        out[k]['duration_ns_'+d]=sum(dur[k])/len(dur[k])
json.dump(out,open('gpurun_out/$name/summary.json','w'),indent=1)
for k,v in out.items():
    print(k)
    for c,x in v.items(): print('   ',c,round(x,1))
PY
