cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/r02_hot_pmc; mkdir -p $out
i=0
for ctr in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_FLAT" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/p$i -- python3 $GRAFT_REPO_ROOT/bench.py --pmc-child --workload cfg5b_50M_anchor_ec > /dev/null 2> $out/p$i.err || echo "pass $i failed"
done
python3 - <<'PY'
import csv, glob, os, collections
out=os.environ.get("GRAFT_REPO_ROOT","/root/repo")+"/gpurun_out/r02_hot_pmc"
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out+"/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:34]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in agg.items():
    if "extract_anchor" in k:
        print(k, {a: round(max(b)) for a,b in sorted(v.items())})
PY
