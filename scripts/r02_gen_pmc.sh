cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/r02_gen_pmc; mkdir -p $out
export F2Q_NO_HOT=1 F2Q_GEN_GRID=6
i=0
for ctr in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_FLAT" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_BRANCH SQ_INSTS_FLAT_LDS_ONLY SQ_LDS_BANK_CONFLICT SQ_INSTS_GDS"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/p$i -- python3 $GRAFT_REPO_ROOT/bench.py --pmc-child --workload cfg5b_50M_anchor_ec > /dev/null 2> $out/p$i.err || echo "pass $i failed"
  echo "pass $i done"
done
python3 - <<'PY'
import csv, glob, os, collections
out=os.environ.get("GRAFT_REPO_ROOT","/root/repo")+"/gpurun_out/r02_gen_pmc"
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for f in glob.glob(out+"/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"][:40]
        agg[k][r["Counter_Name"]]+=float(r["Counter_Value"])
for k,v in agg.items():
    if "general" in k or "anchor" in k:
        print(k, {a: round(b/4) for a,b in sorted(v.items())})
PY
