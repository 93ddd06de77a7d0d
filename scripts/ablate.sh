# VALU per tile of k_count_fixed4_lds with parts switched off (where the instructions go)
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/ablate; mkdir -p $out
for cfg in "m1_ph30" "m0_ph30:--miss 0" "m1_ph1:--phred 1" "m0_ph1:--miss 0 --phred 1" "m1_ph30_nonN:--p-n 0"; do
  name=${cfg%%:*}; args=""; [ "$cfg" != "$name" ] && args=${cfg#*:}
  timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $out/$name -- python bench.py --pmc-child --workload cfg3_50M_10k_m1 $args > /dev/null 2> $out/$name.err
  python - <<PY
import csv,glob,collections
fs=glob.glob('$out/$name/**/*counter_collection.csv', recursive=True)
agg=collections.defaultdict(list); dur=[]
for r in csv.DictReader(open(fs[0])):
    if 'k_count_fixed4' in r['Kernel_Name']:
        agg[r['Counter_Name']].append(float(r['Counter_Value']))
        if r['Counter_Name']=='SQ_INSTS_VALU': dur.append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
print('$name', 'dur_us', round(sum(dur)/len(dur)/1e3,1), ' '.join(f"{k}={sum(v)/len(v)/195313:.0f}/tile" for k,v in sorted(agg.items())))
PY
done
