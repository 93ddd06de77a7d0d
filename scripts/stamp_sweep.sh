cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export F2Q_LIB_PATH=$GRAFT_REPO_ROOT/2fast2q_amd/lib/libf2q_hip_stamp.so
for ch in 2000000 4000000 8388608 16777216 50000000; do
  echo chunk $ch
  F2Q_PT_CHUNK=$ch timeout -k 10 300 python bench.py --workload cfg4_50M_100k_m1 --steps 3 --warmup 1 --no-pmc --no-cpu-baseline --no-extras 2>&1 >/dev/null | grep stamp | tail -1
done
