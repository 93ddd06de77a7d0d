# usage: stamp.sh <workload...>   phase stamps of a -DF2Q_STAMP build (2fast2q_amd/lib/libf2q_hip_stamp.so, built by hand:
#   hipcc ... -DF2Q_STAMP -o 2fast2q_amd/lib/libf2q_hip_stamp.so 2fast2q_amd/csrc/f2q_lib.hip -lz)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export F2Q_LIB_PATH=$GRAFT_REPO_ROOT/2fast2q_amd/lib/libf2q_hip_stamp.so
for w in "$@"; do
  timeout -k 10 300 python bench.py --workload $w --steps 3 --warmup 1 --no-pmc --no-cpu-baseline --no-extras 2>&1 >/dev/null | grep stamp | tail -2
done
