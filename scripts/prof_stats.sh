# usage: prof_stats.sh <outdir-name> <bench args...>   -> prints the kernel stats CSV
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
name=$1; shift
mkdir -p gpurun_out/$name
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$name -- python bench.py --no-cpu-baseline "$@" > gpurun_out/$name/bench.json 2> gpurun_out/$name/err.log
f=$(find gpurun_out/$name -name "*kernel_stats.csv" | head -1)
cut -d, -f1-4,6-7 $f | sed 's/(f2q::RunDev const\*, f2q::LibDev const\*, f2q::PackedBlock, f2q::Accum)//' | head -8
