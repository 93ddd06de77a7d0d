# usage: gz_trace_box.sh <tag>   per-round phases of the parallel gzip decoder on the GPU box's host (F2Q_GZ_TRACE), 16 / 8 threads
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; out=gpurun_out/$tag; mkdir -p $out
g++ -O2 -std=c++17 -o /tmp/reader_speed scripts/reader_speed.cpp -lz -lpthread
python - > $out/make.txt 2>&1 <<PY
import importlib, subprocess
pkg = importlib.import_module("2fast2q_amd")
guides = pkg.binding.synth_library(0xF2A5 + 3, 10000, 20)
with pkg.Counter(features=guides, miss=1, phred=30, length=20, start="0") as c:
    fq = bytes(c.synth_fastq(seed=0xBEEF, n_reads=4_000_000, read_len=150))
open("/tmp/x.fastq", "wb").write(fq)
subprocess.check_call("gzip -1 -c /tmp/x.fastq > /tmp/x1.fastq.gz", shell=True)
PY
nproc; grep -m1 "model name" /proc/cpuinfo; cat /sys/fs/cgroup/cpu.max 2>/dev/null || true
for T in 16 8 32; do
  echo "== threads $T"
  F2Q_IO_THREADS=$T F2Q_GZ_TRACE=1 /tmp/reader_speed /tmp/x1.fastq.gz 2>&1 | grep -v "^\[pargz\]" | tail -3
  F2Q_IO_THREADS=$T F2Q_GZ_TRACE=1 /tmp/reader_speed /tmp/x1.fastq.gz 2>&1 | grep "^\[pargz\]" | sed 's/; \[0:.*//' | tail -6
done 2>&1 | tee $out/trace.txt
