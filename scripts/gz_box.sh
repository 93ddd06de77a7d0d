# usage: gz_box.sh <tag>   ordinary .gz on the GPU box: the host reader alone (decoded MB/s, sequential vs the worker pool
#   at several thread counts / chunk sizes), then f2q_count_file end to end (Mreads/s) on a gzip -1 and a gzip -6 file
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; out=gpurun_out/$tag; mkdir -p $out
g++ -O2 -std=c++17 -o /tmp/reader_speed scripts/reader_speed.cpp -lz -lpthread
python - > $out/make.txt 2>&1 <<PY
import importlib, gzip, os, time, subprocess
pkg = importlib.import_module("2fast2q_amd")
guides = pkg.binding.synth_library(0xF2A5 + 3, 10000, 20)
with pkg.Counter(features=guides, miss=1, phred=30, length=20, start="0") as c:
    fq = bytes(c.synth_fastq(seed=0xBEEF, n_reads=4_000_000, read_len=150))
open("/tmp/x.fastq", "wb").write(fq)
for lvl in (1, 6):
    t0 = time.time()
    subprocess.check_call(f"gzip -{lvl} -c /tmp/x.fastq > /tmp/x{lvl}.fastq.gz", shell=True)
    print("gzip", lvl, os.path.getsize(f"/tmp/x{lvl}.fastq.gz"), "bytes", round(time.time() - t0, 1), "s", flush=True)
PY
cat $out/make.txt
nproc
for f in /tmp/x1.fastq.gz /tmp/x6.fastq.gz; do
  echo "== $f sequential"; F2Q_GZ_PAR=0 /tmp/reader_speed $f | tail -1
  for T in 4 8 16; do for kb in 1024 2048 4096; do
    echo "== $f threads $T chunk ${kb}KB"; F2Q_IO_THREADS=$T F2Q_GZ_CHUNK_KB=$kb /tmp/reader_speed $f | tail -1
  done; done
done 2>&1 | tee $out/reader.txt
python - 2>&1 <<PY | tee $out/count_file.txt
import importlib, os, time
pkg = importlib.import_module("2fast2q_amd")
guides = pkg.binding.synth_library(0xF2A5 + 3, 10000, 20)
with pkg.Counter(features=guides, miss=1, phred=30, length=20, start="0") as c:
    for name in ("/tmp/x1.fastq.gz", "/tmp/x6.fastq.gz", "/tmp/x.fastq"):
        for env in ({}, {"F2Q_GZ_PAR": "0"}):
            if env and not name.endswith(".gz"): continue
            os.environ.pop("F2Q_GZ_PAR", None); os.environ.update(env)
            for rep in range(3):
                c.reset(); t0 = time.perf_counter(); t, _ = c.count_file(name); dt = time.perf_counter() - t0
            print(f"{name} {env}: {t['reads'] / dt / 1e6:.2f} Mreads/s wall ({t['reads']} reads)", flush=True)
PY
