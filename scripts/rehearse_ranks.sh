# two ranks of the command line on ONE GPU (gloo for the reduction, both contexts on device 0): the compiled table
# must equal the single-process one.  A rehearsal of the multi-process launch of INTEGRATION.md, not a benchmark.
# Every step has a hard timeout and writes under gpurun_out/rehearse/.
cd $GRAFT_REPO_ROOT
O=gpurun_out/rehearse; rm -rf $O; mkdir -p $O
W=$(mktemp -d)
python - "$W" <<'PY'
import sys, os, gzip
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import synth
from conftest import bgzf_bytes
w = sys.argv[1]
guides = synth.make_library(300, 20, 9)
os.makedirs(w + "/in")
open(w + "/g.csv", "w").write("".join(f"g{i},{g}\n" for i, g in enumerate(guides)))
for k, (n, kind) in enumerate([(40000, "plain"), (30000, "gzip"), (20000, "bgzf")]):
    fq = synth.make_fastq(synth.Spec(seed=100 + k, n_reads=n, read_len=120), guides)
    p = f"{w}/in/s{k}.fastq" + ("" if kind == "plain" else ".gz")
    open(p, "wb").write(fq if kind == "plain" else gzip.compress(fq, 1) if kind == "gzip" else bgzf_bytes(fq))
PY
echo "inputs ready" | tee $O/progress.txt
export F2Q_FILE_CHUNK=300000 PYTHONFAULTHANDLER=1 GLOO_SOCKET_IFNAME=lo
timeout -s ABRT -k 5 90 python -m 2fast2q_amd -c --s $W/in --g $W/g.csv --o $W/one --fn one --m 1 --pb > $O/one.log 2>&1
echo "single process rc=$?" | tee -a $O/progress.txt
for r in 0 1; do
  RANK=$r LOCAL_RANK=$r WORLD_SIZE=2 MASTER_ADDR=127.0.0.1 MASTER_PORT=29521 F2Q_DEVICE=0 F2Q_DIST_BACKEND=gloo \
    timeout -s ABRT -k 5 90 python -m 2fast2q_amd -c --s $W/in --g $W/g.csv --o $W/two --fn two --m 1 --pb > $O/rank$r.log 2>&1 &
done
wait
echo "two ranks done" | tee -a $O/progress.txt
A=$(find $W/one -name "one.csv" | head -1); B=$(find $W/two -name "two.csv" | head -1)
if [ -n "$A" ] && [ -n "$B" ] && cmp $A $B; then echo "compiled tables identical: $(wc -l < $A) rows" | tee -a $O/progress.txt; else echo "compiled tables differ or missing: [$A] [$B]" | tee -a $O/progress.txt; tail -30 $O/rank0.log; tail -30 $O/rank1.log; fi
for s in s0 s1 s2; do a=$(find $W/one -name "${s}_reads.csv" | head -1); b=$(find $W/two -name "${s}_reads.csv" | head -1); [ -n "$a" ] && [ -n "$b" ] && tail -n +2 $a | cmp - <(tail -n +2 $b) && echo "$s reads table identical" | tee -a $O/progress.txt; done
true
# the same two ranks on one large plain file: every rank reads only its own pieces (f2q_count_pieces); wall time of
# one process against two on the ONE GPU (a functional rehearsal: two ranks share one PCIe link and one device here)
python - "$W" <<'PY'
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import importlib
pkg = importlib.import_module("2fast2q_amd")
w = sys.argv[1]
import os
os.makedirs(w + "/big")
guides = [l.split(",")[1].strip() for l in open(w + "/g.csv")]
with pkg.Counter(features=guides, miss=1) as c:
    fq = bytes(c.synth_fastq(seed=5, n_reads=3_000_000, read_len=150))
open(w + "/big/big.fastq", "wb").write(fq)
PY
export F2Q_PIECE_BYTES=67108864
unset F2Q_FILE_CHUNK
t0=$(date +%s%N)
timeout -s ABRT -k 5 120 python -m 2fast2q_amd -c --s $W/big --g $W/g.csv --o $W/bone --fn one --m 1 --pb > $O/big_one.log 2>&1
t1=$(date +%s%N); echo "one process: $(( (t1 - t0) / 1000000 )) ms wall (python start-up included)" > $O/time_one.txt
for r in 0 1; do
  RANK=$r LOCAL_RANK=$r WORLD_SIZE=2 MASTER_ADDR=127.0.0.1 MASTER_PORT=29522 F2Q_DEVICE=0 F2Q_DIST_BACKEND=gloo \
    timeout -s ABRT -k 5 120 python -m 2fast2q_amd -c --s $W/big --g $W/g.csv --o $W/btwo --fn two --m 1 --pb > $O/big_rank$r.log 2>&1 &
done
t0=$(date +%s%N); wait; t1=$(date +%s%N)
echo "two ranks: $(( (t1 - t0) / 1000000 )) ms wall (python start-up, torch import and gloo rendezvous included)" > $O/time_two.txt
cat $O/time_one.txt $O/time_two.txt | tee -a $O/progress.txt
grep -h "Sample big was processed" $O/big_one.log $O/big_rank0.log $O/big_rank1.log | tee -a $O/progress.txt
A=$(find $W/bone -name "one.csv" | head -1); B=$(find $W/btwo -name "two.csv" | head -1)
if [ -n "$A" ] && [ -n "$B" ] && cmp $A $B; then echo "big file: compiled tables identical ($(wc -l < $A) rows), each rank read only its own pieces" | tee -a $O/progress.txt; else echo "big file: tables differ or missing [$A] [$B]" | tee -a $O/progress.txt; tail -20 $O/big_rank0.log; fi
true
# ... and on a BGZF file of the same reads: every rank inflates only its own runs of members (census + count)
python - "$W" <<'PY'
import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from conftest import bgzf_bytes
w = sys.argv[1]
os.makedirs(w + "/bz")
fq = open(w + "/big/big.fastq", "rb").read()[: 400 << 20]
fq = fq[: fq.rfind(b"\n@") + 1]
open(w + "/bz/bz.fastq.gz", "wb").write(bgzf_bytes(fq, level=1))
PY
export F2Q_PIECE_BYTES=33554432
t0=$(date +%s%N)
timeout -s ABRT -k 5 200 python -m 2fast2q_amd -c --s $W/bz --g $W/g.csv --o $W/zone --fn one --m 1 --pb > $O/bz_one.log 2>&1
t1=$(date +%s%N); echo "BGZF, one process: $(( (t1 - t0) / 1000000 )) ms wall" > $O/time_bz_one.txt
for r in 0 1; do
  RANK=$r LOCAL_RANK=$r WORLD_SIZE=2 MASTER_ADDR=127.0.0.1 MASTER_PORT=29523 F2Q_DEVICE=0 F2Q_DIST_BACKEND=gloo \
    timeout -s ABRT -k 5 200 python -m 2fast2q_amd -c --s $W/bz --g $W/g.csv --o $W/ztwo --fn two --m 1 --pb > $O/bz_rank$r.log 2>&1 &
done
t0=$(date +%s%N); wait; t1=$(date +%s%N)
echo "BGZF, two ranks: $(( (t1 - t0) / 1000000 )) ms wall" > $O/time_bz_two.txt
cat $O/time_bz_one.txt $O/time_bz_two.txt | tee -a $O/progress.txt
grep -h "Sample bz was processed" $O/bz_one.log $O/bz_rank0.log $O/bz_rank1.log | tee -a $O/progress.txt
A=$(find $W/zone -name "one.csv" | head -1); B=$(find $W/ztwo -name "two.csv" | head -1)
if [ -n "$A" ] && [ -n "$B" ] && cmp $A $B; then echo "BGZF file: compiled tables identical ($(wc -l < $A) rows), each rank inflated only its own runs of members" | tee -a $O/progress.txt; else echo "BGZF file: tables differ or missing [$A] [$B]" | tee -a $O/progress.txt; tail -20 $O/bz_rank0.log; fi
true
# Extract+Count (--mo EC --us/--ds) as two ranks: the ranks' key tables are merged by key (counts summed, first read = min),
# the compiled table must equal the single-process one; the BGZF file of cassette reads goes by runs of members
python - "$W" <<'PY'
import sys, os, importlib
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from conftest import bgzf_bytes
pkg = importlib.import_module("2fast2q_amd")
w = sys.argv[1]
os.makedirs(w + "/ec")
guides = [l.split(",")[1].strip() for l in open(w + "/g.csv")]
UP, DOWN = "GTTTAAGAGCTA", "CGTTACCAGGTT"
with pkg.Counter(features=guides, miss=1) as c:
    fq = bytes(c.synth_fastq(seed=8, n_reads=1_200_000, read_len=150, cassette=True, up=UP, down=DOWN, max_offset=100))
open(w + "/ec/ecp.fastq", "wb").write(fq[: len(fq) // 2][: fq[: len(fq) // 2].rfind(b"\n@") + 1])
open(w + "/ec/ecz.fastq.gz", "wb").write(bgzf_bytes(fq[len(fq) // 2:][fq[len(fq) // 2:].find(b"\n@") + 1:], level=1))
PY
export F2Q_PIECE_BYTES=16777216 F2Q_HOT_LEARN=65536
timeout -s ABRT -k 5 200 python -m 2fast2q_amd -c --s $W/ec --o $W/eone --fn one --mo EC --us GTTTAAGAGCTA --ds CGTTACCAGGTT --msu 1 --msd 1 --pb > $O/ec_one.log 2>&1
for r in 0 1; do
  RANK=$r LOCAL_RANK=$r WORLD_SIZE=2 MASTER_ADDR=127.0.0.1 MASTER_PORT=29524 F2Q_DEVICE=0 F2Q_DIST_BACKEND=gloo \
    timeout -s ABRT -k 5 200 python -m 2fast2q_amd -c --s $W/ec --o $W/etwo --fn two --mo EC --us GTTTAAGAGCTA --ds CGTTACCAGGTT --msu 1 --msd 1 --pb > $O/ec_rank$r.log 2>&1 &
done
wait
A=$(find $W/eone -name "one.csv" | head -1); B=$(find $W/etwo -name "two.csv" | head -1)
if [ -n "$A" ] && [ -n "$B" ] && cmp $A $B; then echo "Extract+Count: compiled tables identical ($(wc -l < $A) rows: keys in first-seen order), plain file by pieces + BGZF file by runs of members, hot keys in LDS on both ranks" | tee -a $O/progress.txt; else echo "Extract+Count: tables differ or missing [$A] [$B]" | tee -a $O/progress.txt; tail -20 $O/ec_rank0.log; tail -5 $O/ec_one.log; fi
true
