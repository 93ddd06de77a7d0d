"""How many samples in flight (--cp) pay: the command line over a directory of small samples, plain and gzip, at --cp 1, 2, 4, 16.
usage: samples_threads.py [n_samples] [reads_per_sample]"""
import gzip, importlib, os, shutil, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("2fast2q_amd")
f2q = importlib.import_module("2fast2q_amd.fast2q")
n_s = int(sys.argv[1]) if len(sys.argv) > 1 else 16
n_r = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
guides = pkg.binding.synth_library(0xF2A5 + 3, 10000, 20)
d = tempfile.mkdtemp(prefix="f2q_samples_")
for sub in ("plain", "gz"):
    os.makedirs(os.path.join(d, sub))
with open(os.path.join(d, "lib.csv"), "w") as f:
    for i, g in enumerate(guides):
        f.write(f"g{i},{g}\n")
with pkg.Counter(features=guides, miss=1, phred=30, length=20, start="0") as c:
    for k in range(n_s):
        fq = bytes(c.synth_fastq(seed=100 + k, n_reads=n_r, read_len=150))
        open(os.path.join(d, "plain", f"s{k:02d}.fastq"), "wb").write(fq)
        with gzip.open(os.path.join(d, "gz", f"s{k:02d}.fastq.gz"), "wb", compresslevel=1) as z:
            z.write(fq)
        print("sample", k, flush=True)
for sub in ("plain", "gz"):
    for cp in (1, 2, 4, 16, 1):
        out = os.path.join(d, f"out_{sub}_{cp}")
        t0 = time.perf_counter()
        f2q.main(["-c", "--s", os.path.join(d, sub), "--g", os.path.join(d, "lib.csv"), "--o", out, "--m", "1", "--ph", "30", "--st", "0", "--l", "20", "--fn", "x", "--cp", str(cp)])
        dt = time.perf_counter() - t0
        print(f"{sub:5s} --cp {cp:2d}: {n_s} samples x {n_r} reads: {dt:.2f} s whole run ({n_s * n_r / dt / 1e6:.1f} Mreads/s)", flush=True)
        shutil.rmtree(out, ignore_errors=True)
