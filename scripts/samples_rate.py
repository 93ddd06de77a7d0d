"""Wall clock of a whole run over many small samples (the command line's usual shape: one library, a directory of FASTQ
files): contexts kept between samples against F2Q_NO_CTX_CACHE=1 (a fresh context -- library index, pinned staging
buffers -- per sample).  usage: samples_rate.py [n_samples] [reads_per_sample] [n_guides]"""
import importlib, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("2fast2q_amd")
f2q = importlib.import_module("2fast2q_amd.fast2q")
n_s = int(sys.argv[1]) if len(sys.argv) > 1 else 24
n_r = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
n_g = int(sys.argv[3]) if len(sys.argv) > 3 else 10000
guides = pkg.binding.synth_library(0xF2A5 + 3, n_g, 20)
d = tempfile.mkdtemp(prefix="f2q_samples_")
os.makedirs(os.path.join(d, "fq"))
with open(os.path.join(d, "lib.csv"), "w") as f:
    for i, g in enumerate(guides):
        f.write(f"g{i},{g}\n")
with pkg.Counter(features=guides, miss=1, phred=30, length=20, start="0") as c:
    for k in range(n_s):
        open(os.path.join(d, "fq", f"s{k:02d}.fastq"), "wb").write(bytes(c.synth_fastq(seed=100 + k, n_reads=n_r, read_len=150)))
for env in ({}, {"F2Q_NO_CTX_CACHE": "1"}, {}):
    os.environ.pop("F2Q_NO_CTX_CACHE", None); os.environ.update(env)
    out = os.path.join(d, "out_" + ("fresh" if env else "kept"))
    t0 = time.perf_counter()
    f2q.main(["-c", "--s", os.path.join(d, "fq"), "--g", os.path.join(d, "lib.csv"), "--o", out, "--m", "1", "--ph", "30", "--st", "0", "--l", "20", "--fn", "x"])
    dt = time.perf_counter() - t0
    print(f"{'fresh context per sample' if env else 'contexts kept':26s}: {n_s} samples x {n_r} reads, {n_g} guides: {dt:.2f} s  ({n_s * n_r / dt / 1e6:.1f} Mreads/s whole run)", flush=True)
