"""cProfile of a whole command-line run over many small samples (where the harness spends its time besides counting)"""
import cProfile, importlib, os, pstats, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("2fast2q_amd")
f2q = importlib.import_module("2fast2q_amd.fast2q")
n_s, n_r, n_g = 12, 1_000_000, 10000
guides = pkg.binding.synth_library(0xF2A5 + 3, n_g, 20)
d = tempfile.mkdtemp(prefix="f2q_samples_")
os.makedirs(os.path.join(d, "fq"))
with open(os.path.join(d, "lib.csv"), "w") as f:
    for i, g in enumerate(guides):
        f.write(f"g{i},{g}\n")
with pkg.Counter(features=guides, miss=1, phred=30, length=20, start="0") as c:
    for k in range(n_s):
        open(os.path.join(d, "fq", f"s{k:02d}.fastq"), "wb").write(bytes(c.synth_fastq(seed=100 + k, n_reads=n_r, read_len=150)))
print("samples written", flush=True)
args = ["-c", "--s", os.path.join(d, "fq"), "--g", os.path.join(d, "lib.csv"), "--o", os.path.join(d, "out"), "--m", "1", "--ph", "30", "--st", "0", "--l", "20", "--fn", "x", "--cp", "1"]
f2q.main(list(args))                      # warm
pr = cProfile.Profile(); pr.enable(); f2q.main(list(args)); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
