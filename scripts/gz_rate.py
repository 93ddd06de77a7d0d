"""f2q_count_file on the files scripts/file_rate.py left in $KEEP_DIR (gzip and bgzf only)."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("2fast2q_amd")
d = os.environ["KEEP_DIR"]
guides = pkg.binding.synth_library(0xF2A5 + 3, 10000, 20)
with pkg.Counter(features=guides, miss=1) as c:
    for name in ("x.fastq.gz", "b.fastq.gz", "x.fastq"):
        for rep in range(2):
            c.reset(); t0 = time.perf_counter(); t, _ = c.count_file(os.path.join(d, name)); dt = time.perf_counter() - t0
        print(f"{name}: {t['reads']/dt/1e6:.2f} Mreads/s wall", flush=True)
