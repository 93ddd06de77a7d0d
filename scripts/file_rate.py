"""Wall-clock rate of f2q_count_file on a plain and a gzip FASTQ (F2Q_TRACE=1 prints the split)."""
import gzip, importlib, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("2fast2q_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
guides = pkg.binding.synth_library(0xF2A5 + 3, 10000, 20)
d = os.environ.get("KEEP_DIR") or tempfile.mkdtemp()
with pkg.Counter(features=guides, miss=1) as c:
    fq = bytes(c.synth_fastq(seed=1, n_reads=n, read_len=150))
    p = os.path.join(d, "x.fastq"); open(p, "wb").write(fq)
    c.count_block(fq[:1 << 20]); c.reset()
    for rep in range(2):
        c.reset(); t0 = time.perf_counter(); c.count_file(p); dt = time.perf_counter() - t0
        print(f"plain: {n/dt/1e6:.2f} Mreads/s wall ({len(fq)/dt/1e9:.2f} GB/s)", flush=True)
    pz = os.path.join(d, "x.fastq.gz")
    with gzip.open(pz, "wb", compresslevel=1) as f: f.write(fq[: len(fq) // 4])
    c.reset(); t0 = time.perf_counter(); c.count_file(pz); dt = time.perf_counter() - t0
    print(f"gzip : {n/4/dt/1e6:.2f} Mreads/s wall", flush=True)
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    from conftest import bgzf_bytes
    pb = os.path.join(d, "b.fastq.gz"); open(pb, "wb").write(bgzf_bytes(fq[: len(fq) // 2], level=1))
    for rep in range(2):
        c.reset(); t0 = time.perf_counter(); c.count_file(pb); dt = time.perf_counter() - t0
        print(f"bgzf : {n/2/dt/1e6:.2f} Mreads/s wall", flush=True)
