"""PCIe-inclusive rates of the host entry points (FASTQ text on the host -> counts): not the bench metric."""
import gzip, importlib, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("2fast2q_amd")
n = 4_000_000
guides = pkg.binding.synth_library(0xF2A5 + 3, 10000, 20)
with pkg.Counter(features=guides, miss=1) as c:
    fq = bytes(c.synth_fastq(seed=1, n_reads=n, read_len=150))
    c.count_block(fq[:1 << 20])
    c.reset()
    t0 = time.perf_counter(); used, t = c.count_block(fq, want_timing=True); dt = time.perf_counter() - t0
    print(f"f2q_count_block fixed: {n/dt/1e6:.2f} Mreads/s wall ({len(fq)/dt/1e9:.2f} GB/s of FASTQ text), kernels {t['kernel_ms']:.2f} ms")
    d = tempfile.mkdtemp()
    p = os.path.join(d, "x.fastq"); open(p, "wb").write(fq)
    c.reset(); t0 = time.perf_counter(); c.count_file(p); dt = time.perf_counter() - t0
    print(f"f2q_count_file plain: {n/dt/1e6:.2f} Mreads/s wall")
    pz = os.path.join(d, "x.fastq.gz")
    with gzip.open(pz, "wb", compresslevel=1) as f: f.write(fq[: len(fq) // 4])
    c.reset(); t0 = time.perf_counter(); c.count_file(pz); dt = time.perf_counter() - t0
    print(f"f2q_count_file gzip : {n/4/dt/1e6:.2f} Mreads/s wall")
UP, DOWN = "GTTTAAGAGCTA", "CGTTACCAGGTT"
with pkg.Counter(features=guides, miss=1, upstream=UP, downstream=DOWN, miss_search_up=1, miss_search_down=1) as c:
    fq = bytes(c.synth_fastq(seed=1, n_reads=n, read_len=150, cassette=True, up=UP, down=DOWN))
    c.count_block(fq[:1 << 20]); c.reset()
    t0 = time.perf_counter(); used, t = c.count_block(fq, want_timing=True); dt = time.perf_counter() - t0
    print(f"f2q_count_block anchored: {n/dt/1e6:.2f} Mreads/s wall, kernels {t['kernel_ms']:.2f} ms")
