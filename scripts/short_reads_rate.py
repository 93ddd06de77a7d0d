"""kernel time of the fixed-offset path when a share of the reads is shorter than start + length (the clipped-window case)"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("2fast2q_amd")
n = 6_000_000
guides = pkg.binding.synth_library(0xF2A5 + 3, 10000, 20)
with pkg.Counter(features=guides, miss=1, start="30") as c:
    fq = bytes(c.synth_fastq(seed=1, n_reads=n, read_len=150, start=30))
    for share in (0.0, 0.1, 0.5):
        if share == 0.0:
            data = fq
        else:
            # cut a share of the reads to 40 bases (shorter than start + length = 50): rebuild those records
            lines = fq.split(b"\n")
            step = int(1 / share)
            for i in range(0, n, step):
                lines[4 * i + 1] = lines[4 * i + 1][:40]; lines[4 * i + 3] = lines[4 * i + 3][:40]
            data = b"\n".join(lines)
        blk = c.block_from_fastq(data)             # < 2 GiB of text: one resident block
        c.reset(); c.count_resident(blk); c.reset()
        t = c.count_resident(blk)
        _, stats = c.read_counts()
        print(f"short share {share}: kernel {t['kernel_ms']:.3f} ms for {t['reads']} reads ({t['reads']/t['kernel_ms']/1e6:.2f} Greads/s), general {t['general_reads']}, stats {[int(x) for x in stats]}", flush=True)
        blk.free()
