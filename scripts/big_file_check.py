"""One-off check of 64-bit offsets: a FASTQ file larger than 4 GiB (plain, gzip with ISIZE wrap-around, BGZF) counted by
f2q_count_file must give what the device-generated block of the same synthetic stream gives."""
import importlib, os, sys, time, zlib, struct, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
pkg = importlib.import_module("2fast2q_amd")
n, part = 14_400_000, 1_200_000                       # 14.4 M x 313 B = 4.5 GB of text
guides = pkg.binding.synth_library(0xF2A5 + 3, 10000, 20)
d = tempfile.mkdtemp(dir=os.environ.get("TMPDIR", "/tmp"))
with pkg.Counter(features=guides, miss=1) as c:
    blk = c.synth_create(seed=77, n_reads=n, read_len=150)
    c.reset(); c.count_resident(blk); want = [list(x) for x in c.read_counts()]
    blk.free()
    plain, gz, bg = os.path.join(d, "big.fastq"), os.path.join(d, "big.fastq.gz"), os.path.join(d, "bigb.fastq.gz")
    t0 = time.time()
    co = zlib.compressobj(1, zlib.DEFLATED, -15); crc = 0; size = 0
    with open(plain, "wb") as fp, open(gz, "wb") as fg, open(bg, "wb") as fb:
        fg.write(b"\x1f\x8b\x08\x00\0\0\0\0\x00\x03")
        for lo in range(0, n, part):
            a = c.synth_fastq(lo, lo + part, seed=77, n_reads=n, read_len=150).tobytes()
            fp.write(a); fg.write(co.compress(a)); crc = zlib.crc32(a, crc); size += len(a)
            for o in range(0, len(a), 0xFF00):             # BGZF members
                p = a[o:o + 0xFF00]; cb = zlib.compressobj(1, zlib.DEFLATED, -15); body = cb.compress(p) + cb.flush()
                fb.write(b"\x1f\x8b\x08\x04\0\0\0\0\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", 25 + len(body)) + body + struct.pack("<II", zlib.crc32(p) & 0xFFFFFFFF, len(p)))
            print(f"wrote reads up to {lo + part} ({time.time() - t0:.0f} s)", flush=True)
        fg.write(co.flush() + struct.pack("<II", crc & 0xFFFFFFFF, size & 0xFFFFFFFF))
    print("text bytes", size, "> 4 GiB:", size > (1 << 32), flush=True)
    for path in (plain, gz, bg):
        c.reset(); t0 = time.time(); t, trunc = c.count_file(path); dt = time.time() - t0
        got = [list(x) for x in c.read_counts()]
        print(os.path.basename(path), "ok" if (got == want and not trunc and t["reads"] == n) else "MISMATCH", f"{n / dt / 1e6:.1f} Mreads/s", flush=True)
        os.remove(path)
