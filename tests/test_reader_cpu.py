"""The file reader behind f2q_count_file (2fast2q_amd/csrc/f2q_reader.h) against Python's own gzip/open — the
decoders the reference uses (fast2q.py:565-569).  Host only."""
import gzip, os, random, zlib
import pytest
from conftest import bgzf_bytes
from emu_helper import read_file


def text(n, seed=1):
    rng = random.Random(seed)
    rec = []
    for i in range(n):
        L = rng.randint(20, 160)
        rec.append("@r%d\n%s\n+\n%s\n" % (i, "".join(rng.choice("ACGTN") for _ in range(L)), "I" * L))
    return "".join(rec).encode()


@pytest.fixture(scope="module")
def payload():
    return text(12000)


@pytest.mark.parametrize("piece", [4096, 1 << 16, 100_003, 1 << 22])
@pytest.mark.parametrize("threads", [1, 4])
def test_plain(tmp_path, payload, piece, threads):
    p = tmp_path / "a.fastq"; p.write_bytes(payload)
    got, tr, kind = read_file(p, piece, threads)
    assert (got, tr, kind) == (payload, False, "plain")


def test_plain_parallel_slices(tmp_path):
    data = os.urandom(40 << 20).replace(b"\x1f\x8b", b"ab")
    p = tmp_path / "big.fastq"; p.write_bytes(data)
    got, tr, kind = read_file(p, 24 << 20, 5, out_cap=48 << 20)
    assert kind == "plain" and not tr and got == data


def test_empty_files(tmp_path):
    p = tmp_path / "e.fastq"; p.write_bytes(b"")
    assert read_file(p) == (b"", False, "plain")
    z = tmp_path / "e.fastq.gz"
    with gzip.open(z, "wb") as f: pass
    got, tr, kind = read_file(z)
    assert (got, tr) == (b"", False)
    b = tmp_path / "e2.fastq.gz"; b.write_bytes(bgzf_bytes(b""))
    assert read_file(b) == (b"", False, "bgzf")


@pytest.mark.parametrize("piece", [4096, 1 << 16, 1 << 22])
def test_gzip_single_and_multi_member(tmp_path, payload, piece):
    z = tmp_path / "a.fastq.gz"
    with gzip.open(z, "wb", compresslevel=1) as f: f.write(payload)
    assert read_file(z, piece) == (payload, False, "gzip")
    half = len(payload) // 2
    z2 = tmp_path / "b.fastq.gz"; z2.write_bytes(gzip.compress(payload[:half]) + gzip.compress(payload[half:]))
    assert gzip.open(z2).read() == payload
    assert read_file(z2, piece) == (payload, False, "gzip")


@pytest.fixture(params=["mapped", "pread"])
def bgzf_io(request, monkeypatch):
    """BGZF files are memory-mapped (the workers fault their own pages in); F2Q_NO_MMAP=1 keeps the buffered pread path"""
    if request.param == "pread":
        monkeypatch.setenv("F2Q_NO_MMAP", "1")
    return request.param


@pytest.mark.parametrize("piece", [4096, 1 << 16, 300_000, 1 << 22])
@pytest.mark.parametrize("threads", [1, 3, 8])
def test_bgzf(tmp_path, payload, piece, threads, bgzf_io):
    b = tmp_path / "a.fastq.gz"; b.write_bytes(bgzf_bytes(payload))
    assert gzip.open(b).read() == payload                      # what the reference would see
    assert read_file(b, piece, threads) == (payload, False, "bgzf")


def test_bgzf_variants(tmp_path, payload, bgzf_io):
    for k, kw in enumerate([dict(block=1000), dict(block=65280, level=1), dict(eof_marker=False), dict(extra_subfield=True, block=5000),
                            dict(block=17, level=0)]):
        data = payload if kw.get("block", 0) != 17 else payload[:20000]
        b = tmp_path / f"v{k}.gz"; b.write_bytes(bgzf_bytes(data, **kw))
        assert gzip.open(b).read() == data
        assert read_file(b, 1 << 16, 4) == (data, False, "bgzf"), kw


def test_bgzf_empty_members_inside(tmp_path, payload, bgzf_io):
    raw = bgzf_bytes(payload[:5000], eof_marker=True) + bgzf_bytes(payload[5000:9000], eof_marker=True) + bgzf_bytes(payload[9000:])
    b = tmp_path / "m.gz"; b.write_bytes(raw)
    assert gzip.open(b).read() == payload
    assert read_file(b, 8192, 4) == (payload, False, "bgzf")


@pytest.mark.parametrize("cut", [10, 30, 1000, 70_000, -9, -1])
def test_bgzf_cut_off(tmp_path, payload, cut, bgzf_io):
    raw = bgzf_bytes(payload, eof_marker=False)
    raw = raw[:cut]
    b = tmp_path / "t.gz"; b.write_bytes(raw)
    with pytest.raises((EOFError, gzip.BadGzipFile, zlib.error)):
        gzip.open(b).read()
    got, tr, kind = read_file(b, 1 << 16, 4)
    assert tr and payload.startswith(got)
    # every member that is whole is delivered
    whole = 0
    for i in range(0, len(payload), 0xFF00):
        if len(bgzf_bytes(payload[:i + 0xFF00], eof_marker=False)) <= len(raw): whole = min(i + 0xFF00, len(payload))
    assert len(got) == whole


def test_bgzf_corrupt_member(tmp_path, payload, bgzf_io):
    raw = bytearray(bgzf_bytes(payload, block=4000))
    raw[len(raw) // 2] ^= 0x55
    b = tmp_path / "c.gz"; b.write_bytes(bytes(raw))
    got, tr, kind = read_file(b, 1 << 16, 4)
    assert tr and kind == "bgzf" and payload.startswith(got) and len(got) < len(payload)


def test_bgzf_then_ordinary_gzip(tmp_path, payload, bgzf_io):
    a, c = payload[:100_000], payload[100_000:]
    b = tmp_path / "mix.gz"; b.write_bytes(bgzf_bytes(a, eof_marker=False) + gzip.compress(c))
    assert gzip.open(b).read() == payload
    got, tr, kind = read_file(b, 1 << 16, 4)
    assert (got, tr, kind) == (payload, False, "gzip")


def test_gzip_cut_off(tmp_path, payload):
    z = tmp_path / "t.fastq.gz"; z.write_bytes(gzip.compress(payload)[:-2000])
    got, tr, kind = read_file(z)
    assert tr and kind == "gzip" and payload.startswith(got)


def test_gz_content_sniffed_not_named(tmp_path, payload):
    z = tmp_path / "noext"; z.write_bytes(gzip.compress(payload))
    assert read_file(z)[0] == payload


def gz_member(data, name=None, comment=None, extra=None, hcrc=False, level=6):
    """one gzip member with optional header fields (RFC 1952)"""
    flg = (4 if extra is not None else 0) | (8 if name is not None else 0) | (16 if comment is not None else 0) | (2 if hcrc else 0)
    head = b"\x1f\x8b\x08" + bytes([flg]) + b"\0\0\0\0\x00\x03"
    if extra is not None: head += len(extra).to_bytes(2, "little") + extra
    if name is not None: head += name + b"\0"
    if comment is not None: head += comment + b"\0"
    if hcrc: head += (zlib.crc32(head) & 0xFFFF).to_bytes(2, "little")
    co = zlib.compressobj(level, zlib.DEFLATED, -15)
    body = co.compress(data) + co.flush()
    return head + body + (zlib.crc32(data) & 0xFFFFFFFF).to_bytes(4, "little") + (len(data) & 0xFFFFFFFF).to_bytes(4, "little")


def test_gzip_header_fields_padding_and_garbage(tmp_path, payload):
    a, b, c = payload[:70_000], payload[70_000:70_001], payload[70_001:]
    raw = gz_member(a, name=b"x.fastq", comment=b"hello", extra=b"ABCDEFG", hcrc=True) + gz_member(b, level=0) + b"\0" * 37 + gz_member(c, name=b"y")
    z = tmp_path / "h.gz"; z.write_bytes(raw)
    assert gzip.open(z).read() == payload
    for piece in (4096, 70_000, 1 << 20):
        assert read_file(z, piece, 3) == (payload, False, "gzip")
    z2 = tmp_path / "g.gz"; z2.write_bytes(raw + b"\0" * 5 + b"garbage that is not gzip")
    assert read_file(z2, 1 << 16, 2) == (payload, False, "gzip")            # ignored, like zlib's gzread
    z3 = tmp_path / "e.gz"; z3.write_bytes(gz_member(b"") + gz_member(payload[:10]) + gz_member(b""))
    assert read_file(z3) == (payload[:10], False, "gzip")


def test_gzip_wrong_crc_or_size(tmp_path, payload):
    good = gz_member(payload[:50_000])
    for k in (-8, -4):
        raw = bytearray(good); raw[k] ^= 1
        z = tmp_path / f"bad{-k}.gz"; z.write_bytes(bytes(raw) + gz_member(payload[50_000:60_000]))
        got, tr, kind = read_file(z, 1 << 16, 2)
        assert tr and got == payload[:50_000]                                  # the text is delivered, then the damage reported


def test_gzip_large_parallel_crc(tmp_path):
    data = text(90_000, seed=5)                                                # ~16 MB: several pieces, sliced CRC
    assert len(data) > (12 << 20)
    z = tmp_path / "big.gz"; z.write_bytes(gz_member(data[: 9 << 20], level=1) + gz_member(data[9 << 20:], level=1))
    for piece, th in ((6 << 20, 4), (32 << 20, 8), (1 << 20, 1)):
        got, tr, kind = read_file(z, piece, th, out_cap=20 << 20)
        assert (got == data, tr, kind) == (True, False, "gzip")
    # the member boundary exactly on a piece boundary
    got, tr, kind = read_file(z, 3 << 20, 4, out_cap=20 << 20)
    assert got == data and not tr


def test_crc32_slicing_matches_zlib():
    from emu_helper import lib
    rng = random.Random(7)
    for n in [0, 1, 2, 15, 16, 17, 31, 32, 33, 255, 1000, 65537, 1 << 20]:
        data = bytes(rng.getrandbits(8) for _ in range(min(n, 70000))) * (1 if n <= 70000 else n // 70000 + 1)
        data = data[:n]
        for off in (0, 1, 3, 7):
            d = data[off:]
            assert lib().emu_crc32(0, d, len(d)) == (zlib.crc32(d) & 0xFFFFFFFF)
            half = len(d) // 3
            assert lib().emu_crc32(lib().emu_crc32(0, d[:half], half), d[half:], len(d) - half) == (zlib.crc32(d) & 0xFFFFFFFF)


def test_cut_off_gzip_delivers_what_the_reference_reads(tmp_path):
    """tests/golden/truncated_gzip_cases.json: archives cut at several positions and what the REFERENCE counted from them
    (it keeps every complete record before readline raises, fast2q.py:405-407).  The product's reader must deliver
    exactly as many complete records before reporting the damage."""
    import base64
    import json
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "truncated_gzip_cases.json")) as f:
        cases = json.load(f)["cases"]
    assert len(cases) == 8
    for c in cases:
        path = tmp_path / f"cut_{c['level']}_{c['frac']}.fastq.gz"
        path.write_bytes(base64.b64decode(c["gz_b64"]))
        for piece in (1 << 12, 1 << 20):
            text, truncated, kind = read_file(str(path), piece=piece)
            assert truncated and kind == "gzip"
            assert text.count(b"\n") // 4 == c["expected"]["stats"][0], (c["level"], c["frac"], piece)


# ---- ordinary gzip decoded by the worker pool (f2q_pargz.h) ------------------------------------------------------
def big_text(n, seed):
    """FASTQ-like text with varied qualities and a few bytes >= 128 (markers must carry any byte)"""
    rng = random.Random(seed)
    rec = []
    for i in range(n):
        L = rng.randint(30, 151)
        q = "".join(rng.choice("IIIIIIII?5+#FF:,") for _ in range(L))
        rec.append("@inst:%d:%d %d/1\n%s\n+\n%s\n" % (seed, i, rng.randint(0, 9), "".join(rng.choice("ACGT") for _ in range(L)), q))
    b = bytearray("".join(rec).encode())
    for _ in range(50):
        b[rng.randrange(len(b))] = rng.choice([0x80, 0xFF, 0x00, 0xC3])
    return bytes(b)


@pytest.fixture(scope="module")
def par_payload():
    return big_text(40000, 7)


@pytest.fixture()
def par_env(monkeypatch):
    monkeypatch.setenv("F2Q_GZ_PAR_MIN_KB", "16")           # members of 16 KiB and more go to the pool ...
    monkeypatch.setenv("F2Q_GZ_CHUNK_KB", "64")             # ... in chunks of 64 KiB: many chunks and rounds on a few MB
    return monkeypatch


@pytest.mark.parametrize("level", [1, 6, 9])
@pytest.mark.parametrize("threads", [2, 3, 8])
def test_gzip_parallel_equals_gzip(tmp_path, par_payload, par_env, level, threads):
    z = tmp_path / "a.fastq.gz"
    with gzip.open(z, "wb", compresslevel=level) as f: f.write(par_payload)
    for piece in (1 << 16, 1 << 20, 1 << 24):               # smaller than a round (staged text), and larger (straight into the piece)
        assert read_file(z, piece, threads, out_cap=1 << 25) == (par_payload, False, "gzip")
    par_env.setenv("F2Q_GZ_PAR", "0")                        # the sequential decoder on the same file
    assert read_file(z, 1 << 20, threads, out_cap=1 << 25) == (par_payload, False, "gzip")


def test_gzip_parallel_members_and_block_types(tmp_path, par_payload, par_env):
    """several members (one of them small, one empty), stored blocks (level 0), fixed-Huffman blocks (tiny inputs with
    Z_FIXED) and incompressible bytes between the text: the search only looks for dynamic blocks, the chunks in front
    of a stretch without one simply run through it"""
    third = len(par_payload) // 3
    noise = os.urandom(300_000)
    co = zlib.compressobj(6, zlib.DEFLATED, 31, 8, zlib.Z_FIXED)
    fixed = co.compress(par_payload[:50_000]) + co.flush()
    parts = [gzip.compress(par_payload[:third], 6), gzip.compress(b""), gzip.compress(noise, 0), fixed,
             gzip.compress(par_payload[third:third + 2000], 9), gzip.compress(par_payload[third + 2000:] + noise + par_payload[:third], 1)]
    want = par_payload[:third] + noise + par_payload[:50_000] + par_payload[third:] + noise + par_payload[:third]
    z = tmp_path / "m.fastq.gz"; z.write_bytes(b"".join(parts))
    assert gzip.open(z).read() == want
    for threads in (2, 5):
        assert read_file(z, 1 << 18, threads, out_cap=1 << 26) == (want, False, "gzip")


@pytest.mark.parametrize("cut", [0.97, 0.5, 0.13])
def test_gzip_parallel_cut_off_archive(tmp_path, par_payload, par_env, cut):
    """a cut-off or damaged archive delivers exactly the bytes the sequential decoder delivers (every byte decoded before
    the damage), and reports truncation"""
    raw = gzip.compress(par_payload, 6)
    z = tmp_path / "c.fastq.gz"; z.write_bytes(raw[: int(len(raw) * cut)])
    got = read_file(z, 1 << 18, 4, out_cap=1 << 25)
    par_env.setenv("F2Q_GZ_PAR", "0")
    seq = read_file(z, 1 << 18, 4, out_cap=1 << 25)
    assert got == seq and got[1] is True and par_payload.startswith(got[0]) and len(got[0]) > 0.9 * cut * len(par_payload) - 70000
    par_env.delenv("F2Q_GZ_PAR")
    # damage in the middle: a flipped byte
    bad = bytearray(raw); bad[len(raw) // 2] ^= 0x55
    z.write_bytes(bytes(bad))
    got = read_file(z, 1 << 18, 4, out_cap=1 << 25)
    par_env.setenv("F2Q_GZ_PAR", "0")
    seq = read_file(z, 1 << 18, 4, out_cap=1 << 25)
    assert got[1] is True and seq[1] is True and got[0] == seq[0]


def test_gzip_parallel_wrong_crc_is_truncation(tmp_path, par_payload, par_env):
    raw = bytearray(gzip.compress(par_payload, 6))
    raw[-6] ^= 1                                             # the CRC-32 in the trailer
    z = tmp_path / "w.fastq.gz"; z.write_bytes(bytes(raw))
    got = read_file(z, 1 << 20, 4, out_cap=1 << 25)
    assert got[0] == par_payload and got[1] is True
