"""The product's device logic (2fast2q_amd/csrc/f2q_device.h + f2q_host.h) executed lane by lane
on the host (tests/emu) against the golden vectors captured from the reference and against the
oracle.  These are the same functions the HIP kernels call per lane; the GPU parity tests
(test_gpu_parity.py) check the kernels themselves."""
import pytest

import synth
from conftest import case_fastq, load_cases, loader_view, sprinkle_symbols
from emu_helper import Emu
from oracle import oracle as O

CASES = load_cases()


def params_of(case):
    p = case["params"]
    return dict(mode=p.get("Running Mode", "C"), miss=p.get("miss", 1), phred=p.get("phred", 30),
                length=p.get("length", 20), start=p.get("start", "0"), upstream=p.get("upstream"),
                downstream=p.get("downstream"), miss_search_up=p.get("miss_search_up", 0),
                miss_search_down=p.get("miss_search_down", 0), qual_up=p.get("qual_up", 30),
                qual_down=p.get("qual_down", 30))


@pytest.mark.parametrize("v2", [True, False], ids=["v2", "v1"])
@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_lane_logic_matches_reference(case, v2):
    feats = loader_view(case["features"]) if case["features"] is not None else None
    e = Emu(features=[s for _, s in feats] if feats is not None else None, v2=v2, **params_of(case))
    e.count_block(case_fastq(case))
    counts, stats, fast, gen = e.read()
    exp = case["expected"]
    assert stats == exp["stats"]
    if feats is not None:
        assert counts == [r[2] for r in exp["rows"]]
    else:
        rows = e.ec_rows()
        assert [(k, c) for k, c, _ in rows] == [(r[1], r[2]) for r in exp["rows"]]


def test_fast_path_is_exercised():
    # the synthetic fixed-offset cases must go through the packed fast lane, not the general one
    case = next(c for c in CASES if c["name"] == "synth_fixed_m1")
    feats = loader_view(case["features"])
    e = Emu(features=[s for _, s in feats], **params_of(case))
    e.count_block(case_fastq(case))
    _, stats, fast, gen = e.read()
    assert fast == stats[0] and gen == 0          # reads with 'N' in the window stay on the fast path (flag bits)
    assert e.v2_reads() == fast                   # ... and all go through the v2 (4 reads/lane) logic


def test_host_generator_matches_spec():
    guides = synth.make_library(50, 20, 99)
    for kw in (dict(seed=3, n_reads=300), dict(seed=4, n_reads=300, start=17, read_len=61),
               dict(seed=5, n_reads=300, cassette=True, up="GTTTAAGAGCTA", down="CGTTACCAGGTT", max_offset=100),
               dict(seed=6, n_reads=200, cassette=True, up="ACGTACGT", down="TTGGCCAA", max_offset=130, p_n=0.3)):
        e = Emu(features=guides)
        assert e.synth_fastq(0, kw["n_reads"], **kw) == synth.make_fastq(synth.Spec(**kw), guides)
        assert e.synth_fastq(17, 23, **kw) == synth.make_fastq(synth.Spec(**kw), guides, 17, 23)


@pytest.mark.parametrize("v2", [True, False], ids=["v2", "v1"])
@pytest.mark.parametrize("miss", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("glen,n_guides", [(20, 500), (9, 300), (31, 200), (28, 300), (3, 40), (1, 4)])
def test_pigeonhole_vs_oracle_dense(miss, glen, n_guides, v2):
    # dense random libraries + heavy mutation so that ties and multi-piece duplicates are common
    n_guides = min(n_guides, 4 ** glen)
    guides = synth.make_library(n_guides, glen, 1000 + glen)
    spec = synth.Spec(seed=glen * 7 + miss, n_reads=1500, read_len=glen + 9, start=4, p_sub=0.45, p_rand=0.25, p_n=0.05)
    fq = synth.make_fastq(spec, guides)
    kw = dict(miss=miss, length=glen, start="4")
    o = O.Oracle(features=[(f"g{i}", s) for i, s in enumerate(guides)], **kw)
    o.count_fastq(fq)
    e = Emu(features=guides, v2=v2, **kw)
    e.count_block(fq)
    counts, stats, _, _ = e.read()
    assert stats == o.stats() and counts == o.counts()


@pytest.mark.parametrize("miss", [0, 1])
@pytest.mark.parametrize("glen,n_guides,start", [(20, 10000, 0), (20, 13000, 3), (14, 800, 5), (21, 6000, 2), (17, 2000, 9), (16, 3000, 0)])
def test_lds_tables_vs_oracle(miss, glen, n_guides, start):
    """k_count_fixed4_lds's per-read logic: cuckoo tables of half-keyed tags (exact hit + the unique feature at distance 1,
    flagged symbols as forced mismatches) against the oracle, on libraries up to the LDS capacity, with heavy mutation so
    that ties, multi-candidate reads and N symbols are common; and against the pigeonhole tables (lt=False)."""
    guides = synth.make_library(n_guides, glen, 31 * glen + n_guides)
    # near-duplicate features: many guides one substitution away from another one (ambiguous nearest neighbours)
    twins = []
    for i, g in enumerate(guides[:400]):
        p = (i * 7) % glen
        t = g[:p] + "ACGT"[("ACGT".index(g[p]) + 1 + i % 3) % 4] + g[p + 1:]
        twins.append(t)
    lib = list(dict.fromkeys(guides + twins))
    spec = synth.Spec(seed=glen + miss, n_reads=6000, read_len=start + glen + 6, start=start, p_sub=0.35, p_rand=0.1, p_n=0.08, p_lowq=0.1)
    fq = sprinkle_symbols(synth.make_fastq(spec, lib), 3, rate=0.01)
    fq += synth.make_fastq(synth.Spec(seed=5, n_reads=50, read_len=start + glen - 2, start=start), lib)   # clipped windows
    kw = dict(miss=miss, length=glen, start=str(start))
    o = O.Oracle(features=[(str(i), s) for i, s in enumerate(lib)], **kw)
    o.count_fastq(fq)
    e = Emu(features=lib, **kw)
    assert e.lt_ok()
    e.count_block(fq)
    counts, stats, fast, gen = e.read()
    assert stats == o.stats() and counts == o.counts()
    assert e.lt_reads() > 5000 and gen == 0
    e2 = Emu(features=lib, lt=False, **kw)
    e2.count_block(fq)
    assert e2.read()[:2] == (counts, stats) and e2.lt_reads() == 0


@pytest.mark.parametrize("miss", [0, 1])
@pytest.mark.parametrize("glen,n_guides,start,parts", [(20, 9000, 0, 3), (20, 30000, 3, 0), (21, 6000, 2, 5), (17, 2000, 9, 2), (18, 20000, 0, 0),
                                                        (14, 800, 5, 4), (20, 10000, 0, 1), (19, 3000, 7, 1)])
def test_partitioned_tables_vs_oracle(miss, glen, n_guides, start, parts):
    """k_part_scatter / k_part_count's per-read logic (a library dealt into partitions by half 0: table 0 per partition,
    one table 1 over all features; a hit through table 1 counted by its table-1 slot) against the oracle -- on libraries
    beyond one workgroup's LDS (parts = 0: the builder chooses), on small ones forced into partitions and on ONE partition
    (the tables k_part_fused works on), with the
    same near-duplicate features, heavy mutation, N symbols and clipped windows as the LDS-table test."""
    guides = synth.make_library(n_guides, glen, 77 * glen + n_guides)
    twins = []
    for i, g in enumerate(guides[:600]):
        p = (i * 7) % glen
        twins.append(g[:p] + "ACGT"[("ACGT".index(g[p]) + 1 + i % 3) % 4] + g[p + 1:])
    lib = list(dict.fromkeys(guides + twins))
    spec = synth.Spec(seed=glen + miss + 9, n_reads=6000, read_len=start + glen + 6, start=start, p_sub=0.35, p_rand=0.1, p_n=0.08, p_lowq=0.1)
    fq = sprinkle_symbols(synth.make_fastq(spec, lib), 3, rate=0.01)
    fq += synth.make_fastq(synth.Spec(seed=5, n_reads=50, read_len=start + glen - 2, start=start), lib)   # clipped windows
    kw = dict(miss=miss, length=glen, start=str(start))
    o = O.Oracle(features=[(str(i), s) for i, s in enumerate(lib)], **kw)
    o.count_fastq(fq)
    e = Emu(features=lib, pt_parts=parts, **kw)
    assert e.pt_parts() >= max(parts, 2 if n_guides > 15000 else 1)
    e.count_block(fq)
    counts, stats, fast, gen = e.read()
    assert stats == o.stats() and counts == o.counts()
    assert e.pt_reads() > 4000 and gen == 0 and e.lt_reads() == 0


def multi_window_case(starts, length, rl, miss, n_reads=5000):
    """(library, FASTQ bytes, windows) of a seeded multi-window run: features of every part count, parts of a feature
    planted at the windows, substitutions / N, low-quality bases, reads that end inside a window"""
    import random
    rng = random.Random(len(starts) * 100 + length + miss)
    st = [int(x) for x in starts.split(",")]
    W = len(st)
    base = synth.make_library(120, length, 50 + length)
    lib = []
    for _ in range(300):
        k = rng.randint(1, W)
        lib.append(":".join(rng.choice(base) for _ in range(k)))
    lib = list(dict.fromkeys(lib))
    lines = []
    for i in range(n_reads):
        n = rl if rng.random() > 0.06 else rng.randint(0, rl - 1)
        seq = [rng.choice("ACGT") for _ in range(rl)]
        feat = rng.choice(lib).split(":")
        for w, s0 in enumerate(st):                                   # plant parts of a feature (or other parts) at the windows
            part = feat[w % len(feat)] if rng.random() < 0.8 else rng.choice(base)
            seq[s0:s0 + length] = list(part)[: max(0, rl - s0)]
        for _ in range(rng.choice([0, 0, 0, 1, 1, 2])):
            p = rng.randrange(rl); seq[p] = rng.choice("ACGTN")
        q = ["I"] * rl
        for _ in range(rng.choice([0, 0, 1, 2])):
            q[rng.randrange(rl)] = rng.choice("#5>=")
        lines.append(f"@r{i}\n{''.join(seq)[:n]}\n+\n{''.join(q)[:n]}\n")
    return lib, "".join(lines).encode(), W


@pytest.mark.parametrize("miss", [0, 1, 2])
@pytest.mark.parametrize("starts,length,rl", [("0,10", 10, 40), ("3,20,9", 6, 33), ("12,0", 15, 30), ("0,7,14,21", 6, 31), ("5,5", 8, 20), ("0,9,18", 9, 30)])
def test_multi_window_packed_logic_vs_oracle(miss, starts, length, rl):
    """k_count_multi4's per-lane logic (--st a,b,...: ':'-joined parts, failed parts omitted, k-part features) against
    the oracle: libraries holding features of every part count, low-quality parts, N symbols, reads that end inside a
    window (not packed: the byte-exact routine clips them)"""
    lib, fq, W = multi_window_case(starts, length, rl, miss)
    kw = dict(miss=miss, length=length, start=starts)
    o = O.Oracle(features=[(str(i), s) for i, s in enumerate(lib)], **kw)
    o.count_fastq(fq)
    e = Emu(features=lib, **kw)
    e.count_block(fq)
    counts, stats, fast, gen = e.read()
    assert stats == o.stats() and counts == o.counts()
    # the packed slot holds key and feature index in one u64: longer joint keys keep the byte-exact general path
    packed = 2 * W * length + len(lib).bit_length() <= 64
    # the tiles hold the windows only, back to back: a read that ends inside a window is clipped by the byte-exact routine
    short = sum(len(x) < max(int(v) for v in starts.split(",")) + length for x in fq.split(b"\n")[1::4])
    assert (gen == short and e.v2_reads() == 5000 - short and short > 100) if packed else (fast == 0 and gen == 5000)
    assert packed or (starts, length) == ("12,0", 15)
    # a plain feature as long as a joined key makes the run fall back to the byte-exact general path
    lib2 = lib + ["A" * (2 * length + 1)]
    o2 = O.Oracle(features=[(str(i), s) for i, s in enumerate(lib2)], **kw)
    o2.count_fastq(fq)
    e2 = Emu(features=lib2, **kw)
    e2.count_block(fq)
    c2, s2, fast2, gen2 = e2.read()
    assert s2 == o2.stats() and c2 == o2.counts() and fast2 == 0 and gen2 == 5000


def multi_window_uniform_case(starts, length, rl, n_feat=200, n_reads=6000, seed=0):
    """a library whose features all have one part per window (dual-guide pairs), reads with the parts planted at the
    windows: substitutions, N, low-quality parts, some reads that end inside a window"""
    import random
    rng = random.Random(1000 * len(starts) + length + seed)
    st = [int(x) for x in starts.split(",")]
    base = synth.make_library(len(st) * n_feat, length, 60 + length)            # every part belongs to one feature (a pair library
    lib = [":".join(base[len(st) * f + w] for w in range(len(st))) for f in range(n_feat)]   # of unrelated guides)
    lines = []
    for i in range(n_reads):
        n = rl if rng.random() > 0.04 else rng.randint(0, rl - 1)
        seq = [rng.choice("ACGT") for _ in range(rl)]
        feat = rng.choice(lib).split(":")
        for w, s0 in enumerate(st):
            part = feat[w] if rng.random() < 0.9 else rng.choice(base)
            seq[s0:s0 + length] = list(part)[: max(0, rl - s0)]
        for _ in range(rng.choice([0, 0, 1, 1, 2])):
            p = rng.randrange(rl); seq[p] = rng.choice("ACGTN")
        q = ["I"] * rl
        for _ in range(rng.choice([0, 0, 0, 1, 2])):
            q[rng.randrange(rl)] = rng.choice("#5>=")
        lines.append(f"@r{i}\n{''.join(seq)[:n]}\n+\n{''.join(q)[:n]}\n")
    return lib, "".join(lines).encode()


@pytest.mark.parametrize("miss", [0, 1])
@pytest.mark.parametrize("starts,length,rl", [("0,10", 10, 40), ("3,40,21", 6, 60), ("5,100", 8, 150), ("0,7,14,21", 5, 31), ("60,2", 9, 75), ("0,20", 7, 30)])
def test_multi_window_lds_tables_vs_oracle(miss, starts, length, rl):
    """several windows, every feature with one part per window (dual-guide libraries): the joined keys go through the
    library-in-LDS logic (k_count_fixed4_lds<.., MW>: the tiles hold the windows back to back, the Phred rule is applied
    part by part) -- against the oracle and against the k-part packed tables (k_count_multi4's logic)"""
    lib, fq = multi_window_uniform_case(starts, length, rl)
    kw = dict(miss=miss, length=length, start=starts)
    o = O.Oracle(features=[(str(i), s) for i, s in enumerate(lib)], **kw)
    o.count_fastq(fq)
    e = Emu(features=lib, **kw)
    e.count_block(fq)
    counts, stats, fast, gen = e.read()
    assert stats == o.stats() and counts == o.counts()
    short = sum(len(x) < max(int(v) for v in starts.split(",")) + length for x in fq.split(b"\n")[1::4])
    assert gen == short and e.lt_reads() == 6000 - short and stats[2] > 0 or miss == 0
    assert e.lt_ok()
    e2 = Emu(features=lib, lt=False, **kw)
    e2.count_block(fq)
    assert e2.read()[:2] == (counts, stats) and e2.lt_reads() == 0
    # a combinatorial library (one guide paired with many partners: with the plain joined key more than four features
    # would share a half and the tables could not be built; two-window keys are looked up in their mixed form, mw_mix)
    parts0 = [f.split(":") for f in lib[:40]]
    lib4 = lib + [":".join([parts0[i % 5][0]] + p[1:]) for i, p in enumerate(parts0[5:40])] + [":".join(p[:-1] + [parts0[i % 3][-1]]) for i, p in enumerate(parts0[3:30])]
    lib4 = list(dict.fromkeys(lib4))
    e4 = Emu(features=lib4, **kw)
    e4.count_block(fq)
    o4 = O.Oracle(features=[(str(i), s) for i, s in enumerate(lib4)], **kw)
    o4.count_fastq(fq)
    assert e4.read()[:2] == (o4.counts(), o4.stats())
    if starts.count(",") == 1:
        assert e4.lt_ok() and e4.lt_reads() == 6000 - short
    # a library that also holds a feature of fewer parts keeps to the k-part tables
    lib3 = lib + [lib[0].split(":")[0]]
    e3 = Emu(features=lib3, **kw)
    e3.count_block(fq)
    o3 = O.Oracle(features=[(str(i), s) for i, s in enumerate(lib3)], **kw)
    o3.count_fastq(fq)
    assert e3.read()[:2] == (o3.counts(), o3.stats()) and e3.lt_reads() == 0


def test_lds_tables_applicability():
    """built only for uniform ACGT libraries of 14..21-base features searched with --m <= 1 that fit the tables"""
    g20 = synth.make_library(500, 20, 1)
    assert Emu(features=g20, miss=1, length=20).lt_ok() and Emu(features=g20, miss=0, length=20).lt_ok()
    assert not Emu(features=g20, miss=2, length=20).lt_ok()
    assert not Emu(features=g20 + ["ACGTACGTACGTAC"], miss=1, length=20).lt_ok()            # mixed lengths
    assert not Emu(features=g20[:-1] + [g20[-1][:5] + "N" + g20[-1][6:]], miss=1, length=20).lt_ok()   # irregular feature
    assert not Emu(features=synth.make_library(300, 13, 2), miss=1, length=13).lt_ok()
    assert not Emu(features=synth.make_library(300, 22, 2), miss=1, length=22).lt_ok()
    assert not Emu(features=synth.make_library(15000, 20, 3), miss=1, length=20).lt_ok()     # beyond the LDS capacity
    # shared halves: 3 features with one left half fit (2 buckets x 2 slots), 5 cannot -> no tables, pigeonhole kernel
    left = "ACGTTGCAAC"
    rights = ["AAAAAAAAAA", "CCCCCCCCCC", "GGGGGGGGGG", "TTTTTTTTTT", "ACACACACAC"]
    assert Emu(features=[left + r for r in rights[:3]], miss=1, length=20).lt_ok()
    assert not Emu(features=[left + r for r in rights], miss=1, length=20).lt_ok()


@pytest.mark.parametrize("start,length", [(0, 20), (3, 20), (13, 17), (15, 31), (16, 16), (1, 1), (30, 5), (2, 0)])
def test_window_geometry_sweep(start, length):
    # every alignment of the window against the 16-base / 4-quality word grid, plus clipped reads
    glen = max(length, 1)
    guides = synth.make_library(min(200, 4 ** glen), glen, 500 + start)
    spec = synth.Spec(seed=start * 31 + length, n_reads=1200, read_len=start + glen + 3, start=start, p_lowq=0.2)
    fq = synth.make_fastq(spec, guides) + synth.make_fastq(synth.Spec(seed=77, n_reads=100, read_len=start + glen - 1,
                                                                        start=max(0, start - 1)), guides)
    kw = dict(miss=1, length=length, start=str(start))
    o = O.Oracle(features=[(f"g{i}", s) for i, s in enumerate(guides)], **kw)
    o.count_fastq(fq)
    for v2 in (True, False):
        e = Emu(features=guides, v2=v2, **kw)
        e.count_block(fq)
        counts, stats, _, _ = e.read()
        assert stats == o.stats() and counts == o.counts(), v2


@pytest.mark.parametrize("miss", [0, 1, 2, 3])
def test_odd_symbols_in_the_window(miss):
    """lower case, N, IUPAC and junk symbols inside and outside the window: in-band flag bits on the fast
    path (all-ACGT library) and the general path (library that itself holds an N) both match the oracle"""
    guides = synth.make_library(150, 12, 4242)
    fq = sprinkle_symbols(synth.make_fastq(synth.Spec(seed=miss, n_reads=2500, read_len=40, start=5, p_sub=0.3), guides), 9)
    for lib in (guides, guides[:100] + [g[:5] + "N" + g[6:] for g in guides[100:]]):
        kw = dict(miss=miss, length=12, start="5")
        o = O.Oracle(features=[(str(i), s) for i, s in enumerate(lib)], **kw)
        o.count_fastq(fq)
        for v2 in (True, False):
            e = Emu(features=lib, v2=v2, **kw)
            e.count_block(fq)
            counts, stats, fast, gen = e.read()
            assert stats == o.stats() and counts == o.counts()
            assert (gen == 0) == (lib is guides)


def general_key_case(miss, glen, n=160, n_reads=2000):
    base = synth.make_library(n, glen, 77 + glen)
    a, b, c = n * 5 // 8, n * 13 // 16, n * 29 // 32
    lib = list(base[:a])
    lib += [g[:3] + "N" + g[4:] for g in base[a:b]]                      # odd symbol: an irregular feature
    lib += [base[0][:glen - 1] + ch for ch in "CGT" if base[0][:glen - 1] + ch not in lib]   # near-duplicates -> ties
    lib += sorted({g[:glen - 2] for g in base[b:c]})                    # shorter features
    lib += [g + "AC" for g in base[c:]]                                  # longer features
    spec = synth.Spec(seed=miss + glen, n_reads=n_reads, read_len=glen + 12, start=6, p_sub=0.35, p_rand=0.1, p_n=0.05)
    return lib, sprinkle_symbols(synth.make_fastq(spec, base), 5)


@pytest.mark.parametrize("miss", [0, 1, 2, 3, 7])
@pytest.mark.parametrize("glen", [12, 40])
def test_general_key_index_vs_oracle(miss, glen):
    """the byte-string index (GkDesc: exact table + miss+1 piece tables per feature length) that serves every key the
    2-bit tables cannot: libraries holding odd symbols or several lengths, and windows longer than 31 bases.  Shared
    prefixes make long probe chains and ties; features of other lengths must never be candidates (fast2q.py:683)."""
    lib, fq = general_key_case(miss, glen)
    kw = dict(miss=miss, length=glen, start="6")
    o = O.Oracle(features=[(f"g{i}", s) for i, s in enumerate(lib)], **kw)
    o.count_fastq(fq)
    e = Emu(features=lib, **kw)
    e.count_block(fq)
    counts, stats, fast, gen = e.read()
    assert stats == o.stats() and counts == o.counts()
    assert gen > 0 and (glen <= 31 or gen == stats[0]) and stats[1] > 0 and (miss == 0 or stats[2] > 0)


UP, DOWN = "GTTTAAGAGCTA", "CGTTACCAGGTT"


UP2, DOWN2 = "ACCTGGATCCAA", "TTCAGGCATGCA"


def multi_pair_case(n_reads, seed, n_guides=120, three=False):
    """reads that carry two (three) cassettes UPi + guide + DOWNi; substitutions, low qualities, an occasional N, some
    cassettes destroyed so that parts drop out; the library joins the planted guides with ':' and also holds plain ones"""
    import random
    rng = random.Random(seed)
    guides = synth.make_library(n_guides, 18, 300 + seed)
    ups, downs = [UP, UP2] + (["GGTACCTTAGCA"] if three else []), [DOWN, DOWN2] + (["CATGTTGACCTA"] if three else [])
    lib = [":".join(guides[(i * (k + 3) + k) % n_guides] for k in range(len(ups))) for i in range(n_guides)]
    lib += guides[:20]                                                 # one-part features: reads whose other part failed
    recs = []
    for i in range(n_reads):
        g = rng.randrange(n_guides)
        parts = [guides[(g * (k + 3) + k) % n_guides] for k in range(len(ups))]
        seq = "".join(rng.choice("ACGT") for _ in range(rng.randrange(0, 9)))
        for k in range(len(ups)):
            seq += ups[k] + parts[k] + downs[k] + "".join(rng.choice("ACGT") for _ in range(rng.randrange(0, 6)))
        b = bytearray(seq.encode())
        q = bytearray(b"I" * len(b))
        for _ in range(rng.choice([0, 0, 0, 1, 1, 2, 4])):
            b[rng.randrange(len(b))] = rng.choice(b"ACGT")
        if rng.random() < 0.05:
            b[rng.randrange(len(b))] = ord("N")
        if rng.random() < 0.2:
            q[rng.randrange(len(q))] = rng.choice(b"#+5:")
        recs.append(b"@r%d\n%s\n+\n%s\n" % (i, bytes(b), bytes(q)))
    return lib, b"".join(recs), ups, downs


@pytest.mark.parametrize("mode", ["C", "EC"])
@pytest.mark.parametrize("ms,three", [(0, False), (1, False), (1, True), (2, False)])
def test_multi_pair_packed_logic_vs_oracle(mode, ms, three):
    """several --us/--ds pairs on the planes (pairs_lane): every pair searched on its own, parts joined with ':', failed
    parts left out -- against the oracle, Counter mode (string match against ':' features, m = 1) and Extract+Count"""
    lib, fq, ups, downs = multi_pair_case(2500, 7 + ms + (10 if three else 0), three=three)
    kw = dict(mode=mode, miss=1, upstream=",".join(ups), downstream=",".join(downs), miss_search_up=ms, miss_search_down=ms)
    o = O.Oracle(features=[(str(i), s) for i, s in enumerate(lib)] if mode == "C" else None, **kw)
    o.count_fastq(fq)
    e = Emu(features=lib if mode == "C" else None, **kw)
    e.count_block(fq)
    counts, stats, fast, gen = e.read()
    assert stats == o.stats()
    assert e.anchor_reads() == fast and fast > 0.8 * stats[0]            # the packed path did the work
    if mode == "C":
        assert counts == o.counts() and stats[1] > 0 and stats[2] > 0
    else:
        assert [(k, n) for k, n, _ in e.ec_rows()] == list(zip(o.keys(), o.counts()))
        assert any(":" in k for k in o.keys())


def pair_library_case(n_reads, seed, la=18, lb=18, combinatorial=False, n_feat=150):
    """two cassettes UP + A + DOWN, UP2 + B + DOWN2 against a library of A:B features only (dual-guide libraries): pairs of
    unrelated guides, or combinatorial (a guide with several partners); substitutions, N, low qualities, destroyed
    cassettes, windows of other lengths, a lone cassette whose window is as long as a joined key"""
    import random
    rng = random.Random(seed)
    ga, gb = synth.make_library(n_feat, la, 400 + seed), synth.make_library(n_feat, lb, 500 + seed)
    if combinatorial:
        lib = list(dict.fromkeys(f"{ga[i % 12]}:{gb[(i * 7) % 40]}" for i in range(3 * n_feat)))
    else:
        lib = [f"{ga[i]}:{gb[i]}" for i in range(n_feat)]
    recs = []
    for i in range(n_reads):
        a, b = rng.choice(lib).split(":")
        if rng.random() < 0.1:
            b = rng.choice(gb)                                        # a pair the library may not hold
        if rng.random() < 0.04:
            a = a[:-1] if rng.random() < 0.5 else a + rng.choice("ACGT")      # a window of another length
        seq = "".join(rng.choice("ACGT") for _ in range(rng.randrange(0, 9)))
        if rng.random() < 0.03 and la + 1 + lb <= 31:
            seq += UP + a + rng.choice("ACGTN") + b + DOWN             # one cassette around A?B: one substitution from A:B
        else:
            seq += UP + a + DOWN + "".join(rng.choice("ACGT") for _ in range(rng.randrange(0, 6))) + UP2 + b + DOWN2
        seq += "".join(rng.choice("ACGT") for _ in range(rng.randrange(0, 6)))
        bts = bytearray(seq.encode())
        q = bytearray(b"I" * len(bts))
        for _ in range(rng.choice([0, 0, 0, 1, 1, 2, 3])):
            bts[rng.randrange(len(bts))] = rng.choice(b"ACGT")
        if rng.random() < 0.06:
            bts[rng.randrange(len(bts))] = ord("N")
        if rng.random() < 0.2:
            q[rng.randrange(len(q))] = rng.choice(b"#+5:")
        recs.append(b"@r%d\n%s\n+\n%s\n" % (i, bytes(bts), bytes(q)))
    return lib, b"".join(recs)


@pytest.mark.parametrize("miss", [0, 1])
@pytest.mark.parametrize("la,lb,combinatorial,ms", [(18, 18, False, 1), (20, 20, True, 0), (10, 12, False, 1), (7, 20, True, 1), (14, 9, False, 2)])
def test_pair_tables_vs_oracle(miss, la, lb, combinatorial, ms):
    """two --us/--ds pairs against a pure A:B library: the joined key as two 2-bit words through the pair tables (PwDesc:
    exact by (A, B), --m 1 through the features sharing A or sharing B) -- against the oracle and the string index"""
    lib, fq = pair_library_case(4000, 3 * la + lb + ms, la, lb, combinatorial)
    kw = dict(miss=miss, upstream=f"{UP},{UP2}", downstream=f"{DOWN},{DOWN2}", miss_search_up=ms, miss_search_down=ms)
    o = O.Oracle(features=[(str(i), s) for i, s in enumerate(lib)], **kw)
    o.count_fastq(fq)
    e = Emu(features=lib, **kw)
    assert e.pw_ok()
    e.count_block(fq)
    counts, stats, fast, gen = e.read()
    assert stats == o.stats() and counts == o.counts()
    assert fast > 0.9 * stats[0] and stats[1] > 1000 and (miss == 0 or stats[2] > 50)
    e2 = Emu(features=lib, pw=False, **kw)
    assert not e2.pw_ok()
    e2.count_block(fq)
    assert e2.read()[:2] == (counts, stats)
    # a library with a feature of another shape keeps to the string index
    assert not Emu(features=lib + [lib[0].split(":")[0]], **kw).pw_ok() and not Emu(features=lib + [lib[0] + ":" + lib[1]], **kw).pw_ok()


@pytest.mark.parametrize("mode", ["C", "EC"])
@pytest.mark.parametrize("anchors", ["both", "up", "down"])
@pytest.mark.parametrize("ms,qs", [(0, 30), (1, 30), (2, 30), (1, 12), (3, 41)])
def test_packed_anchor_logic_vs_oracle(mode, anchors, ms, qs):
    """the bit-plane anchored lane logic (planar tiles, bit-sliced mismatch counters, fail vectors)"""
    guides = synth.make_library(200, 20, 900 + ms)
    spec = synth.Spec(seed=ms * 7 + qs, n_reads=2500, read_len=150, cassette=True, up=UP, down=DOWN, max_offset=110,
                      p_sub=0.2, p_lowq=0.15, p_n=0.02)
    fq = synth.make_fastq(spec, guides)
    # damage some anchors and some anchor qualities so that msu/msd and qsu/qsd matter
    import random
    rng = random.Random(5)
    lines = fq.split(b"\n")
    for i in range(1, len(lines), 4):
        if rng.random() < 0.5:
            b = bytearray(lines[i]); q = bytearray(lines[i + 2])
            for _ in range(rng.randint(1, 3)):
                p = rng.randrange(len(b)); b[p] = rng.choice(b"ACGT")
            q[rng.randrange(len(q))] = rng.choice(b"#+5:?")
            lines[i], lines[i + 2] = bytes(b), bytes(q)
    fq = b"\n".join(lines)
    kw = dict(mode=mode, miss=1, length=20, miss_search_up=ms, miss_search_down=ms, qual_up=qs, qual_down=30)
    if anchors in ("both", "up"):
        kw["upstream"] = UP
    if anchors in ("both", "down"):
        kw["downstream"] = DOWN
    o = O.Oracle(features=[(str(i), s) for i, s in enumerate(guides)] if mode == "C" else None, **kw)
    o.count_fastq(fq)
    e = Emu(features=guides if mode == "C" else None, **kw)
    e.count_block(fq)
    counts, stats, fast, gen = e.read()
    assert stats == o.stats()
    assert e.anchor_reads() == fast and fast > 0.9 * stats[0]           # the packed path did the work
    if mode == "C":
        assert counts == o.counts()
    else:
        assert [(k, n) for k, n, _ in e.ec_rows()] == list(zip(o.keys(), o.counts()))


@pytest.mark.parametrize("mode", ["C", "EC"])
@pytest.mark.parametrize("anchors,ms,rl", [("both", 1, 251), ("up", 0, 301), ("down", 2, 200), ("pairs", 1, 320)])
def test_packed_anchor_long_reads(mode, anchors, ms, rl):
    """reads of 161 .. 320 bases (MiSeq 2 x 250 / 2 x 300 amplicons) stay on the bit-plane path: ten 32-base plane words
    per read (NW = 10), the cassette anywhere in the read, mixed with short reads and a few reads beyond 320 bases,
    which take the byte-exact routine"""
    guides = synth.make_library(150, 20, 77 + rl)
    spec = synth.Spec(seed=rl + ms, n_reads=1500, read_len=rl, cassette=True, up=UP, down=DOWN, max_offset=rl - 50, p_sub=0.2, p_lowq=0.1, p_n=0.02)
    fq = synth.make_fastq(spec, guides)
    fq += synth.make_fastq(synth.Spec(seed=3, n_reads=200, read_len=90, cassette=True, up=UP, down=DOWN, max_offset=40), guides)
    fq += synth.make_fastq(synth.Spec(seed=4, n_reads=40, read_len=400, cassette=True, up=UP, down=DOWN, max_offset=330), guides)
    kw = dict(mode=mode, miss=1, length=20, miss_search_up=ms, miss_search_down=ms)
    feats = guides
    if anchors == "pairs":
        kw["upstream"] = UP + "," + UP; kw["downstream"] = DOWN + "," + DOWN
        feats = [g + ":" + g for g in guides[:100]] + guides[100:]
    else:
        if anchors in ("both", "up"):
            kw["upstream"] = UP
        if anchors in ("both", "down"):
            kw["downstream"] = DOWN
    o = O.Oracle(features=[(str(i), s) for i, s in enumerate(feats)] if mode == "C" else None, **kw)
    o.count_fastq(fq)
    e = Emu(features=feats if mode == "C" else None, **kw)
    e.count_block(fq)
    counts, stats, fast, gen = e.read()
    assert stats == o.stats()
    # only the reads beyond 320 bases leave the packed path (and, with ':' features in the library, the reads with an N)
    assert e.anchor_reads() == fast and fast + gen == 1740 and (gen == 40 if anchors != "pairs" else 40 <= gen < 120)
    if mode == "C":
        assert counts == o.counts()
    else:
        assert [(k, n) for k, n, _ in e.ec_rows()] == list(zip(o.keys(), o.counts()))


def test_packed_anchor_short_and_ragged_reads():
    guides = synth.make_library(50, 12, 77)
    parts = []
    for n, rl in ((300, 40), (300, 96), (300, 97), (200, 160), (100, 161), (50, 13), (60, 320), (30, 321)):
        parts.append(synth.make_fastq(synth.Spec(seed=rl, n_reads=n, read_len=rl, cassette=True, up="ACGTAC", down="TTGCA",
                                                 max_offset=max(0, rl - 30)), guides))
    fq = b"".join(parts)
    for kw in (dict(upstream="ACGTAC", downstream="TTGCA", miss_search_up=1), dict(upstream="ACGTAC", length=12),
               dict(downstream="TTGCA", length=12, miss_search_down=2)):
        o = O.Oracle(features=[(str(i), s) for i, s in enumerate(guides)], miss=1, **kw)
        o.count_fastq(fq)
        e = Emu(features=guides, miss=1, **kw)
        e.count_block(fq)
        counts, stats, fast, gen = e.read()
        assert stats == o.stats() and counts == o.counts()
        assert gen == 30               # only the 321-base reads exceed the packed kernels (reads holding an N are flagged in place)


@pytest.mark.parametrize("anchors", ["both", "up", "down"])
def test_packed_anchor_with_odd_symbols(anchors):
    """N / IUPAC / lower case anywhere in the read (anchors included): flag bits in Counter mode, raw bytes otherwise"""
    guides = synth.make_library(120, 16, 31337)
    spec = synth.Spec(seed=4, n_reads=3000, read_len=120, cassette=True, up=UP, down=DOWN, max_offset=70, p_sub=0.25)
    fq = sprinkle_symbols(synth.make_fastq(spec, guides), 11, rate=0.01)
    kw = dict(miss=2, length=16, miss_search_up=1, miss_search_down=1)
    if anchors in ("both", "up"):
        kw["upstream"] = UP
    if anchors in ("both", "down"):
        kw["downstream"] = DOWN
    for mode in ("C", "EC"):
        o = O.Oracle(features=[(str(i), s) for i, s in enumerate(guides)] if mode == "C" else None, mode=mode, **kw)
        o.count_fastq(fq)
        e = Emu(features=guides if mode == "C" else None, mode=mode, **kw)
        e.count_block(fq)
        counts, stats, fast, gen = e.read()
        assert stats == o.stats()
        if mode == "C":
            assert counts == o.counts() and fast > 0.5 * stats[0]
        else:
            assert [(k, n) for k, n, _ in e.ec_rows()] == list(zip(o.keys(), o.counts()))


def lower_case_some(fastq, seed, share=0.5, rate=0.08, whole=0.1):
    """rewrite bases of `share` of the reads in lower case (each base with probability `rate`; `whole` of those reads entirely)"""
    import random
    rng = random.Random(seed)
    lines = fastq.split(b"\n")
    for i in range(1, len(lines), 4):
        if rng.random() < share:
            b = bytearray(lines[i])
            allb = rng.random() < whole
            for j in range(len(b)):
                if b[j] in b"ACGT" and (allb or rng.random() < rate):
                    b[j] |= 0x20
            lines[i] = bytes(b)
    return b"\n".join(lines)


@pytest.mark.parametrize("mode", ["C", "EC"])
@pytest.mark.parametrize("anchors,ms", [("both", 1), ("up", 0), ("down", 2), ("pairs", 1)])
def test_packed_anchor_mixed_case_reads(mode, anchors, ms):
    """lower-case bases in anchored runs stay on the bit-plane path: they match no anchor symbol (the search is
    case-sensitive, fast2q.py:337) yet are ordinary bases of the upper-cased window (:354) -- marked in the quality
    plane for the anchors, their codes stored for the key (F2Q_LEN_CASE).  Half of the reads carry lower-case bases
    (in anchors, windows, flanks; some reads entirely); a read with lower case AND an N takes the byte-exact routine."""
    guides = synth.make_library(150, 20, 4711)
    spec = synth.Spec(seed=ms + 40, n_reads=4000, read_len=150, cassette=True, up=UP, down=DOWN, max_offset=100, p_sub=0.2, p_lowq=0.1, p_n=0.01)
    fq = lower_case_some(synth.make_fastq(spec, guides), 17)
    kw = dict(mode=mode, miss=1, length=20, miss_search_up=ms, miss_search_down=ms)
    feats = guides
    if anchors == "pairs":
        kw["upstream"] = UP + "," + UP; kw["downstream"] = DOWN + "," + DOWN
        feats = [g + ":" + g for g in guides[:100]] + guides[100:]
    else:
        if anchors in ("both", "up"):
            kw["upstream"] = UP
        if anchors in ("both", "down"):
            kw["downstream"] = DOWN
    o = O.Oracle(features=[(str(i), s) for i, s in enumerate(feats)] if mode == "C" else None, **kw)
    o.count_fastq(fq)
    e = Emu(features=feats if mode == "C" else None, **kw)
    e.count_block(fq)
    counts, stats, fast, gen = e.read()
    assert stats == o.stats()
    assert e.anchor_reads() == fast and fast + gen == 4000 and gen < (60 if anchors != "pairs" else 120)     # lower case + N in one read, N with ':' features
    if mode == "C":
        assert counts == o.counts()
    else:
        assert [(k, n) for k, n, _ in e.ec_rows()] == list(zip(o.keys(), o.counts()))


@pytest.mark.parametrize("start,length,rl", [(0, 20, 150), (7, 29, 60), (3, 12, 14), (10, 8, 9), (0, 0, 30), (5, 30, 80)])
def test_extract_count_fixed_window(start, length, rl):
    """Extract+Count with --st/--l on the packed path: clipped and empty windows, odd symbols, Phred filter"""
    guides = synth.make_library(60, max(8, min(length, 20)), 777)
    fq = sprinkle_symbols(synth.make_fastq(synth.Spec(seed=start + length, n_reads=2500, read_len=rl, start=min(start, rl - 1),
                                                       p_lowq=0.2), guides), 2, rate=0.01)
    fq += synth.make_fastq(synth.Spec(seed=1, n_reads=200, read_len=max(1, start)), guides)     # reads ending before the window
    kw = dict(mode="EC", start=str(start), length=length)
    o = O.Oracle(**kw)
    o.count_fastq(fq)
    e = Emu(**kw)
    e.count_block(fq)
    _, stats, fast, gen = e.read()
    assert stats == o.stats()
    assert [(k, n) for k, n, _ in e.ec_rows()] == list(zip(o.keys(), o.counts()))
    assert (fast > 0) == (length <= 29)          # windows longer than 29 bases use the byte-string table (general path)


@pytest.mark.parametrize("start,length,rl,rate", [(0, 20, 150, 0.03), (5, 26, 40, 0.04), (2, 29, 31, 0.02), (4, 12, 14, 0.1)])
def test_extract_count_fixed_window_n_reads_stay_packed(start, length, rl, rate):
    """Extract+Count with --st/--l: a window with up to three 'N' / 'n' has a single-word key and stays on the tiles
    (round 3; such reads took the byte-exact kernel before); more 'N's, or a window too long to spell them, do not"""
    def fits(nn, n):                                  # ec64_word's rule (f2q_device.h)
        return n <= 29 and nn <= 3 and 2 * n + 2 + 5 * nn <= 58
    guides = synth.make_library(60, max(8, min(length, 20)), 778)
    fq = sprinkle_symbols(synth.make_fastq(synth.Spec(seed=start + length, n_reads=3000, read_len=rl, start=start, p_lowq=0.1), guides),
                          5, rate=rate, symbols=b"Nn")
    kw = dict(mode="EC", start=str(start), length=length)
    o = O.Oracle(**kw)
    o.count_fastq(fq)
    e = Emu(**kw)
    e.count_block(fq)
    _, stats, fast, gen = e.read()
    assert stats == o.stats()
    assert [(k, n) for k, n, _ in e.ec_rows()] == list(zip(o.keys(), o.counts()))
    want_gen = 0
    for seq in fq.split(b"\n")[1::4]:
        w = seq[start:start + length].upper()
        nn = w.count(b"N")
        want_gen += bool(nn) and not fits(nn, len(w))
    assert gen == want_gen and fast + gen == 3000
    assert any("N" in k for k in o.keys())


def test_fuzz_lane_logic_vs_oracle():
    """400 seeded random cases (tests/fuzz_cases.py): every mode, awkward symbols / lengths / framing"""
    from fuzz_cases import make_case
    for seed in range(400):
        kw, feats, fq = make_case(seed)
        o = O.Oracle(features=[(str(i), s) for i, s in enumerate(feats)] if feats is not None else None, **kw)
        used_o = o.count_fastq(fq)
        e = Emu(features=feats, **kw)
        used = e.count_block(fq)
        counts, stats, _, _ = e.read()
        assert used == used_o, (seed, kw)
        assert stats == o.stats(), (seed, kw)
        if feats is not None:
            assert counts == o.counts(), (seed, kw)
        else:
            assert [(k, n) for k, n, _ in e.ec_rows()] == list(zip(o.keys(), o.counts())), (seed, kw)
