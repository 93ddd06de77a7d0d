"""GPU parity tests: the HIP path (through the C-ABI of libf2q_hip.so) against the golden vectors
captured from the reference, and against the oracle on seeded synthetic inputs.  Bit-exact."""
import os
import pytest

import synth
from conftest import case_fastq, load_cases, loader_view, pkg, sprinkle_symbols
from oracle import oracle as O
from test_lane_logic_cpu import params_of

pytestmark = pytest.mark.gpu
CASES = load_cases()


@pytest.fixture(scope="module")
def P():
    return pkg()


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_golden_case(P, case):
    feats = loader_view(case["features"]) if case["features"] is not None else None
    data = case_fastq(case)
    exp = case["expected"]
    with P.Counter(features=[s for _, s in feats] if feats is not None else None, **params_of(case)) as c:
        used = c.count_block(data)
        counts, stats = c.read_counts()
        assert list(stats) == exp["stats"]
        if feats is not None:
            assert list(counts) == [r[2] for r in exp["rows"]]
        else:
            assert [(k, n) for k, n, _ in c.ec_results()] == [(r[1], r[2]) for r in exp["rows"]]
        # streaming the same buffer in record-aligned pieces accumulates to the same result
        if len(data) > 200 and case["name"].startswith("synth_fixed_m"):
            c.reset()
            for piece in O.split_fastq_on_records(data, 5):
                assert c.count_block(piece) == len(piece)
            counts2, stats2 = c.read_counts()
            assert list(stats2) == exp["stats"] and list(counts2) == [r[2] for r in exp["rows"]]


@pytest.mark.parametrize("miss", [0, 1, 2, 3])
@pytest.mark.parametrize("glen,n_guides,n_reads", [(20, 1000, 200000), (20, 30000, 100000), (12, 3000, 50000), (31, 500, 30000)])
def test_device_synth_vs_oracle(P, miss, glen, n_guides, n_reads):
    """device-generated resident block == host FASTQ of the same spec == oracle"""
    guides = P.binding.synth_library(0xF2A5 + glen, n_guides, glen)
    assert guides[:20] == synth.make_library(20, glen, 0xF2A5 + glen)
    spec = dict(seed=100 + miss, n_reads=n_reads, read_len=150, start=3)
    kw = dict(miss=miss, length=glen, start="3")
    with P.Counter(features=guides, **kw) as c:
        fq = bytes(c.synth_fastq(**spec))
        assert fq[:4000] == synth.make_fastq(synth.Spec(**spec), guides, 0, 40)[:4000]
        orc = O.count_fastq_parallel(fq, 8, features=[(str(i), s) for i, s in enumerate(guides)], **kw)
        c.count_block(fq)
        counts, stats = c.read_counts()
        assert list(stats) == orc.stats()
        assert list(counts) == orc.counts()
        c.reset()
        blk = c.synth_create(**spec)
        info = blk.info()
        assert info["n_reads"] == n_reads
        t = c.count_resident(blk)
        counts2, stats2 = c.read_counts()
        assert list(stats2) == orc.stats() and list(counts2) == orc.counts()
        assert t["reads"] == n_reads and t["general_reads"] == info["n_general"]
        # idempotence of accumulation: a second pass doubles everything
        c.count_resident(blk)
        counts3, stats3 = c.read_counts()
        assert list(stats3) == [2 * v for v in orc.stats()] and list(counts3) == [2 * v for v in orc.counts()]


def test_anchored_device_synth_vs_oracle(P):
    up, down = "GTTTAAGAGCTA", "CGTTACCAGGTT"
    guides = P.binding.synth_library(0xF2A5 + 5, 2000, 20)
    spec = dict(seed=55, n_reads=60000, read_len=150, cassette=True, up=up, down=down, max_offset=100)
    for mode in ("C", "EC"):
        kw = dict(mode=mode, miss=1, upstream=up, downstream=down, miss_search_up=1, miss_search_down=1)
        with P.Counter(features=guides, miss=1) as gen:
            fq = bytes(gen.synth_fastq(**spec))
        feats = [(str(i), s) for i, s in enumerate(guides)] if mode == "C" else None
        orc = O.Oracle(features=feats, **kw)
        orc.count_fastq(fq)
        with P.Counter(features=guides if mode == "C" else None, **kw) as c:
            c.count_block(fq)
            counts, stats = c.read_counts()
            assert list(stats) == orc.stats()
            if mode == "C":
                assert list(counts) == orc.counts()
            else:
                assert [(k, n) for k, n, _ in c.ec_results()] == list(zip(orc.keys(), orc.counts()))


def test_full_size_invariants(P):
    """BASELINE config 2 at full size (10M reads, 1k guides, m=0): size-independent properties."""
    guides = P.binding.synth_library(0xF2A5 + 2, 1000, 20)
    n = 10_000_000
    with P.Counter(features=guides, miss=0) as c:
        blk = c.synth_create(seed=2, n_reads=n)
        c.count_resident(blk)
        counts, stats = c.read_counts()
        assert stats[0] == n and stats[0] == stats[1] + stats[2] + stats[3] + stats[4]
        assert counts.sum() == stats[1] + stats[2] and stats[2] == 0
        # shard-sum invariance: 4 shards of the same stream == the whole
        blk.free()
        c.reset()
        for s in range(4):
            b = c.synth_create(seed=2, n_reads=n // 4, first_read=s * (n // 4))
            c.count_resident(b)
            b.free()
        counts2, stats2 = c.read_counts()
        assert list(stats2) == list(stats) and (counts2 == counts).all()


def test_error_paths(P):
    with P.Counter(miss=1) as c:                       # Counter mode without a library
        with pytest.raises(P.F2QError):
            c.count_block(b"@r\nACGT\n+\nIIII\n")
    with pytest.raises(P.F2QError):
        P.Counter(upstream="ACGT,ACGT", downstream="ACGT")      # unpaired anchors (fast2q.py:553-556)
    with P.Counter(features=["ACGT"], length=4) as c:
        assert c.count_block(b"") == 0
        assert c.count_block(b"@r\nACGT\n+\n") == 0            # partial record: nothing consumed
        assert list(c.read_counts()[1]) == [0, 0, 0, 0, 0]


@pytest.mark.parametrize("force_v1", ["0", "1"], ids=["v2", "v1"])
@pytest.mark.parametrize("start,length", [(0, 20), (3, 20), (13, 17), (15, 31), (16, 16), (1, 1), (30, 5), (2, 0)])
def test_window_geometry_sweep_gpu(P, monkeypatch, start, length, force_v1):
    """every alignment of the window against the tile word grid, clipped reads, both fast kernels"""
    monkeypatch.setenv("F2Q_FORCE_V1", force_v1)
    glen = max(length, 1)
    guides = synth.make_library(min(200, 4 ** glen), glen, 500 + start)
    spec = synth.Spec(seed=start * 31 + length, n_reads=3000, read_len=start + glen + 3, start=start, p_lowq=0.2)
    fq = synth.make_fastq(spec, guides) + synth.make_fastq(synth.Spec(seed=77, n_reads=100, read_len=start + glen - 1,
                                                                        start=max(0, start - 1)), guides)
    kw = dict(miss=1, length=length, start=str(start))
    orc = O.Oracle(features=[(f"g{i}", s) for i, s in enumerate(guides)], **kw)
    orc.count_fastq(fq)
    with P.Counter(features=guides, **kw) as c:
        c.count_block(fq)
        counts, stats = c.read_counts()
        assert list(stats) == orc.stats() and list(counts) == orc.counts()


def test_all_reads_miss_fills_the_queue(P):
    """a library unrelated to the reads: every quality-passing read goes through the LDS ring"""
    guides = P.binding.synth_library(1, 5000, 20)
    other = P.binding.synth_library(2, 5000, 20)
    with P.Counter(features=other, miss=2) as gen:
        fq = bytes(gen.synth_fastq(seed=3, n_reads=300000, read_len=40))
    orc = O.count_fastq_parallel(fq, 8, features=[(str(i), s) for i, s in enumerate(guides)], miss=2)
    with P.Counter(features=guides, miss=2) as c:
        c.count_block(fq)
        counts, stats = c.read_counts()
        assert list(stats) == orc.stats() and list(counts) == orc.counts()


@pytest.mark.parametrize("miss", [0, 1, 2])
def test_odd_symbols_in_the_window_gpu(P, miss):
    guides = synth.make_library(150, 12, 4242)
    fq = sprinkle_symbols(synth.make_fastq(synth.Spec(seed=miss, n_reads=20000, read_len=40, start=5, p_sub=0.3), guides), 9)
    for lib in (guides, guides[:100] + [g[:5] + "N" + g[6:] for g in guides[100:]]):
        kw = dict(miss=miss, length=12, start="5")
        o = O.Oracle(features=[(str(i), s) for i, s in enumerate(lib)], **kw)
        o.count_fastq(fq)
        with P.Counter(features=lib, **kw) as c:
            _, t = c.count_block(fq, want_timing=True)
            counts, stats = c.read_counts()
            assert list(stats) == o.stats() and list(counts) == o.counts()
            assert (t["general_reads"] == 0) == (lib is guides)


UP, DOWN = "GTTTAAGAGCTA", "CGTTACCAGGTT"


@pytest.mark.parametrize("mode", ["C", "EC"])
@pytest.mark.parametrize("anchors", ["both", "up", "down"])
@pytest.mark.parametrize("ms,qs", [(0, 30), (1, 30), (2, 12), (3, 41)])
def test_packed_anchor_kernel_vs_oracle(P, mode, anchors, ms, qs):
    guides = synth.make_library(300, 20, 900 + ms)
    spec = dict(seed=ms * 7 + qs, n_reads=30000, read_len=150, cassette=True, up=UP, down=DOWN, max_offset=110,
                p_sub=0.2, p_lowq=0.15, p_n=0.02)
    kw = dict(mode=mode, miss=1, length=20, miss_search_up=ms, miss_search_down=ms, qual_up=qs, qual_down=30)
    if anchors in ("both", "up"):
        kw["upstream"] = UP
    if anchors in ("both", "down"):
        kw["downstream"] = DOWN
    with P.Counter(features=guides, miss=1) as gen:
        fq = bytearray(gen.synth_fastq(**spec))
    import random
    rng = random.Random(5)
    pos = 0
    while True:                                   # damage some anchors / qualities in place (lines keep their length)
        a = fq.find(b"\n", pos) + 1
        b = fq.find(b"\n", a)
        if a <= 0 or b < 0:
            break
        qa = fq.find(b"\n", b + 1) + 1
        qb = fq.find(b"\n", qa)
        if rng.random() < 0.5:
            for _ in range(rng.randint(1, 3)):
                fq[a + rng.randrange(b - a)] = rng.choice(b"ACGT")
            fq[qa + rng.randrange(qb - qa)] = rng.choice(b"#+5:?")
        pos = qb + 1
    fq = bytes(fq)
    orc = O.count_fastq_parallel(fq, 8, features=[(str(i), s) for i, s in enumerate(guides)], **kw) if mode == "C" else None
    if mode == "EC":
        orc = O.Oracle(**kw)
        orc.count_fastq(fq)
    with P.Counter(features=guides if mode == "C" else None, **kw) as c:
        _, t = c.count_block(fq, want_timing=True)
        counts, stats = c.read_counts()
        assert list(stats) == orc.stats()
        assert t["fast_reads"] > 0.9 * t["reads"]
        if mode == "C":
            assert list(counts) == orc.counts()
        else:
            assert [(k, n) for k, n, _ in c.ec_results()] == list(zip(orc.keys(), orc.counts()))
        # the device-generated block of the undamaged spec agrees with its own host FASTQ
        c.reset()
        with P.Counter(features=guides, miss=1) as gen:
            clean_fq = bytes(gen.synth_fastq(**spec))
        o2 = O.Oracle(features=[(str(i), s) for i, s in enumerate(guides)] if mode == "C" else None, **kw)
        o2.count_fastq(clean_fq)
        blk = c.synth_create(guides=guides, **spec)
        c.count_resident(blk)
        counts2, stats2 = c.read_counts()
        assert list(stats2) == o2.stats()
        if mode == "C":
            assert list(counts2) == o2.counts()
        else:
            assert [(k, n) for k, n, _ in c.ec_results()] == list(zip(o2.keys(), o2.counts()))


def test_ec_table_growth_across_blocks(P):
    """many distinct keys over several blocks: both Extract+Count tables are re-hashed on the device"""
    guides = synth.make_library(60000, 20, 4)
    kw = dict(mode="EC", upstream=UP, downstream=DOWN)
    orc = O.Oracle(**kw)
    with P.Counter(**kw) as c:
        for blk in range(4):
            with P.Counter(features=guides, miss=1) as gen:
                fq = bytes(gen.synth_fastq(seed=50 + blk, n_reads=40000, read_len=150, cassette=True, up=UP, down=DOWN,
                                           p_rand=0.5))
            fq = sprinkle_symbols(fq, blk, rate=0.002)        # a few keys with odd symbols -> byte-string table
            orc.count_fastq(fq)
            c.count_block(fq)
        counts, stats = c.read_counts()
        assert list(stats) == orc.stats()
        assert [(k, n) for k, n, _ in c.ec_results()] == list(zip(orc.keys(), orc.counts()))


@pytest.mark.parametrize("step", ["256", "2048", "100000"])
@pytest.mark.parametrize("fixed", [False, True], ids=["anchored", "fixed_window"])
def test_ec_block_walked_in_steps(P, monkeypatch, step, fixed):
    """Extract+Count walks a block in steps of F2Q_EC_STEP reads (tables sized for one step beyond the keys they hold,
    grown by device rehash in between): tiny steps = many launches and many rehashes inside ONE block, same result,
    same first-seen order"""
    monkeypatch.setenv("F2Q_EC_STEP", step)
    monkeypatch.setenv("F2Q_NO_HOT", "1")                     # the stepped walk (the hot-key kernel takes whole views)
    guides = synth.make_library(5000, 20, 41)
    if fixed:
        kw = dict(mode="EC", start="10", length=20)
        fq = synth.make_fastq(synth.Spec(seed=77, n_reads=60000, read_len=80, start=10, p_rand=0.4), guides)
    else:
        kw = dict(mode="EC", upstream=UP, downstream=DOWN, miss_search_up=1)
        with P.Counter(features=guides, miss=1) as gen:
            fq = bytes(gen.synth_fastq(seed=9, n_reads=60000, read_len=150, cassette=True, up=UP, down=DOWN, p_rand=0.4))
    fq = sprinkle_symbols(fq, 5, rate=0.003)                  # raw records + byte-string keys in the same run
    orc = O.Oracle(**kw)
    orc.count_fastq(fq)
    with P.Counter(**kw) as c:
        _, t = c.count_block(fq, want_timing=True)
        _, stats = c.read_counts()
        assert list(stats) == orc.stats()
        assert [(k, n) for k, n, _ in c.ec_results()] == list(zip(orc.keys(), orc.counts()))
        assert t["launches"] >= 60000 // int(step)


@pytest.mark.parametrize("learn", ["256", "3000", "20000"])
@pytest.mark.parametrize("anchors", ["both", "up"])
def test_ec_hot_keys_vs_oracle(P, monkeypatch, learn, anchors):
    """anchored Extract+Count with the hot keys in LDS (k_extract_anchor_hot): a screen-like sample (300 guides carry most
    reads) over three blocks, the set learnt after F2Q_HOT_LEARN reads -- inside the first block, at a block boundary or
    spanning blocks.  Keys, counts and first-seen order against the oracle, and against the stepped kernel."""
    monkeypatch.setenv("F2Q_HOT_LEARN", learn)
    guides = synth.make_library(300, 20, 77)
    kw = dict(mode="EC", upstream=UP, miss_search_up=1)
    if anchors == "both":
        kw["downstream"] = DOWN
    else:
        kw["length"] = 20
    blocks = []
    with P.Counter(features=guides, miss=1) as gen:
        for k, n in enumerate((10000, 30000, 50000)):
            fq = bytes(gen.synth_fastq(seed=300 + k, n_reads=n, read_len=150, cassette=True, up=UP, down=DOWN, p_sub=0.15, p_rand=0.05))
            blocks.append(sprinkle_symbols(fq, k, rate=0.0003))
    orc = O.Oracle(**kw)
    res = []
    for no_hot in ("0", "1"):
        monkeypatch.setenv("F2Q_NO_HOT", no_hot)
        with P.Counter(**kw) as c:
            for fq in blocks:
                if no_hot == "0":
                    orc.count_fastq(fq)
                _, t = c.count_block(fq, want_timing=True)
                assert t["fast_reads"] > 0.9 * t["reads"]
            _, stats = c.read_counts()
            res.append((list(stats), c.ec_results()))
    assert res[0][0] == orc.stats() and res[1] == res[0]
    assert [(k, n) for k, n, _ in res[0][1]] == list(zip(orc.keys(), orc.counts()))
    assert max(n for _, n, _ in res[0][1]) > 100                  # there were hot keys


@pytest.mark.parametrize("learn,start,length,rl", [("256", 10, 20, 80), ("5000", 0, 29, 40), ("256", 3, 12, 14), ("1000000", 10, 20, 80)])
def test_ec_hot_keys_fixed_window(P, monkeypatch, learn, start, length, rl):
    """Extract+Count with a fixed window and the hot keys in LDS (k_extract_fixed4_hot) over three blocks: against the
    oracle and against the stepped kernel; then a block of all-new keys that fills the table (reads set aside and inserted
    after the table has grown, k_ec_deferred_fixed)"""
    monkeypatch.setenv("F2Q_HOT_LEARN", learn)
    guides = synth.make_library(300, length, 91)
    kw = dict(mode="EC", start=str(start), length=length)
    blocks = [sprinkle_symbols(synth.make_fastq(synth.Spec(seed=60 + k, n_reads=n, read_len=rl, start=start, p_sub=0.15, p_rand=0.05), guides), k, rate=0.0005)
              for k, n in enumerate((9000, 20000, 40000))]
    blocks.append(synth.make_fastq(synth.Spec(seed=70, n_reads=150000, read_len=rl, start=start, p_sub=0.0, p_rand=0.97), guides))
    orc = O.Oracle(**kw)
    res = []
    for no_hot in ("0", "1"):
        monkeypatch.setenv("F2Q_NO_HOT", no_hot)
        with P.Counter(**kw) as c:
            for fq in blocks:
                if no_hot == "0":
                    orc.count_fastq(fq)
                c.count_block(fq)
            _, stats = c.read_counts()
            res.append((list(stats), c.ec_results()))
    assert res[0][0] == orc.stats() and res[1] == res[0]
    assert [(k, n) for k, n, _ in res[0][1]] == list(zip(orc.keys(), orc.counts()))


@pytest.mark.parametrize("seed", range(int(os.environ.get("F2Q_EC_FUZZ_SEEDS", "10"))))     # one-off long runs: more seeds
def test_ec_hot_keys_fuzz(P, monkeypatch, seed):
    """random Extract+Count runs through the hot-key kernels: anchored or fixed window, learning threshold, number and size
    of blocks, share of repeated / novel / 'N' / long keys, block order -- keys, counts and first-seen order against the
    oracle (which sees the blocks in read-index order)"""
    import random
    rng = random.Random(1000 + seed)
    monkeypatch.setenv("F2Q_HOT_LEARN", str(rng.choice([64, 700, 5000, 40000])))
    anchored = rng.random() < 0.6
    glen = rng.choice([12, 20, 29]) if not anchored else rng.choice([16, 20, 33])
    guides = synth.make_library(rng.choice([40, 400, 3000]), glen, 500 + seed)
    if anchored:
        kw = dict(mode="EC", upstream=UP, miss_search_up=rng.choice([0, 1]))
        if rng.random() < 0.7:
            kw.update(downstream=DOWN, miss_search_down=kw["miss_search_up"])
        else:
            kw["length"] = glen
    else:
        kw = dict(mode="EC", start=str(rng.choice([0, 5, 16])), length=glen)
    blocks = []
    for k in range(rng.choice([1, 2, 4])):
        n = rng.choice([3000, 20000, 70000])
        p_rand = rng.choice([0.02, 0.3, 0.9])
        if anchored:
            spec = dict(seed=seed * 10 + k, n_reads=n if glen <= 31 else min(n, 20000), read_len=150 if glen < 30 else 120, cassette=True, up=UP, down=DOWN,
                        max_offset=60, p_sub=rng.choice([0.0, 0.2]), p_rand=p_rand, p_n=rng.choice([0.0, 0.05]))
            if glen <= 31:
                with P.Counter(features=guides, miss=1) as gen:
                    fq = bytes(gen.synth_fastq(**spec))
            else:
                fq = synth.make_fastq(synth.Spec(**spec), guides)      # (the device generator plants guides of <= 31 bases)
        else:
            fq = synth.make_fastq(synth.Spec(seed=seed * 10 + k, n_reads=n, read_len=int(kw["start"]) + glen + rng.choice([0, 7]),
                                             start=int(kw["start"]), p_sub=rng.choice([0.0, 0.2]), p_rand=p_rand, p_n=rng.choice([0.0, 0.05])), guides)
        blocks.append(sprinkle_symbols(fq, k, rate=rng.choice([0.0, 0.0004])))
    orc = O.Oracle(**kw)
    for fq in blocks:
        orc.count_fastq(fq)
    bases, b0 = [], 0
    for fq in blocks:
        bases.append(b0); b0 += fq.count(b"\n") // 4
    order = list(range(len(blocks)))
    if rng.random() < 0.5:
        rng.shuffle(order)                                          # blocks counted out of read-index order
    with P.Counter(**kw) as c:
        for i in order:
            c.set_read_base(bases[i])
            c.count_block(blocks[i])
        _, stats = c.read_counts()
        assert list(stats) == orc.stats()
        assert [(k, n) for k, n, _ in c.ec_results()] == list(zip(orc.keys(), orc.counts()))


def test_ec_hot_keys_table_fills_up(P, monkeypatch):
    """the table is sized for new keys at the rate seen while learning; a later block of all-new keys fills it, the
    inserts give up after F2Q_HOT_MAXPROBE slots and those reads are decided after the table has grown"""
    monkeypatch.setenv("F2Q_HOT_LEARN", "4096")
    guides = synth.make_library(300, 20, 78)
    kw = dict(mode="EC", upstream=UP, downstream=DOWN)
    with P.Counter(features=guides, miss=1) as gen:
        a = bytes(gen.synth_fastq(seed=1, n_reads=8192, read_len=150, cassette=True, up=UP, down=DOWN, p_sub=0.0, p_rand=0.0))
        b = bytes(gen.synth_fastq(seed=2, n_reads=400000, read_len=150, cassette=True, up=UP, down=DOWN, p_sub=0.0, p_rand=0.95))
    orc = O.Oracle(**kw)
    with P.Counter(**kw) as c:
        launches = 0
        for fq in (a, b):
            orc.count_fastq(fq)
            _, t = c.count_block(fq, want_timing=True)
            launches += t["launches"]
        _, stats = c.read_counts()
        assert list(stats) == orc.stats()
        assert [(k, n) for k, n, _ in c.ec_results()] == list(zip(orc.keys(), orc.counts()))
    assert launches >= 5                                           # learning, hot launch, deferred passes, raw reads


def test_ec_hot_keys_blocks_out_of_order(P, monkeypatch):
    """a block with LOWER read indices counted after the hot set was built: the snapshot of a hot key's first read sends
    such reads through the table's insert, which lowers the minimum -> same first-seen order as counting in order"""
    monkeypatch.setenv("F2Q_HOT_LEARN", "2048")
    guides = synth.make_library(200, 20, 79)
    kw = dict(mode="EC", upstream=UP, downstream=DOWN)
    with P.Counter(features=guides, miss=1) as gen:
        a = bytes(gen.synth_fastq(seed=11, n_reads=20000, read_len=150, cassette=True, up=UP, down=DOWN, p_sub=0.1))
        b = bytes(gen.synth_fastq(seed=12, n_reads=30000, read_len=150, cassette=True, up=UP, down=DOWN, p_sub=0.1))
    orc = O.Oracle(**kw)
    orc.count_fastq(a); orc.count_fastq(b)
    with P.Counter(**kw) as c:
        c.set_read_base(20000); c.count_block(b)
        c.set_read_base(0); c.count_block(a)
        _, stats = c.read_counts()
        assert list(stats) == orc.stats()
        assert [(k, n) for k, n, _ in c.ec_results()] == list(zip(orc.keys(), orc.counts()))


@pytest.mark.parametrize("anchors", ["both", "up", "down"])
def test_packed_anchor_with_odd_symbols_gpu(P, anchors):
    guides = synth.make_library(120, 16, 31337)
    spec = synth.Spec(seed=4, n_reads=20000, read_len=120, cassette=True, up=UP, down=DOWN, max_offset=70, p_sub=0.25)
    fq = sprinkle_symbols(synth.make_fastq(spec, guides), 11, rate=0.01)
    kw = dict(miss=2, length=16, miss_search_up=1, miss_search_down=1)
    if anchors in ("both", "up"):
        kw["upstream"] = UP
    if anchors in ("both", "down"):
        kw["downstream"] = DOWN
    for mode in ("C", "EC"):
        o = O.Oracle(features=[(str(i), s) for i, s in enumerate(guides)] if mode == "C" else None, mode=mode, **kw)
        o.count_fastq(fq)
        with P.Counter(features=guides if mode == "C" else None, mode=mode, **kw) as c:
            c.count_block(fq)
            counts, stats = c.read_counts()
            assert list(stats) == o.stats()
            if mode == "C":
                assert list(counts) == o.counts()
            else:
                assert [(k, n) for k, n, _ in c.ec_results()] == list(zip(o.keys(), o.counts()))


@pytest.mark.parametrize("host_pack", ["0", "1"], ids=["device_packer", "host_packer"])
def test_both_packers_agree_with_the_oracle(P, monkeypatch, host_pack):
    """the device ingest (k_nl_count .. k_pack) and the host packer produce the same counts on awkward text:
    CRLF, blank-line shift, unterminated last line, partial record, ragged lengths, odd symbols"""
    monkeypatch.setenv("F2Q_HOST_PACK", host_pack)
    guides = synth.make_library(100, 20, 5150)
    body = synth.make_fastq(synth.Spec(seed=1, n_reads=3000, read_len=70), guides)
    ragged = b"".join(synth.make_fastq(synth.Spec(seed=2 + k, n_reads=200, read_len=rl), guides) for k, rl in enumerate((19, 20, 21, 33, 64, 65)))
    texts = {
        "plain": body, "crlf": body.replace(b"\n", b"\r\n"), "no_final_newline": body[:-1],
        "partial_record": body + b"@x\nACGT\n+\n", "blank_shift": b"\n" + body, "ragged": ragged,
        "symbols": sprinkle_symbols(body, 3, rate=0.02), "trailing_ws": body.replace(b"I\n", b"I \t\n", 500),
        "one_record": b"@r\n" + guides[0].encode() + b"\n+\n" + b"I" * 20 + b"\n", "empty": b"",
    }
    for name, fq in texts.items():
        for kw in (dict(miss=1), dict(miss=1, upstream="ACGT", length=12), dict(mode="EC", start="3", length=17)):
            feats = None if kw.get("mode") == "EC" else guides
            o = O.Oracle(features=[(str(i), s) for i, s in enumerate(guides)] if feats else None, **kw)
            used_o = o.count_fastq(fq)
            with P.Counter(features=feats, **kw) as c:
                used = c.count_block(fq)
                counts, stats = c.read_counts()
                assert used == used_o, (name, kw)
                assert list(stats) == o.stats(), (name, kw)
                if feats:
                    assert list(counts) == o.counts(), (name, kw)
                else:
                    assert [(k, n) for k, n, _ in c.ec_results()] == list(zip(o.keys(), o.counts())), (name, kw)


def test_resident_block_from_host_fastq(P):
    """f2q_block_from_fastq keeps a host FASTQ resident in the tile layout; counting it repeatedly accumulates"""
    guides = synth.make_library(500, 20, 8)
    fq = synth.make_fastq(synth.Spec(seed=14, n_reads=20000, read_len=101), guides)
    o = O.Oracle(features=[(str(i), s) for i, s in enumerate(guides)], miss=2)
    o.count_fastq(fq)
    with P.Counter(features=guides, miss=2) as c:
        blk = c.block_from_fastq(fq)
        assert blk.info()["n_reads"] == 20000
        for k in (1, 2, 3):
            c.count_resident(blk)
            counts, stats = c.read_counts()
            assert list(stats) == [k * v for v in o.stats()] and list(counts) == [k * v for v in o.counts()]
        blk.free()
        empty = c.block_from_fastq(b"")
        assert empty.info()["n_reads"] == 0
        c.count_resident(empty)
        empty.free()


def _full(P, monkeypatch, env, lib, spec, **kw):
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    with P.Counter(features=lib if kw.get("mode", "C") == "C" else None, **kw) as c:
        blk = c.synth_create(guides=lib, **spec)
        t = c.count_resident(blk)
        counts, stats = c.read_counts()
        ec = c.ec_results() if kw.get("mode") == "EC" else None
        blk.free()
    for k in env:
        monkeypatch.delenv(k)
    return list(counts), list(stats), ec, t


def test_full_size_config3_cross_paths(P, monkeypatch):
    """BASELINE config 3 at full size (50M x 150 bp, 10k guides, --m 1): the 4-reads-per-lane kernel, the
    one-read-per-lane kernel and the byte-exact general kernel (validated against the oracle at small sizes)
    must agree bit for bit; plus the size-independent identities."""
    lib = P.binding.synth_library(0xF2A5 + 3, 10000, 20)
    spec = dict(seed=0xBEEF, n_reads=50_000_000, read_len=150)
    kw = dict(miss=1, phred=30, length=20, start="0")
    c2, s2, _, t2 = _full(P, monkeypatch, {}, lib, spec, **kw)                       # the library-in-LDS kernel
    c1, s1, _, _ = _full(P, monkeypatch, {"F2Q_FORCE_V1": "1"}, lib, spec, **kw)
    c3, s3, _, _ = _full(P, monkeypatch, {"F2Q_NO_LT": "1"}, lib, spec, **kw)       # the pigeonhole kernel on L2 tables
    assert s2[0] == 50_000_000 and s2[0] == sum(s2[1:]) and sum(c2) == s2[1] + s2[2]
    assert t2["general_reads"] == 0
    assert (c1, s1) == (c2, s2) and (c3, s3) == (c2, s2)
    spec_q = dict(spec, n_reads=12_500_000)                       # a quarter through the general kernel (byte-wise, slow)
    cg, sg, _, tg = _full(P, monkeypatch, {"F2Q_FORCE_GENERAL": "1"}, lib, spec_q, **kw)
    cq, sq, _, _ = _full(P, monkeypatch, {}, lib, spec_q, **kw)
    assert tg["general_reads"] == 12_500_000 and (cg, sg) == (cq, sq)


def test_full_size_config5_cross_paths(P, monkeypatch):
    """BASELINE config 5 (anchored, Counter and Extract+Count) at 50M reads: invariants, and the packed
    bit-plane kernel against the byte-exact general kernel on a 10M-read slice of the same stream."""
    lib = P.binding.synth_library(0xF2A5 + 5, 10000, 20)
    up, down = "GTTTAAGAGCTA", "CGTTACCAGGTT"
    spec = dict(seed=0xBEEF, n_reads=50_000_000, read_len=150, cassette=True, up=up, down=down, max_offset=100)
    kw = dict(miss=1, upstream=up, downstream=down, miss_search_up=1, miss_search_down=1)
    c5, s5, _, t5 = _full(P, monkeypatch, {}, lib, spec, **kw)
    assert s5[0] == 50_000_000 and s5[0] == sum(s5[1:]) and sum(c5) == s5[1] + s5[2] and t5["general_reads"] == 0
    _, se, ec, _ = _full(P, monkeypatch, {}, lib, spec, mode="EC", **{k: v for k, v in kw.items() if k != "miss"})
    assert se[0] == 50_000_000 and se[1] + se[4] == se[0] and sum(n for _, n, _ in ec) == se[1]
    assert se[4] == s5[4]                                         # the same reads fail quality in both modes
    by_key = {k: n for k, n, _ in ec}
    # a read counted as a PERFECT hit of guide g in Counter mode is a read whose extracted key is g in EC mode
    assert sum(by_key.get(g, 0) for g in lib) == s5[1]
    sl = dict(spec, n_reads=10_000_000)
    cp, sp, _, _ = _full(P, monkeypatch, {}, lib, sl, **kw)
    cg, sg, _, tg = _full(P, monkeypatch, {"F2Q_FORCE_GENERAL": "1"}, lib, sl, **kw)
    assert tg["general_reads"] == 10_000_000 and (cp, sp) == (cg, sg)
    _, sp2, ecp, _ = _full(P, monkeypatch, {}, lib, sl, mode="EC", **{k: v for k, v in kw.items() if k != "miss"})
    _, sg2, ecg, _ = _full(P, monkeypatch, {"F2Q_FORCE_GENERAL": "1"}, lib, sl, mode="EC", **{k: v for k, v in kw.items() if k != "miss"})
    assert sp2 == sg2 and ecp == ecg
    _, sn2, ecn, _ = _full(P, monkeypatch, {"F2Q_NO_HOT": "1"}, lib, sl, mode="EC", **{k: v for k, v in kw.items() if k != "miss"})
    assert sp2 == sn2 and ecp == ecn                              # hot keys in LDS == the stepped kernel


def test_full_size_config4_per_gpu(P, monkeypatch):
    """BASELINE config 4, the per-GPU share (50M x 150 bp reads vs 100k x 20 bp guides, --m 1): the large-library
    path (feature index per read -> range-partitioned LDS histograms, tables beyond one XCD's L2) against the oracle
    at 200k reads, against the byte-exact general kernel on a 5M-read slice, and the size-independent identities
    (5-counter identity, sum of counts, shard-sum invariance) at full size."""
    lib = P.binding.synth_library(0xF2A5 + 4, 100000, 20)
    kw = dict(miss=1, phred=30, length=20, start="0")
    small = dict(seed=0xBEEF, n_reads=200_000, read_len=150)
    with P.Counter(features=lib, **kw) as c:
        fq = bytes(c.synth_fastq(**small))
    orc = O.count_fastq_parallel(fq, 16, features=[(str(i), s) for i, s in enumerate(lib)], **kw)
    cs, ss, _, ts = _full(P, monkeypatch, {"F2Q_PT_MIN_READS": "0"}, lib, small, **kw)       # the partitioned kernels on the small block too
    assert ts["general_reads"] == 0 and ss == orc.stats() and cs == orc.counts()
    ck, sk, _, tk = _full(P, monkeypatch, {}, lib, small, **kw)                               # (a block this small: the packed-table kernel)
    assert tk["path"] == 2 and (ck, sk) == (cs, ss)
    sl = dict(small, n_reads=5_000_000)
    cp, sp, _, _ = _full(P, monkeypatch, {}, lib, sl, **kw)
    cg, sg, _, tg = _full(P, monkeypatch, {"F2Q_FORCE_GENERAL": "1"}, lib, sl, **kw)
    assert tg["general_reads"] == 5_000_000 and (cp, sp) == (cg, sg)
    full = dict(small, n_reads=50_000_000)
    c4, s4, _, t4 = _full(P, monkeypatch, {}, lib, full, **kw)
    assert t4["general_reads"] == 0 and t4["path"] == 4 and ts["path"] == 4       # the partitioned kernels (k_part_*)
    c4b, s4b, _, t4b = _full(P, monkeypatch, {"F2Q_NO_PT": "1"}, lib, full, **kw)  # ... against the packed tables in L2
    assert t4b["path"] == 2 and (c4b, s4b) == (c4, s4)
    assert s4[0] == 50_000_000 and s4[0] == sum(s4[1:]) and sum(c4) == s4[1] + s4[2]
    # shard-sum invariance: 8 consecutive shards of the same stream (what 8 ranks would count) add up to the whole
    with P.Counter(features=lib, **kw) as c:
        for r in range(8):
            b = c.synth_create(first_read=r * 6_250_000, **dict(full, n_reads=6_250_000))
            c.count_resident(b)
            b.free()
        c8, s8 = c.read_counts()
    assert list(s8) == s4 and list(c8) == c4


@pytest.mark.parametrize("miss", [0, 1])
@pytest.mark.parametrize("glen,n_guides,start,rl", [(20, 10000, 0, 150), (20, 13000, 3, 60), (14, 800, 5, 40), (21, 6000, 2, 75), (17, 2000, 9, 150)])
def test_lds_table_kernel_vs_oracle(P, monkeypatch, miss, glen, n_guides, start, rl):
    """k_count_fixed4_lds (library in LDS: cuckoo tag tables + u16 histogram) against the oracle and against the
    pigeonhole kernel (F2Q_NO_LT=1) on dense mutations, N symbols, near-duplicate features and clipped reads"""
    guides = P.binding.synth_library(31 * glen + n_guides, n_guides, glen)
    twins = []
    for i, g in enumerate(guides[:300]):
        p = (i * 7) % glen
        twins.append(g[:p] + "ACGT"[("ACGT".index(g[p]) + 1 + i % 3) % 4] + g[p + 1:])
    lib = list(dict.fromkeys(guides + twins))
    kw = dict(miss=miss, length=glen, start=str(start))
    spec = dict(seed=glen + miss, n_reads=120000, read_len=rl, start=start, p_sub=0.35, p_rand=0.1, p_n=0.08, p_lowq=0.1)
    with P.Counter(features=lib, **kw) as c:
        fq = bytes(c.synth_fastq(**spec))
        fq = sprinkle_symbols(fq, 3, rate=0.01) + bytes(c.synth_fastq(**dict(spec, n_reads=300, read_len=start + glen - 2)))
        orc = O.count_fastq_parallel(fq, 8, features=[(str(i), s) for i, s in enumerate(lib)], **kw)
        _, t = c.count_block(fq, want_timing=True)
        counts, stats = c.read_counts()
        assert list(stats) == orc.stats() and list(counts) == orc.counts()
        assert t["general_reads"] == 0
    monkeypatch.setenv("F2Q_NO_LT", "1")
    with P.Counter(features=lib, **kw) as c:
        c.count_block(fq)
        counts2, stats2 = c.read_counts()
    monkeypatch.delenv("F2Q_NO_LT")
    assert list(stats2) == list(stats) and list(counts2) == list(counts)


@pytest.mark.parametrize("miss", [0, 1, 2])
@pytest.mark.parametrize("starts,length,rl", [("0,10", 10, 40), ("3,20,9", 6, 33), ("12,0", 15, 30), ("0,7,14,21", 6, 31), ("5,5", 8, 20)])
def test_multi_window_packed_kernel_vs_oracle(P, monkeypatch, miss, starts, length, rl):
    """k_count_multi4 (--st a,b,... on packed tiles, k-part feature tables) against the oracle and against the byte-exact
    general kernel; runs whose joint key does not fit the packed slot must stay on the general path"""
    from test_lane_logic_cpu import multi_window_case
    lib, fq, W = multi_window_case(starts, length, rl, miss, n_reads=40000)
    kw = dict(miss=miss, length=length, start=starts)
    o = O.Oracle(features=[(str(i), s) for i, s in enumerate(lib)], **kw)
    o.count_fastq(fq)
    with P.Counter(features=lib, **kw) as c:
        _, t = c.count_block(fq, want_timing=True)
        counts, stats = c.read_counts()
    assert list(stats) == o.stats() and list(counts) == o.counts()
    packed = 2 * W * length + len(lib).bit_length() <= 64
    # (the tiles hold the windows only: a read that ends inside a window goes to the byte-exact kernel)
    short = sum(len(x) < max(int(v) for v in starts.split(",")) + length for x in fq.split(b"\n")[1::4])
    assert (t["general_reads"] == short and short > 100) if packed else (t["fast_reads"] == 0)
    monkeypatch.setenv("F2Q_FORCE_GENERAL", "1")
    with P.Counter(features=lib, **kw) as c:
        _, t2 = c.count_block(fq, want_timing=True)
        counts2, stats2 = c.read_counts()
    monkeypatch.delenv("F2Q_FORCE_GENERAL")
    assert t2["fast_reads"] == 0 and list(stats2) == list(stats) and list(counts2) == list(counts)


@pytest.mark.parametrize("miss", [0, 1])
@pytest.mark.parametrize("starts,length,rl", [("0,10", 10, 40), ("3,40,21", 6, 60), ("5,100", 8, 150), ("0,7,14,21", 5, 31), ("60,2", 9, 75), ("0,20", 7, 30)])
@pytest.mark.parametrize("combo", [False, True], ids=["pairs", "combinatorial"])
def test_multi_window_lds_kernel_vs_oracle(P, monkeypatch, miss, starts, length, rl, combo):
    """k_count_fixed4_lds<.., MW>: several windows (also far apart: the tiles hold the windows back to back), every feature
    with one part per window, joined keys in the LDS tables, Phred rule part by part -- against the oracle, the k-part
    packed tables (k_count_multi4, F2Q_NO_LT=1) and the byte-exact general kernel"""
    from test_lane_logic_cpu import multi_window_uniform_case
    lib, fq = multi_window_uniform_case(starts, length, rl, n_reads=40000, seed=1)
    if combo:            # one guide with many partners: two-window keys are looked up in their mixed form (mw_mix)
        parts0 = [f.split(":") for f in lib[:60]]
        lib = list(dict.fromkeys(lib + [":".join([parts0[i % 6][0]] + p[1:]) for i, p in enumerate(parts0[6:60])] +
                                 [":".join(p[:-1] + [parts0[i % 4][-1]]) for i, p in enumerate(parts0[4:50])]))
    kw = dict(miss=miss, length=length, start=starts)
    o = O.count_fastq_parallel(fq, 8, features=[(str(i), s) for i, s in enumerate(lib)], **kw)
    short = sum(len(x) < max(int(v) for v in starts.split(",")) + length for x in fq.split(b"\n")[1::4])
    res = []
    for env in ({}, {"F2Q_NO_LT": "1"}, {"F2Q_FORCE_GENERAL": "1"}, {"F2Q_HOST_PACK": "1"}, {"F2Q_GENERIC": "1"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        with P.Counter(features=lib, **kw) as c:
            _, t = c.count_block(fq, want_timing=True)
            counts, stats = c.read_counts()
        for k in env:
            monkeypatch.delenv(k)
        assert list(stats) == o.stats() and list(counts) == o.counts(), env
        res.append((t["path"], t["general_reads"]))
    assert [r[0] for r in res[:2]] == [10, 5] and res[3][0] == 10 and res[4][0] == 10       # F2Q_PATH_MULTI_LDS, F2Q_PATH_MULTI
    assert res[0][1] == short and res[1][1] == short and res[2][1] == 40000


@pytest.mark.parametrize("miss", [0, 1, 3])
@pytest.mark.parametrize("glen", [12, 40])
def test_general_key_index_kernel_vs_oracle(P, miss, glen):
    """k_count_general over the byte-string index (GkDesc): a library with odd symbols, three lengths and near-duplicates,
    and 40-base windows that no 2-bit table holds"""
    from test_lane_logic_cpu import general_key_case
    lib, fq = general_key_case(miss, glen, n=4000, n_reads=30000)
    kw = dict(miss=miss, length=glen, start="6")
    o = O.Oracle(features=[(str(i), s) for i, s in enumerate(lib)], **kw)
    o.count_fastq(fq)
    with P.Counter(features=lib, **kw) as c:
        c.count_block(fq)
        counts, stats = c.read_counts()
    assert list(stats) == o.stats() and list(counts) == o.counts()
    assert stats[1] > 0 and (miss == 0 or stats[2] > 0)


@pytest.mark.parametrize("mode", ["C", "EC"])
@pytest.mark.parametrize("ms,three", [(0, False), (1, False), (1, True)])
def test_multi_pair_packed_kernel_vs_oracle(P, monkeypatch, mode, ms, three):
    """k_count_anchor_pairs (several --us/--ds pairs on the planes, joined keys matched as strings) against the oracle
    and against the byte-exact general kernel"""
    from test_lane_logic_cpu import multi_pair_case
    lib, fq, ups, downs = multi_pair_case(30000, 21 + ms + (10 if three else 0), n_guides=400, three=three)
    kw = dict(mode=mode, miss=1, upstream=",".join(ups), downstream=",".join(downs), miss_search_up=ms, miss_search_down=ms)
    o = O.Oracle(features=[(str(i), s) for i, s in enumerate(lib)] if mode == "C" else None, **kw)
    o.count_fastq(fq)
    res = []
    for env in ("0", "1"):
        monkeypatch.setenv("F2Q_FORCE_GENERAL", env)
        with P.Counter(features=lib if mode == "C" else None, **kw) as c:
            _, t = c.count_block(fq, want_timing=True)
            counts, stats = c.read_counts()
            assert (t["fast_reads"] > 0.8 * t["reads"]) if env == "0" else (t["fast_reads"] == 0)
            res.append((list(stats), list(counts) if mode == "C" else [(k, n) for k, n, _ in c.ec_results()]))
    monkeypatch.delenv("F2Q_FORCE_GENERAL")
    assert res[0] == res[1] and res[0][0] == o.stats()
    assert res[0][1] == (o.counts() if mode == "C" else list(zip(o.keys(), o.counts())))


@pytest.mark.parametrize("miss", [0, 1])
@pytest.mark.parametrize("la,lb,combinatorial,ms", [(18, 18, False, 1), (20, 20, True, 0), (10, 12, False, 1), (7, 20, True, 1), (14, 9, False, 2)])
def test_pair_tables_kernel_vs_oracle(P, monkeypatch, miss, la, lb, combinatorial, ms):
    """k_count_anchor_pairs with the pair tables (two --us/--ds pairs, a pure A:B library: the joined key as two 2-bit words)
    against the oracle, the string index (F2Q_NO_PW=1) and the byte-exact general kernel"""
    from test_lane_logic_cpu import pair_library_case, UP, UP2, DOWN, DOWN2
    lib, fq = pair_library_case(30000, 3 * la + lb + ms + 1, la, lb, combinatorial)
    kw = dict(miss=miss, upstream=f"{UP},{UP2}", downstream=f"{DOWN},{DOWN2}", miss_search_up=ms, miss_search_down=ms)
    o = O.count_fastq_parallel(fq, 8, features=[(str(i), s) for i, s in enumerate(lib)], **kw)
    res = []
    for env in ({}, {"F2Q_NO_PW": "1"}, {"F2Q_FORCE_GENERAL": "1"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        with P.Counter(features=lib, **kw) as c:
            _, t = c.count_block(fq, want_timing=True)
            counts, stats = c.read_counts()
        for k in env:
            monkeypatch.delenv(k)
        assert list(stats) == o.stats() and list(counts) == o.counts(), env
        res.append((t["path"], t["general_reads"], t["kernel_ms"]))
    assert res[0][0] == 8 and res[1][0] == 8 and res[0][1] < 3000 and res[2][1] == 30000        # F2Q_PATH_PAIRS


def test_two_window_full_size(P, monkeypatch):
    """50M reads, two 10-base windows (--st 0,10 --l 10) against 10k two-part features: the packed multi-window kernel
    against the single-window run of the 20-base form of the same library on the same reads (same counts: see below),
    the byte-exact general kernel on a 5M slice, and the identities"""
    guides = P.binding.synth_library(0xF2A5 + 3, 10000, 20)
    lib2 = [g[:10] + ":" + g[10:] for g in guides]
    spec = dict(seed=0xBEEF, n_reads=50_000_000, read_len=150)
    kw = dict(miss=1, phred=30, length=10, start="0,10")
    with P.Counter(features=lib2, **kw) as c:
        blk = c.synth_create(guides=guides, **spec)
        t2 = c.count_resident(blk)
        c2, s2 = c.read_counts()
        blk.free()
    assert t2["general_reads"] == 0 and s2[0] == 50_000_000 and s2[0] == sum(s2[1:]) and c2.sum() == s2[1] + s2[2]
    assert t2["path"] == 10                                      # the joined keys on the library-in-LDS kernel
    monkeypatch.setenv("F2Q_NO_LT", "1")                         # ... and on the k-part packed tables (k_count_multi4)
    with P.Counter(features=lib2, **kw) as c:
        blk = c.synth_create(guides=guides, **spec)
        t3 = c.count_resident(blk)
        c3, s3 = c.read_counts()
        blk.free()
    monkeypatch.delenv("F2Q_NO_LT")
    assert t3["path"] == 5 and list(c3) == list(c2) and list(s3) == list(s2)
    c1, s1, _, _ = _full(P, monkeypatch, {}, guides, spec, miss=1, phred=30, length=20, start="0")
    # both parts pass <=> the 20-base window passes, and then the verdicts are the same (the ':' sits between the halves);
    # a read with one failed part yields a one-part key, and the library has no one-part feature
    assert c1 == list(c2) and s1[1:3] == list(s2[1:3]) and s1[3] + s1[4] == s2[3] + s2[4] and s2[4] < s1[4]
    sl = dict(spec, n_reads=5_000_000)
    res = []
    for env in ({}, {"F2Q_FORCE_GENERAL": "1"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        with P.Counter(features=lib2, **kw) as c:
            blk = c.synth_create(guides=guides, **sl)
            t = c.count_resident(blk)
            cc, ss = c.read_counts()
            res.append((list(cc), list(ss), t["general_reads"]))
            blk.free()
        for k in env:
            monkeypatch.delenv(k)
    assert res[0][:2] == res[1][:2] and res[0][2] == 0 and res[1][2] == 5_000_000


def test_lds_histogram_overflow_protocol(P, monkeypatch):
    """two guides take 40M reads: every workgroup's u16 counters pass 0x8000 several times and hand the surplus to the
    global vector; the result must equal the pigeonhole kernel's (u32 histogram) bit for bit"""
    lib = P.binding.synth_library(77, 2, 20)
    spec = dict(seed=9, n_reads=40_000_000, read_len=150)
    kw = dict(miss=1, phred=30, length=20, start="0")
    ca, sa, _, _ = _full(P, monkeypatch, {}, lib, spec, **kw)
    cb, sb, _, _ = _full(P, monkeypatch, {"F2Q_NO_LT": "1"}, lib, spec, **kw)
    assert (ca, sa) == (cb, sb) and sa[0] == 40_000_000 and sum(ca) == sa[1] + sa[2] and min(ca) > 10_000_000


def test_range_histogram_with_popular_features(P, monkeypatch):
    """a library too large for a per-workgroup histogram (hit_buf + k_hist_ranges) with almost every read on two features
    of different range passes, against the general kernel"""
    lib = P.binding.synth_library(77, 30000, 20)
    spec = dict(seed=5, n_reads=12_000_000, read_len=60, p_sub=0.02, p_rand=0.0, p_n=0.0)
    res = []
    for env in ({}, {"F2Q_FORCE_GENERAL": "1"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        with P.Counter(features=lib, miss=1, length=20, start="0") as c:
            blk = c.synth_create(guides=[lib[3], lib[29999]], **spec)
            t = c.count_resident(blk)
            counts, stats = c.read_counts()
            blk.free()
        for k in env:
            monkeypatch.delenv(k)
        res.append((list(counts), list(stats), t["general_reads"]))
    assert res[0][2] == 0 and res[1][2] == 12_000_000 and res[0][:2] == res[1][:2]
    assert res[0][0][3] > 5_000_000 and res[0][0][29999] > 5_000_000 and sum(res[0][0]) == res[0][1][1] + res[0][1][2]


@pytest.mark.parametrize("mode", ["C", "EC"])
def test_queued_steps_match_waited_steps(P, mode):
    """f2q_count_resident_queued: steps queued back to back (what bench.py times) give what the waited calls give, and
    every step comes back with its own kernel time"""
    guides = synth.make_library(500, 20, 5)
    kw = dict(features=guides, miss=1) if mode == "C" else dict(mode="EC", start="0", length=20)
    with P.Counter(**kw) as c:
        blk = c.synth_create(guides=guides, seed=9, n_reads=300_000) if mode == "EC" else c.synth_create(seed=9, n_reads=300_000)
        c.reset(); c.count_resident(blk); c.count_resident(blk); c.count_resident(blk)
        want = [list(x) for x in c.read_counts()] + [c.ec_results() if mode == "EC" else None]
        assert c.queued_times() == []
        c.reset()
        for _ in range(3):
            c.count_resident_queued(blk)
        times = c.queued_times()
        assert len(times) == 3 and all(t > 0 for t in times) and c.queued_times() == []
        assert [list(x) for x in c.read_counts()] + [c.ec_results() if mode == "EC" else None] == want
        blk.free()


def test_library_is_built_from_this_tree(P):
    """the .so the tests load carries the hash of the sources in this tree (no stale binary)"""
    import __graft_entry__ as g
    assert P.binding.build_id() == g.source_id()


@pytest.mark.parametrize("start,length,rl", [(0, 20, 150), (7, 29, 60), (3, 12, 14), (10, 8, 9), (0, 0, 30), (5, 30, 80)])
def test_extract_count_fixed_window_gpu(P, start, length, rl):
    guides = synth.make_library(60, max(8, min(length, 20)), 777)
    fq = sprinkle_symbols(synth.make_fastq(synth.Spec(seed=start + length, n_reads=20000, read_len=rl, start=min(start, rl - 1),
                                                       p_lowq=0.2), guides), 2, rate=0.01)
    fq += synth.make_fastq(synth.Spec(seed=1, n_reads=200, read_len=max(1, start)), guides)
    kw = dict(mode="EC", start=str(start), length=length)
    o = O.Oracle(**kw)
    o.count_fastq(fq)
    with P.Counter(**kw) as c:
        _, t = c.count_block(fq, want_timing=True)
        _, stats = c.read_counts()
        assert list(stats) == o.stats()
        assert [(k, n) for k, n, _ in c.ec_results()] == list(zip(o.keys(), o.counts()))
        assert (t["fast_reads"] > 0) == (length <= 29)


@pytest.mark.parametrize("hot", ["1", "0"], ids=["hot_keys", "stepped"])
@pytest.mark.parametrize("host_pack", ["0", "1"], ids=["device_packer", "host_packer"])
@pytest.mark.parametrize("start,length,rl,rate", [(0, 20, 150, 0.03), (5, 26, 40, 0.04), (2, 29, 31, 0.02), (4, 12, 14, 0.1)])
def test_extract_count_fixed_window_n_reads_stay_packed_gpu(P, monkeypatch, start, length, rl, rate, host_pack, hot):
    """Extract+Count with --st/--l: windows with up to three 'N' / 'n' keep to the tiles (single-word keys spelt from the
    flag bits) in k_extract_fixed4_hot, the stepped k_extract_fixed4 and, with a table that fills up, k_ec_deferred_fixed"""
    def fits(nn, n):
        return n <= 29 and nn <= 3 and 2 * n + 2 + 5 * nn <= 58
    monkeypatch.setenv("F2Q_HOST_PACK", host_pack)
    if hot == "0":
        monkeypatch.setenv("F2Q_NO_HOT", "1")
    else:
        monkeypatch.setenv("F2Q_HOT_LEARN", "2000")
    guides = synth.make_library(60, max(8, min(length, 20)), 778)
    fq = sprinkle_symbols(synth.make_fastq(synth.Spec(seed=start + length, n_reads=30000, read_len=rl, start=start, p_lowq=0.1), guides),
                          5, rate=rate, symbols=b"Nn")
    kw = dict(mode="EC", start=str(start), length=length)
    o = O.Oracle(**kw)
    o.count_fastq(fq)
    want_gen = 0
    for seq in fq.split(b"\n")[1::4]:
        w = seq[start:start + length].upper()
        want_gen += bool(w.count(b"N")) and not fits(w.count(b"N"), len(w))
    with P.Counter(**kw) as c:
        lines = fq.split(b"\n")
        cut = 4 * (len(lines) // 8)                                      # two blocks: the table grows between them
        for part in (b"\n".join(lines[:cut]) + b"\n", b"\n".join(lines[cut:])):
            c.count_block(part)
        _, stats = c.read_counts()
        assert list(stats) == o.stats()
        assert [(k, n) for k, n, _ in c.ec_results()] == list(zip(o.keys(), o.counts()))
    with P.Counter(**kw) as c:
        _, t = c.count_block(fq, want_timing=True)
        assert t["general_reads"] == want_gen and t["fast_reads"] == 30000 - want_gen
    assert any("N" in k for k in o.keys())


def test_fuzz_kernels_vs_oracle(P):
    """2000 seeded random cases (tests/fuzz_cases.py) through the C ABI: device packer + every kernel family"""
    from fuzz_cases import make_case
    fast = general = 0
    for seed in range(2000):
        kw, feats, fq = make_case(seed)
        o = O.Oracle(features=[(str(i), s) for i, s in enumerate(feats)] if feats is not None else None, **kw)
        used_o = o.count_fastq(fq)
        with P.Counter(features=feats, **kw) as c:
            used, t = c.count_block(fq, want_timing=True)
            counts, stats = c.read_counts()
            assert used == used_o, (seed, kw)
            assert list(stats) == o.stats(), (seed, kw)
            if feats is not None:
                assert list(counts) == o.counts(), (seed, kw)
            else:
                assert [(k, n) for k, n, _ in c.ec_results()] == list(zip(o.keys(), o.counts())), (seed, kw)
            fast += t["fast_reads"]; general += t["general_reads"]
    assert fast > 1000 and general > 1000


PATH_FIXED_PACKED, PATH_FIXED_LDS, PATH_FIXED_PART = 2, 3, 4


@pytest.mark.parametrize("miss", [0, 1])
@pytest.mark.parametrize("glen,n_guides,start,rl,parts,chunk", [(20, 10000, 0, 150, 3, ""), (20, 30000, 3, 60, 0, "20000"), (21, 6000, 2, 75, 5, "7000"),
                                                                  (17, 2000, 9, 150, 2, ""), (18, 20000, 0, 40, 0, ""), (20, 6000, 0, 150, 12, "50000"),
                                                                  (20, 6000, 5, 150, 20, "")])
def test_partitioned_kernels_vs_oracle(P, monkeypatch, miss, glen, n_guides, start, rl, parts, chunk):
    """k_part_scatter / k_part_count / k_part_hist1 / k_part_reduce (a library dealt into LDS-sized partitions: BASELINE
    config 4's path) against the oracle and against the packed-table kernel (F2Q_NO_PT=1): libraries beyond one
    workgroup's LDS and small ones forced into 2...20 partitions (16, 8 and 4 scatter waves per workgroup), blocks
    counted in one round and in several (F2Q_PT_CHUNK), dense mutations, N symbols, near-duplicates, clipped reads"""
    guides = P.binding.synth_library(77 * glen + n_guides, n_guides, glen)
    twins = []
    for i, g in enumerate(guides[:600]):
        p = (i * 7) % glen
        twins.append(g[:p] + "ACGT"[("ACGT".index(g[p]) + 1 + i % 3) % 4] + g[p + 1:])
    lib = list(dict.fromkeys(guides + twins))
    kw = dict(miss=miss, length=glen, start=str(start))
    spec = dict(seed=glen + miss + 9, n_reads=150000, read_len=rl, start=start, p_sub=0.35, p_rand=0.1, p_n=0.08, p_lowq=0.1)
    if parts:
        monkeypatch.setenv("F2Q_PT_PARTS", str(parts))
    monkeypatch.setenv("F2Q_PT_MIN_READS", "0")                  # (blocks this small keep the packed-table kernel otherwise)
    if chunk:
        monkeypatch.setenv("F2Q_PT_CHUNK", chunk)
    with P.Counter(features=lib, **kw) as c:
        fq = bytes(c.synth_fastq(**spec))
        fq = sprinkle_symbols(fq, 3, rate=0.01) + bytes(c.synth_fastq(**dict(spec, n_reads=300, read_len=start + glen - 2)))
        orc = O.count_fastq_parallel(fq, 8, features=[(str(i), s) for i, s in enumerate(lib)], **kw)
        for rep in range(2):                                     # twice: the slabs must come back cleared
            c.reset()
            _, t = c.count_block(fq, want_timing=True)
            counts, stats = c.read_counts()
            assert t["path"] == PATH_FIXED_PART and t["general_reads"] == 0
            assert list(stats) == orc.stats() and list(counts) == orc.counts()
    for k in ("F2Q_PT_PARTS", "F2Q_PT_CHUNK", "F2Q_PT_MIN_READS"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("F2Q_NO_PT", "1")
    with P.Counter(features=lib, **kw) as c:
        _, t = c.count_block(fq, want_timing=True)
        counts2, stats2 = c.read_counts()
    monkeypatch.delenv("F2Q_NO_PT")
    assert t["path"] in (PATH_FIXED_PACKED, PATH_FIXED_LDS)
    assert list(stats2) == list(stats) and list(counts2) == list(counts)


def test_partitioned_histogram_hand_off(P, monkeypatch):
    """two of 100 k guides take nearly all of 30 M reads: the u16 counters of k_part_count pass 0x8000 hundreds of times
    and two partitions receive almost every entry (streams of very different lengths); against the packed-table kernel"""
    lib = P.binding.synth_library(0xF2A5 + 4, 100000, 20)
    spec = dict(seed=5, n_reads=30_000_000, read_len=150, p_sub=0.05, p_rand=0.01)
    res = []
    for env in ({}, {"F2Q_NO_PT": "1"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        with P.Counter(features=lib, miss=1, phred=30, length=20, start="0") as c:
            blk = c.synth_create(guides=[lib[3], lib[99999]], **spec)
            t = c.count_resident(blk)
            counts, stats = c.read_counts()
            blk.free()
        for k in env:
            monkeypatch.delenv(k)
        res.append((list(counts), list(stats), t["path"]))
    assert res[0][2] == PATH_FIXED_PART and res[1][2] == PATH_FIXED_PACKED and res[0][:2] == res[1][:2]
    assert res[0][0][3] > 10_000_000 and res[0][0][99999] > 10_000_000 and sum(res[0][0]) == res[0][1][1] + res[0][1][2]


def test_hit_buf_patch_after_wide_store(P, monkeypatch):
    """k_count_fixed4<USE_LDS=false> (a library beyond the LDS histogram on the packed tables, F2Q_NO_PT=1): a lane's four
    feature indices leave as one 16-byte store that also writes "none" for the reads whose exact probe missed; a hit the
    near search finds later is patched in by another lane of the wave.  With 70 % of the reads one substitution away
    from their guide nearly every word is patched: any reordering of the two stores would lose hits.  Against the oracle."""
    lib = P.binding.synth_library(4242, 30000, 20)
    kw = dict(miss=1, phred=30, length=20, start="0")
    spec = dict(seed=77, n_reads=400_000, read_len=60, p_sub=0.7, p_rand=0.02, p_n=0.01)
    monkeypatch.setenv("F2Q_NO_PT", "1")
    with P.Counter(features=lib, **kw) as c:
        fq = bytes(c.synth_fastq(**spec))
        orc = O.count_fastq_parallel(fq, 16, features=[(str(i), s) for i, s in enumerate(lib)], **kw)
        _, t = c.count_block(fq, want_timing=True)
        counts, stats = c.read_counts()
    monkeypatch.delenv("F2Q_NO_PT")
    assert t["path"] == PATH_FIXED_PACKED and t["general_reads"] == 0
    assert list(stats) == orc.stats() and list(counts) == orc.counts()
    assert stats[2] > 0.5 * stats[0]                              # most reads are imperfect hits


@pytest.mark.parametrize("mixed", [False, True], ids=["uniform_library", "mixed_lengths"])
@pytest.mark.parametrize("miss", [0, 1])
def test_clipped_reads_are_never_dropped_by_the_packed_kernel(P, monkeypatch, miss, mixed):
    """Regression test for the branch at f2q_count_kernels.h `else if ((int)(l & 0x7FFF) < need)` of k_count_fixed4: written
    as `L < 1 || (short && !rows_ok)` hipcc 7.2 produced code that DROPPED reads (they reached no counter at all; found
    by fuzz case 101).  A third of the reads end inside the window; with a uniform library they are decided in place
    (rows_ok), with features of several lengths they take the byte-exact routine (R_SLOW).  Every record must reach
    exactly one of the five counters, and counts and counters must equal the oracle's."""
    monkeypatch.setenv("F2Q_NO_LT", "1")                      # the packed-table kernel (k_count_fixed4), not the LDS tables
    monkeypatch.setenv("F2Q_NO_PT", "1")
    guides = P.binding.synth_library(101, 600, 20)
    lib = guides + ([g[:17] for g in guides[:80]] + [g[:12] for g in guides[80:120]] if mixed else [])
    lib = list(dict.fromkeys(lib))
    kw = dict(miss=miss, phred=30, length=20, start="4")
    with P.Counter(features=lib, **kw) as c:
        parts = []
        for i, rl in enumerate([60, 21, 60, 23, 4, 60, 24, 10, 60]):          # read lengths: 24 is the first that holds the window
            parts.append(bytes(c.synth_fastq(guides=guides, seed=50 + i, n_reads=3000, read_len=rl, start=4, p_sub=0.2, p_rand=0.05, p_n=0.02)))
        fq = b"".join(parts)
        n_records = fq.count(b"\n") // 4
        orc = O.count_fastq_parallel(fq, 8, features=[(str(i), s) for i, s in enumerate(lib)], **kw)
        _, t = c.count_block(fq, want_timing=True)
        counts, stats = c.read_counts()
    for k in ("F2Q_NO_LT", "F2Q_NO_PT"):
        monkeypatch.delenv(k)
    assert t["path"] == PATH_FIXED_PACKED
    assert stats[0] == n_records == 27000 and stats[0] == sum(stats[1:])       # nothing dropped, nothing counted twice
    assert list(stats) == orc.stats() and list(counts) == orc.counts()


def test_device_resident_text(P):
    """f2q_text_upload + f2q_count_text: FASTQ text already in device memory, framed / packed / counted there; equals
    f2q_count_block on the same bytes and the oracle, accumulates over calls, reports the bytes consumed (a partial
    record at the end stays), takes odd reads to the general path like the host entry"""
    guides = P.binding.synth_library(5, 3000, 20)
    kw = dict(miss=1, phred=30, length=20, start="3")
    with P.Counter(features=guides, **kw) as c:
        fq = sprinkle_symbols(bytes(c.synth_fastq(seed=4, n_reads=60000, read_len=70, start=3, p_sub=0.2, p_n=0.02)), 9, rate=0.004)
        fq_tail = fq + b"@partial\nACGT\n"
        orc = O.count_fastq_parallel(fq, 8, features=[(str(i), s) for i, s in enumerate(guides)], **kw)
        txt = c.text_upload(fq_tail)
        used, t = c.count_text(txt, want_timing=True)
        counts, stats = c.read_counts()
        assert used == len(fq) and t["reads"] == 60000 and t["total_ms"] >= t["kernel_ms"] > 0
        assert list(stats) == orc.stats() and list(counts) == orc.counts()
        c.count_text(txt)                                        # accumulates, and the text is still there
        counts2, stats2 = c.read_counts()
        assert list(stats2) == [2 * v for v in orc.stats()] and list(counts2) == [2 * v for v in orc.counts()]
        txt.free()
        c.reset()
        c.count_block(fq_tail)
        counts3, stats3 = c.read_counts()
        assert list(stats3) == list(stats) and list(counts3) == list(counts)
        empty = c.text_upload(b"")
        assert c.count_text(empty) == 0
        empty.free()


def _full_size_direct(P, kw, spec_extra, n_total=50_000_000, slice_reads=4_000_000, threads=16):
    """the device result of the whole synthetic stream against the oracle run over the SAME stream, slice by slice
    (FASTQ text of a slice from the device twin of the generator -> oracle on `threads` host threads -> summed)"""
    lib = P.binding.synth_library(0xF2A5 + 3, 10000, 20)
    feats = [(str(i), s) for i, s in enumerate(lib)]
    tot_counts, tot_stats = None, None
    with P.Counter(features=lib, **kw) as c:
        for lo in range(0, n_total, slice_reads):
            n = min(slice_reads, n_total - lo)
            fq = bytes(c.synth_fastq(lo=lo, hi=lo + n, seed=0xBEEF, n_reads=n_total, read_len=150, **spec_extra))   # reads lo .. lo + n - 1 of the stream
            orc = O.count_fastq_parallel(fq, threads, features=feats, **kw)
            del fq
            tot_counts = orc.counts() if tot_counts is None else [a + b for a, b in zip(tot_counts, orc.counts())]
            tot_stats = orc.stats() if tot_stats is None else [a + b for a, b in zip(tot_stats, orc.stats())]
        blk = c.synth_create(seed=0xBEEF, n_reads=n_total, read_len=150, **spec_extra)
        t = c.count_resident(blk)
        counts, stats = c.read_counts()
        blk.free()
    assert t["general_reads"] == 0
    assert list(stats) == tot_stats and tot_stats[0] == n_total
    assert list(counts) == tot_counts


@pytest.mark.slow
@pytest.mark.timeout(1200)
def test_full_size_config3_direct_oracle(P):
    """BASELINE config 3 at full size (50 M x 150 bp vs 10 k x 20 bp guides, --m 1 --ph 30): every per-feature count and
    the five counters of the library-in-LDS kernel against the oracle over the whole stream (about a minute of host work)"""
    _full_size_direct(P, dict(miss=1, phred=30, length=20, start="0"), {})


@pytest.mark.slow
@pytest.mark.timeout(1200)
def test_full_size_config5a_direct_oracle(P):
    """BASELINE config 5a at full size (anchored Counter mode, --us/--ds 12-mers, --msu 1 --msd 1 --m 1) against the
    oracle over the whole stream"""
    up, down = "GTTTAAGAGCTA", "CGTTACCAGGTT"
    _full_size_direct(P, dict(miss=1, phred=30, upstream=up, downstream=down, miss_search_up=1, miss_search_down=1),
                      dict(cassette=True, up=up, down=down, max_offset=100))


@pytest.mark.parametrize("mode", ["C", "EC"])
@pytest.mark.parametrize("anchors,ms,rl", [("both", 1, 251), ("up", 0, 301), ("down", 2, 200), ("pairs", 1, 320)])
def test_packed_anchor_long_reads_gpu(P, mode, anchors, ms, rl):
    """reads of 161 .. 320 bases (MiSeq 2 x 250 / 2 x 300 amplicons) in anchored runs stay on the bit-plane kernels (ten
    32-base plane words per read): general_reads counts only the reads beyond 320 bases; against the oracle"""
    up, down = "GTTTAAGAGCTA", "CGTTACCAGGTT"
    guides = P.binding.synth_library(77 + rl, 150, 20)
    kw = dict(mode=mode, miss=1, length=20, miss_search_up=ms, miss_search_down=ms)
    feats = guides
    if anchors == "pairs":
        kw["upstream"] = up + "," + up; kw["downstream"] = down + "," + down
        feats = [g + ":" + g for g in guides[:100]] + guides[100:]
    else:
        if anchors in ("both", "up"):
            kw["upstream"] = up
        if anchors in ("both", "down"):
            kw["downstream"] = down
    with P.Counter(features=feats if mode == "C" else None, **kw) as c:
        gen = dict(guides=guides, cassette=True, up=up, down=down)
        fq = bytes(c.synth_fastq(seed=rl + ms, n_reads=30000, read_len=rl, max_offset=rl - 50, p_sub=0.2, p_lowq=0.1, p_n=0.0, **gen))
        fq += bytes(c.synth_fastq(seed=3, n_reads=2000, read_len=90, max_offset=40, **gen))
        fq += bytes(c.synth_fastq(seed=4, n_reads=300, read_len=400, max_offset=330, **gen))
        orc = O.Oracle(features=[(str(i), s) for i, s in enumerate(feats)] if mode == "C" else None, **kw)
        orc.count_fastq(fq)
        _, t = c.count_block(fq, want_timing=True)
        counts, stats = c.read_counts()
        # (with ':' features in the library the few reads that carry an N take the byte-exact routine as well)
        assert t["general_reads"] + t["fast_reads"] == 32300 and (t["general_reads"] == 300 if anchors != "pairs" else 300 <= t["general_reads"] < 340)
        assert list(stats) == orc.stats()
        if mode == "C":
            assert list(counts) == orc.counts()
        else:
            assert [(k, n) for k, n, _ in c.ec_results()] == list(zip(orc.keys(), orc.counts()))
        # and device-generated long reads (the bench's generator at 250 bp)
        c.reset()
        blk = c.synth_create(seed=11, n_reads=50000, read_len=250, max_offset=190, p_n=0.0, **gen)
        tt = c.count_resident(blk)
        _, st2 = c.read_counts()
        blk.free()
        assert tt["general_reads"] == 0 and st2[0] == 50000


@pytest.mark.parametrize("mode", ["C", "EC"])
@pytest.mark.parametrize("anchors,ms", [("both", 1), ("up", 0), ("down", 2), ("pairs", 1)])
def test_packed_anchor_mixed_case_reads_gpu(P, mode, anchors, ms):
    """lower-case bases in anchored runs stay on the bit-plane kernels (marks for the case-sensitive anchor search, stored
    codes for the upper-cased window): half of 60 k reads carry lower-case bases, some entirely; general_reads counts only
    the reads that hold lower case AND a non-ACGT symbol; against the oracle"""
    from test_lane_logic_cpu import lower_case_some
    up, down = "GTTTAAGAGCTA", "CGTTACCAGGTT"
    guides = P.binding.synth_library(4711, 150, 20)
    kw = dict(mode=mode, miss=1, length=20, miss_search_up=ms, miss_search_down=ms)
    feats = guides
    if anchors == "pairs":
        kw["upstream"] = up + "," + up; kw["downstream"] = down + "," + down
        feats = [g + ":" + g for g in guides[:100]] + guides[100:]
    else:
        if anchors in ("both", "up"):
            kw["upstream"] = up
        if anchors in ("both", "down"):
            kw["downstream"] = down
    with P.Counter(features=feats if mode == "C" else None, **kw) as c:
        fq = bytes(c.synth_fastq(guides=guides, seed=ms + 40, n_reads=60000, read_len=150, cassette=True, up=up, down=down, max_offset=100,
                                 p_sub=0.2, p_lowq=0.1, p_n=0.01))
        fq = lower_case_some(fq, 17)
        orc = O.Oracle(features=[(str(i), s) for i, s in enumerate(feats)] if mode == "C" else None, **kw)
        orc.count_fastq(fq)
        _, t = c.count_block(fq, want_timing=True)
        counts, stats = c.read_counts()
        assert t["fast_reads"] + t["general_reads"] == 60000 and t["general_reads"] < (900 if anchors != "pairs" else 1500)
        assert list(stats) == orc.stats()
        if mode == "C":
            assert list(counts) == orc.counts()
        else:
            assert [(k, n) for k, n, _ in c.ec_results()] == list(zip(orc.keys(), orc.counts()))
