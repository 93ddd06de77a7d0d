"""GPU: the drop-in surface -- reads_counter() on files, and the `2fast2q -c` command line end to end."""
import csv
import gzip
import importlib
import os

import pytest

import synth
from conftest import ROOT, bgzf_bytes, pkg, sprinkle_symbols
from oracle import oracle as O

pytestmark = pytest.mark.gpu
fast2q = importlib.import_module("2fast2q_amd.fast2q")


def default_param(**kw):
    p = {"Running Mode": "C", "miss": 1, "phred": 30, "length": 20, "start": "0", "upstream": None, "downstream": None,
         "miss_search_up": 0, "miss_search_down": 0, "qual_up": 30, "qual_down": 30, "Progress bar": False,
         "big_file_split": False}
    p.update(kw)
    return p


@pytest.mark.parametrize("gz", [False, True])
def test_reads_counter_seam(tmp_path, gz):
    guides = synth.make_library(300, 20, 41)
    fq = synth.make_fastq(synth.Spec(seed=21, n_reads=5000, read_len=80), guides)
    path = tmp_path / ("a.fastq.gz" if gz else "a.fastq")
    (gzip.open(path, "wb") if gz else open(path, "wb")).write(fq)
    features = {g: fast2q.Features(f"n{i}", 0) for i, g in enumerate(guides)}
    memo = {"failed_reads": set(), "passed_reads": {}}
    out = fast2q.reads_counter(0, str(path), features, default_param(), memo)
    feats, memo2, stats = out
    orc = O.Oracle(features=[(f"n{i}", g) for i, g in enumerate(guides)], miss=1)
    orc.count_fastq(fq)
    assert feats is features and memo2 is memo
    assert [f.counts for f in feats.values()] == orc.counts()
    assert stats == orc.stats_dict()


def test_reads_counter_extract_count(tmp_path):
    guides = synth.make_library(100, 20, 42)
    up, down = "GTTTAAGAGCTA", "CGTTACCAGGTT"
    fq = synth.make_fastq(synth.Spec(seed=22, n_reads=3000, cassette=True, up=up, down=down), guides)
    path = tmp_path / "b.fastq"
    path.write_bytes(fq)
    kw = dict(upstream=up, downstream=down, miss_search_up=1, miss_search_down=1)
    feats, _, stats = fast2q.reads_counter(0, str(path), {}, default_param(**{"Running Mode": "EC"}, **kw), {})
    orc = O.Oracle(mode="EC", **kw)
    orc.count_fastq(fq)
    assert [(k, f.name, f.counts) for k, f in feats.items()] == [(k, k, n) for k, n in zip(orc.keys(), orc.counts())]
    assert stats == orc.stats_dict()


def test_truncated_gzip_partial_counts(tmp_path, capsys):
    guides = synth.make_library(50, 20, 43)
    fq = synth.make_fastq(synth.Spec(seed=23, n_reads=20000, read_len=60), guides)
    raw = gzip.compress(fq)
    path = tmp_path / "t.fastq.gz"
    path.write_bytes(raw[: len(raw) // 2])
    features = {g: fast2q.Features(str(i), 0) for i, g in enumerate(guides)}
    feats, _, stats = fast2q.reads_counter(0, str(path), features, default_param(), {})
    assert 0 < stats["reads"] < 20000                                # partial processing kept (fast2q.py:405-407)
    assert "incomplete or corrupted gzip" in capsys.readouterr().out


def test_cut_off_gzip_vs_reference(tmp_path, capsys):
    """archives cut at several positions: counts and stats must be what the reference's reads_counter returned for the
    same bytes (tests/golden/truncated_gzip_cases.json) -- the cut-off last line is not a line"""
    import base64
    import json
    with open(os.path.join(ROOT, "tests", "golden", "truncated_gzip_cases.json")) as f:
        cases = json.load(f)["cases"]
    for c in cases:
        path = tmp_path / f"cut_{c['level']}_{c['frac']}.fastq.gz"
        path.write_bytes(base64.b64decode(c["gz_b64"]))
        features = {seq: fast2q.Features(name, 0) for name, seq in c["features"]}
        feats, _, stats = fast2q.reads_counter(0, str(path), features, default_param(), {})
        assert [f.counts for f in feats.values()] == c["expected"]["counts"], (c["level"], c["frac"])
        assert [stats[k] for k in fast2q.binding.STAT_NAMES] == c["expected"]["stats"]
        assert "incomplete or corrupted gzip" in capsys.readouterr().out


def test_empty_and_tiny_files(tmp_path):
    """files with nothing / less than one record / one record without a final newline: the reader thread, the carry
    and the end-of-file piece of f2q_count_file"""
    guides = synth.make_library(20, 20, 3)
    one = b"@r\n" + guides[3].encode() + b"\n+\n" + b"I" * 20
    cases = {"empty.fastq": (b"", 0), "three_lines.fastq": (b"@r\nACGT\n+\n", 0), "one.fastq": (one, 1),
             "one_and_a_bit.fastq": (one + b"\n@x\nAC", 1)}
    with pkg().Counter(features=guides, miss=0) as c:
        for name, (data, n) in cases.items():
            for kind in ("plain", "gzip", "bgzf"):
                path = tmp_path / (name if kind == "plain" else name + ".gz")
                path.write_bytes({"plain": data, "gzip": gzip.compress(data), "bgzf": bgzf_bytes(data)}[kind])
                c.reset()
                t, trunc = c.count_file(str(path))
                counts, stats = c.read_counts()
                assert not trunc and stats[0] == n and sum(counts) == n and (n == 0 or counts[3] == 1), (name, kind)
    with pytest.raises(pkg().binding.F2QError):
        with pkg().Counter(features=guides) as c:
            c.count_file(str(tmp_path / "does_not_exist.fastq"))


@pytest.mark.parametrize("kind", ["plain", "gzip", "gzip_par", "bgzf"])
@pytest.mark.parametrize("world", [2, 3])
def test_count_file_shard_sums_to_the_whole(tmp_path, monkeypatch, kind, world):
    """f2q_count_file_shard: `world` contexts (one per rank) on the same file; pieces dealt round robin, framing global.
    Counter mode: the rank vectors add up to the oracle's; Extract+Count: tables merged by key (sum, min first-read)
    give the oracle's dict in its order.  Small pieces so that records straddle piece and rank boundaries."""
    sharding = importlib.import_module("2fast2q_amd.sharding")
    monkeypatch.setenv("F2Q_FILE_CHUNK", "50000")
    guides = synth.make_library(150, 20, 5)
    fq = sprinkle_symbols(synth.make_fastq(synth.Spec(seed=31, n_reads=12000, read_len=101), guides), 3, rate=0.004)
    fq = fq.replace(b"\n", b"\r\n", 500) + b"@tail\nACGT"              # CRLF lines and a partial record at the end
    if kind == "gzip_par":                                            # the one deflate stream decoded by the worker pool, in 64 KiB chunks
        monkeypatch.setenv("F2Q_GZ_PAR_MIN_KB", "16"); monkeypatch.setenv("F2Q_GZ_CHUNK_KB", "64")
    path = tmp_path / ("s.fastq" if kind == "plain" else "s.fastq.gz")
    path.write_bytes({"plain": fq, "gzip": gzip.compress(fq, 1), "gzip_par": gzip.compress(fq, 6), "bgzf": bgzf_bytes(fq, block=30000)}[kind])
    for kw in (dict(miss=1), dict(mode="EC", start="5", length=18), dict(mode="EC", upstream="ACGT", length=9)):
        feats = guides if "mode" not in kw else None
        orc = O.Oracle(features=[(str(i), g) for i, g in enumerate(guides)] if feats else None, **kw)
        orc.count_fastq(fq)
        tot_counts, tot_stats, tables, reads = None, [0] * 5, [], 0
        for rank in range(world):
            with pkg().Counter(features=feats, **kw) as c:
                t, trunc = c.count_file_shard(str(path), rank, world)
                assert not trunc
                counts, stats = c.read_counts()
                reads += t["reads"]
                tot_stats = [a + int(b) for a, b in zip(tot_stats, stats)]
                if feats:
                    tot_counts = list(counts) if tot_counts is None else [a + b for a, b in zip(tot_counts, counts)]
                else:
                    tables.append(c.ec_results())
        assert tot_stats == orc.stats() and reads == orc.stats()[0]
        if feats:
            assert tot_counts == orc.counts()
        else:
            assert [(k, n) for k, n, _ in sharding.merge_ec_tables(tables)] == list(zip(orc.keys(), orc.counts()))


@pytest.mark.gpu
@pytest.mark.parametrize("world,piece,block", [(2, 1 << 16, 20000), (3, 200000, 0xFF00), (4, 4096, 3000), (1, 1 << 20, 0xFF00)])
def test_count_pieces_of_a_bgzf_file(tmp_path, world, piece, block):
    """the same protocol on a BGZF file: a piece is a run of whole members (their text sizes come from the trailers), each
    rank inflates only its own runs -- once for the census, once to count -- plus the members behind a run that finish its
    last record.  Empty members inside the file, a partial last record, ordinary gzip refused."""
    sharding = importlib.import_module("2fast2q_amd.sharding")
    guides = synth.make_library(150, 20, 5)
    fq = sprinkle_symbols(synth.make_fastq(synth.Spec(seed=34, n_reads=7000, read_len=101), guides), 3, rate=0.004)
    fq = fq.replace(b"\n", b"\r\n", 300) + b"@tail\nACGT"
    cut = len(fq) // 3
    raw = bgzf_bytes(fq[:cut], block=block, eof_marker=True) + bgzf_bytes(fq[cut:], block=block)     # an empty member in the middle
    path = tmp_path / "s.fastq.gz"
    path.write_bytes(raw)
    for kw in (dict(miss=1), dict(mode="EC", upstream="ACGT", length=9)):
        feats = guides if "mode" not in kw else None
        orc = O.Oracle(features=[(str(i), g) for i, g in enumerate(guides)] if feats else None, **kw)
        orc.count_fastq(fq)
        ctxs = [pkg().Counter(features=feats, **kw) for _ in range(world)]
        n_pieces, ok = ctxs[0].file_pieces(str(path), piece)
        assert ok and n_pieces >= max(1, len(fq) // max(piece, block + 1) // 2)
        census = sum(c.census_pieces(str(path), r, world, piece, n_pieces) for r, c in enumerate(ctxs))
        assert int(census[0::2].sum()) == fq.count(b"\n")
        tot_counts, tot_stats, tables, reads = None, [0] * 5, [], 0
        for r, c in enumerate(ctxs):
            t = c.count_pieces(str(path), r, world, piece, census)
            counts, stats = c.read_counts()
            reads += t["reads"]
            tot_stats = [a + int(b) for a, b in zip(tot_stats, stats)]
            if feats:
                tot_counts = list(counts) if tot_counts is None else [a + b for a, b in zip(tot_counts, counts)]
            else:
                tables.append(c.ec_results())
            c.close()
        assert tot_stats == orc.stats() and reads == orc.stats()[0]
        if feats:
            assert tot_counts == orc.counts()
        else:
            assert [(k, n) for k, n, _ in sharding.merge_ec_tables(tables)] == list(zip(orc.keys(), orc.counts()))
    # BGZF that turns into ordinary gzip half way: not shardable (front-to-back reading only)
    mixed = tmp_path / "m.fastq.gz"
    mixed.write_bytes(bgzf_bytes(fq[:cut], block=block, eof_marker=False) + gzip.compress(fq[cut:], 1))
    with pkg().Counter(features=guides) as c:
        assert c.file_pieces(str(mixed), piece) == (0, False)


@pytest.mark.parametrize("world", [1, 2, 3, 5])
@pytest.mark.parametrize("piece", [4096, 50000, 1 << 20])
def test_count_pieces_without_foreign_bytes(tmp_path, world, piece):
    """f2q_file_pieces / f2q_census_pieces / f2q_count_pieces: every rank counts the newlines of its own pieces, the
    census vectors are summed (here: in this process; in a run: one all-reduce), and each rank frames and counts the
    records that start in its pieces.  The rank results add up to the oracle's on a file with CRLF lines, lines that
    straddle pieces, a partial last record -- for Counter mode and for Extract+Count (keys merged by first read)."""
    sharding = importlib.import_module("2fast2q_amd.sharding")
    guides = synth.make_library(150, 20, 5)
    fq = sprinkle_symbols(synth.make_fastq(synth.Spec(seed=33, n_reads=9000, read_len=101), guides), 3, rate=0.004)
    fq = fq.replace(b"\n", b"\r\n", 500) + b"@tail\nACGT"
    path = tmp_path / "s.fastq"
    path.write_bytes(fq)
    for kw in (dict(miss=1), dict(mode="EC", upstream="ACGT", length=9)):
        feats = guides if "mode" not in kw else None
        orc = O.Oracle(features=[(str(i), g) for i, g in enumerate(guides)] if feats else None, **kw)
        orc.count_fastq(fq)
        ctxs = [pkg().Counter(features=feats, **kw) for _ in range(world)]
        n_pieces, ok = ctxs[0].file_pieces(str(path), piece)
        assert ok and n_pieces == -(-len(fq) // piece)
        census = sum(c.census_pieces(str(path), r, world, piece, n_pieces) for r, c in enumerate(ctxs))
        assert int(census[0::2].sum()) == fq.count(b"\n")
        tot_counts, tot_stats, tables, reads = None, [0] * 5, [], 0
        for r, c in enumerate(ctxs):
            t = c.count_pieces(str(path), r, world, piece, census)
            counts, stats = c.read_counts()
            reads += t["reads"]
            tot_stats = [a + int(b) for a, b in zip(tot_stats, stats)]
            if feats:
                tot_counts = list(counts) if tot_counts is None else [a + b for a, b in zip(tot_counts, counts)]
            else:
                tables.append(c.ec_results())
            c.close()
        assert tot_stats == orc.stats() and reads == orc.stats()[0]
        if feats:
            assert tot_counts == orc.counts()
        else:
            assert [(k, n) for k, n, _ in sharding.merge_ec_tables(tables)] == list(zip(orc.keys(), orc.counts()))
    gz = tmp_path / "s.fastq.gz"
    gz.write_bytes(gzip.compress(fq, 1))
    with pkg().Counter(features=guides) as c:
        assert c.file_pieces(str(gz), piece) == (0, False)            # compressed input: the streaming shard path
    # a line longer than the look-ahead behind a piece: refused (the caller falls back to f2q_count_file_shard)
    long_line = tmp_path / "long.fastq"
    long_line.write_bytes(b"@r\n" + b"A" * (3 << 20) + b"\n+\n" + b"I" * (3 << 20) + b"\n" + fq)
    with pkg().Counter(features=guides, miss=1) as c:
        n_p, ok = c.file_pieces(str(long_line), 1 << 20)
        cen = c.census_pieces(str(long_line), 0, 1, 1 << 20, n_p)
        with pytest.raises(pkg().F2QError):
            c.count_pieces(str(long_line), 0, 1, 1 << 20, cen)


def _output_cases():
    from test_abi_and_host import _golden_compiling_cases
    return _golden_compiling_cases()[0]


@pytest.mark.parametrize("case", _output_cases(), ids=[c["name"] for c in _output_cases()])
def test_output_contract_through_the_device(case, tmp_path, monkeypatch):
    """the multi-sample runs of tests/golden/compiling_cases.json -- made by running the reference's aligner + compiling --
    through aligner() -> reads_counter() -> libf2q_hip.so -> compiling(): compiled.csv and compiled_stats.csv must
    be the reference's bytes (only the clock is scripted, as it was for the reference)"""
    from test_abi_and_host import _Clock, _golden_compiling_cases, golden_param
    exp = case["expected"]
    indir = tmp_path / "in"
    out = tmp_path / "out"
    indir.mkdir(); out.mkdir()
    for fname, text in case["files"]:
        data = text.encode("latin-1")
        (indir / fname).write_bytes(gzip.compress(data) if fname.endswith(".gz") else data)
    param = golden_param(case, out)
    lib = {}
    if case["features"] is not None:
        for name, seq in case["features"]:
            lib.setdefault(seq.upper().replace(" ", ""), name)
    monkeypatch.setattr(fast2q.time, "perf_counter", _Clock(_golden_compiling_cases()[1]))
    for i, (fname, _) in enumerate(case["files"]):
        per_sample = {s: fast2q.Features(n, 0) for s, n in lib.items()}
        fast2q.aligner(i, str(indir / fname), per_sample, param, {})
    fast2q.compiling(param)
    assert (out / "compiled.csv").read_bytes().decode("latin-1") == exp["compiled.csv"]
    assert (out / "compiled_stats.csv").read_bytes().decode("latin-1") == exp["compiled_stats.csv"]
    assert sorted(os.listdir(out)) == exp["files_left"]


def test_cli_test_mode_end_to_end(tmp_path, monkeypatch):
    """`2fast2q -c -t` (reference tests/test_cli.py): exit 0, one output folder, exactly 6 files; and -- what
    upstream leaves commented out -- compiled.csv equals an independent count of the same input."""
    monkeypatch.chdir(tmp_path)
    fast2q.main(["-c", "-t", "--pb"])
    subdirs = [d for d in tmp_path.iterdir() if d.is_dir()]
    assert len(subdirs) == 1
    files = sorted(p.name for p in subdirs[0].iterdir())
    assert len(files) == 6 and "compiled.csv" in files and "compiled_stats.csv" in files
    feats = fast2q.features_loader(os.path.join(ROOT, "2fast2q_amd", "data", "D39V_guides.csv"))
    fq = gzip.open(fast2q.ensure_example_fastq()).read()
    orc = O.Oracle(features=[(f.name, s) for s, f in feats.items()], miss=1)
    orc.count_fastq(fq)
    want = sorted(zip(orc.names, orc.counts()))
    rows = list(csv.reader(open(subdirs[0] / "compiled.csv", newline="")))
    assert rows[0] == ["#Feature", "example"]
    assert [(r[0], int(r[1])) for r in rows[1:]] == want
    stats = list(csv.reader(open(subdirs[0] / "compiled_stats.csv")))[-1]
    assert [int(x) for x in (stats[3], stats[5], stats[6], stats[7], stats[8])] == orc.stats()
    assert (subdirs[0] / "compiled.csv").read_bytes().count(b"\r\n") == len(rows)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["C", "EC"])
def test_contexts_are_kept_between_samples(tmp_path, monkeypatch, mode):
    """reads_counter keeps its context from one sample to the next (library index, staging buffers): every sample must
    still start from zeroed counters and empty Extract+Count tables; F2Q_NO_CTX_CACHE=1 is the fresh-context path"""
    guides = synth.make_library(80, 20, 45)
    param = dict(fast2q.initializer(fast2q.input_parser(["-c", "--s", str(tmp_path), "--g", "x", "--o", str(tmp_path), "--mo", mode, "--m", "1"])))
    files, want = [], []
    for k, n in enumerate((1200, 300, 2500)):
        fq = synth.make_fastq(synth.Spec(seed=60 + k, n_reads=n, read_len=60), guides)
        (tmp_path / f"s{k}.fastq").write_bytes(fq)
        orc = O.Oracle(features=[(g, g) for g in guides] if mode == "C" else None, mode=mode, miss=1)
        orc.count_fastq(fq)
        files.append(str(tmp_path / f"s{k}.fastq")); want.append((orc.counts(), orc.stats(), orc.keys() if mode == "EC" else None))
    for env in ("", "1"):
        monkeypatch.setenv("F2Q_NO_CTX_CACHE", env)
        for f, (counts, stats, keys) in zip(files + files[:1], want + want[:1]):
            feats = {g: fast2q.Features(g, 0) for g in guides} if mode == "C" else {}
            feats, _, local = fast2q.reads_counter(0, f, feats, param, {})
            assert [local[k] for k in fast2q.binding.STAT_NAMES] == stats
            if mode == "C":
                assert [feats[g].counts for g in guides] == counts
            else:
                assert list(feats) == keys and [v.counts for v in feats.values()] == counts
        fast2q.close_contexts()


def test_cli_directory_of_samples(tmp_path):
    guides = synth.make_library(60, 20, 44)
    (tmp_path / "in").mkdir()
    csvp = tmp_path / "lib.csv"
    csvp.write_text("".join(f"{100 - i},{g}\n" for i, g in enumerate(guides)))       # numeric names: int sort (:791)
    exp = {}
    for k, n in (("s1", 1500), ("s2", 700)):
        fq = synth.make_fastq(synth.Spec(seed=30 + n, n_reads=n, read_len=50), guides)
        (tmp_path / "in" / f"{k}.fastq").write_bytes(fq)
        orc = O.Oracle(features=[(str(100 - i), g) for i, g in enumerate(guides)], miss=0)
        orc.count_fastq(fq)
        exp[k] = dict(zip(orc.names, orc.counts()))
    fast2q.main(["-c", "--s", str(tmp_path / "in"), "--g", str(csvp), "--o", str(tmp_path), "--m", "0", "--pb", "--fn", "res"])
    out = [d for d in tmp_path.iterdir() if d.is_dir() and d.name.startswith("2FAST2Q_output_")][0]
    rows = list(csv.reader(open(out / "res.csv", newline="")))
    assert rows[0] == ["#Feature", "s1", "s2"]
    assert [r[0] for r in rows[1:]] == [str(v) for v in sorted(100 - i for i in range(60))]
    for r in rows[1:]:
        assert int(r[1]) == exp["s1"][r[0]] and int(r[2]) == exp["s2"][r[0]]


@pytest.mark.parametrize("in_flight", ["", "4", "1"])
def test_cli_samples_in_parallel_threads(tmp_path, monkeypatch, in_flight):
    """--cp N: samples in flight (threads over independent contexts, two by default, F2Q_SAMPLES_IN_FLIGHT overrides);
    results identical to one at a time"""
    if in_flight:
        monkeypatch.setenv("F2Q_SAMPLES_IN_FLIGHT", in_flight)
    guides = synth.make_library(40, 20, 45)
    (tmp_path / "in").mkdir()
    csvp = tmp_path / "lib.csv"
    csvp.write_text("".join(f"g{i},{g}\n" for i, g in enumerate(guides)))
    exp = {}
    for k in range(6):
        fq = synth.make_fastq(synth.Spec(seed=60 + k, n_reads=800 + 100 * k, read_len=45), guides)
        with gzip.open(tmp_path / "in" / f"s{k}.fastq.gz", "wb") as f:
            f.write(fq)
        orc = O.Oracle(features=[(f"g{i}", g) for i, g in enumerate(guides)], miss=1)
        orc.count_fastq(fq)
        exp[f"s{k}"] = dict(zip(orc.names, orc.counts()))
    fast2q.main(["-c", "--s", str(tmp_path / "in"), "--g", str(csvp), "--o", str(tmp_path), "--pb", "--cp", "4"])
    out = [d for d in tmp_path.iterdir() if d.is_dir() and d.name.startswith("2FAST2Q_output_")][0]
    rows = list(csv.reader(open(out / "compiled.csv", newline="")))
    assert rows[0] == ["#Feature"] + [f"s{k}" for k in range(6)]
    for r in rows[1:]:
        for k in range(6):
            assert int(r[1 + k]) == exp[f"s{k}"][r[0]]


def test_integration_md_stub_runs(tmp_path, monkeypatch):
    """the ctypes stub printed in INTEGRATION.md (what a reference maintainer would paste at the reads_counter seam)
    is executed as written against libf2q_hip.so"""
    import re
    from dataclasses import dataclass
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    code = re.search(r"```python\n(.*?)```", text, re.S).group(1)
    monkeypatch.setenv("F2Q_LIB", pkg().LIB_PATH)

    @dataclass
    class Features:
        name: str
        counts: int
    ns = {"Features": Features}
    exec(compile(code, "INTEGRATION.md", "exec"), ns)
    guides = synth.make_library(80, 20, 46)
    fq = synth.make_fastq(synth.Spec(seed=70, n_reads=4000, read_len=64), guides)
    path = tmp_path / "x.fastq.gz"
    with gzip.open(path, "wb") as f:
        f.write(fq)
    param = default_param()
    feats = {g: Features(f"n{i}", 0) for i, g in enumerate(guides)}
    out = ns["reads_counter"](0, str(path), feats, param, {"failed_reads": set(), "passed_reads": {}})
    orc = O.Oracle(features=[(f"n{i}", g) for i, g in enumerate(guides)], miss=1)
    orc.count_fastq(fq)
    assert [f.counts for f in out[0].values()] == orc.counts() and out[2] == orc.stats_dict()
    ec = ns["reads_counter"](0, str(path), {}, default_param(**{"Running Mode": "EC"}), {})
    orc2 = O.Oracle(mode="EC")
    orc2.count_fastq(fq)
    assert [(k, f.counts) for k, f in ec[0].items()] == list(zip(orc2.keys(), orc2.counts()))


def test_console_script_like_the_reference_test(tmp_path):
    """reference tests/test_cli.py:5-25 verbatim in spirit: run `2fast2q -c -t` as a subprocess in an empty
    directory; exit code 0, exactly one output folder, exactly 6 files, compiled.csv among them"""
    import subprocess
    import sys
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bin", "2fast2q"), "-c", "-t"], cwd=tmp_path,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert res.returncode == 0, res.stderr
    subdirs = [d for d in tmp_path.iterdir() if d.is_dir()]
    assert len(subdirs) == 1
    files = list(subdirs[0].iterdir())
    assert len(files) == 6 and (subdirs[0] / "compiled.csv").exists()


@pytest.mark.parametrize("gz", ["plain", "gzip", "gzip_par", "bgzf"])
@pytest.mark.parametrize("chunk", ["4096", "65536", "1000003"])
@pytest.mark.parametrize("staging", ["as_it_comes", "every_piece", "never"])
def test_file_streaming_across_chunk_boundaries(tmp_path, monkeypatch, gz, chunk, staging):
    """f2q_count_file streams a file in blocks: records and lines that straddle a block boundary, a block that
    holds no complete record, CRLF, and an unterminated last line must not change the counts"""
    monkeypatch.setenv("F2Q_FILE_CHUNK", chunk)
    # the text of the next piece may travel to the device ahead of time (f2q_count_file): with every piece forced down
    # that path the carried tail lands at every alignment in front of it; "never" is the plain path
    if staging == "every_piece": monkeypatch.setenv("F2Q_FORCE_STAGING", "1")
    if staging == "never": monkeypatch.setenv("F2Q_NO_STAGING", "1")
    guides = synth.make_library(120, 20, 47)
    fq = synth.make_fastq(synth.Spec(seed=80, n_reads=9000, read_len=151), guides)
    fq = fq.replace(b"\n", b"\r\n", 3000)                      # some CRLF line ends
    fq += b"@long\n" + b"ACGT" * 3000 + b"\n+\n" + b"I" * 12000 + b"\n"      # a 12 kb read: longer than the smallest chunk
    fq += synth.make_fastq(synth.Spec(seed=82, n_reads=300, read_len=75), guides)
    fq += b"@huge\n" + guides[7].encode() + b"ACGT" * 20000 + b"\n+\n" + b"I" * 80020 + b"\n"   # 80 kb: longer than the carry head room
    fq += synth.make_fastq(synth.Spec(seed=81, n_reads=500, read_len=40), guides)[:-1]   # no final newline
    if gz == "gzip_par":                                              # the worker pool on the one deflate stream (f2q_pargz.h)
        monkeypatch.setenv("F2Q_GZ_PAR_MIN_KB", "16"); monkeypatch.setenv("F2Q_GZ_CHUNK_KB", "64")
    path = tmp_path / ("f.fastq" if gz == "plain" else "f.fastq.gz")
    path.write_bytes({"plain": fq, "gzip": gzip.compress(fq, 1), "gzip_par": gzip.compress(fq, 1), "bgzf": bgzf_bytes(fq, block=20000)}[gz])
    for kw in (dict(miss=1), dict(mode="EC", upstream="ACGT", length=9)):
        orc = O.Oracle(features=[(str(i), g) for i, g in enumerate(guides)] if "mode" not in kw else None, **kw)
        orc.count_fastq(fq)
        with pkg().Counter(features=guides if "mode" not in kw else None, **kw) as c:
            t, trunc = c.count_file(str(path))
            counts, stats = c.read_counts()
            assert not trunc and list(stats) == orc.stats() and t["reads"] == orc.stats()[0]
            if "mode" not in kw:
                assert list(counts) == orc.counts()
            else:
                assert [(k, n) for k, n, _ in c.ec_results()] == list(zip(orc.keys(), orc.counts()))
