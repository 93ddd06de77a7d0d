#!/opt/conda/bin/python3.9
"""tests/golden/make_golden.py -- regenerates tests/golden/*.json by RUNNING THE REFERENCE.

Runs only in the build container (it needs /root/reference and
/opt/conda/bin/python3.9, which has colorama/tqdm/psutil/matplotlib); the GPU box
and the test-suite only ever read the JSON it writes.

How the reference is run: ``fast2q.fast2q`` is imported from /root/reference and
its own ``initializer`` (:1082), ``features_loader`` (:125) and ``reads_counter``
(:514 -- the drop-in seam) are called on small FASTQ files written to a temp dir.
Numba is not importable offline (0.54.1 vs numpy 1.26), so the
three ``@njit`` helpers (:601,:628,:660) run as the plain Python they are: a
temp-dir ``numba`` module whose ``njit`` returns the function unchanged is put on
sys.path for the import.  No reference source is copied; the fixtures hold inputs
and the outputs the reference produced.

Usage:  /opt/conda/bin/python3.9 tests/golden/make_golden.py
"""
import contextlib
import io
import json
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))          # tests/  (for synth.py)
import synth  # noqa: E402

REF = "/root/reference"


def import_reference():
    shim = tempfile.mkdtemp(prefix="f2q_shim_")
    os.makedirs(os.path.join(shim, "numba"))
    with open(os.path.join(shim, "numba", "__init__.py"), "w") as f:
        f.write("def njit(f=None, *a, **k):\n    return f if callable(f) else (lambda g: g)\n"
                "from . import types, typed\n")
    with open(os.path.join(shim, "numba", "types.py"), "w") as f:
        f.write("unicode_type = str\nclass _T:\n    def __getitem__(self, k):\n        return None\nint8 = _T()\n")
    with open(os.path.join(shim, "numba", "typed.py"), "w") as f:
        f.write("class Dict(dict):\n    @classmethod\n    def empty(cls, key_type=None, value_type=None):\n        return cls()\n")
    sys.path.insert(0, shim)
    sys.path.insert(0, REF)
    os.environ.setdefault("MPLBACKEND", "Agg")
    sys.dont_write_bytecode = True
    from fast2q import fast2q as ref
    ref.version = "2.8.1"
    return ref


def cli_defaults():
    # the dict input_parser() builds (fast2q.py:1226-1309) with its defaults
    return {"cmd": True, "big_file_split": False, "test_mode": False, "out_file_name": "compiled", "length": 20,
            "Progress bar": False, "start": "0", "phred": 30, "miss": 1, "upstream": None, "downstream": None,
            "miss_search_up": 0, "miss_search_down": 0, "qual_up": 30, "qual_down": 30, "Running Mode": "C",
            "delete": True, "cpu": 1}


def run_reference(ref, params, features, fastq_bytes, fname="s.fastq"):
    """features: list of [name, seq-as-written-in-csv] or None (EC). Returns the expected block."""
    tmp = tempfile.mkdtemp(prefix="f2q_gold_")
    p = cli_defaults()
    p.update(params)
    p["out"] = tmp
    p["seq_files"] = tmp
    with contextlib.redirect_stdout(io.StringIO()):
        param = ref.initializer(dict(p))
    feats = {}
    if param["Running Mode"] == "C":
        csvp = os.path.join(tmp, "features.csv")
        with open(csvp, "w") as f:
            for name, seq in features:
                f.write(f"{name},{seq}\n")
        with contextlib.redirect_stdout(io.StringIO()):
            feats = ref.features_loader(csvp)
    raw = os.path.join(tmp, fname)
    if fname.endswith(".gz"):
        import gzip
        with gzip.open(raw, "wb") as f:
            f.write(fastq_bytes)
    else:
        with open(raw, "wb") as f:
            f.write(fastq_bytes)
    reads_stats = {"failed_reads": set(), "passed_reads": {}}
    with contextlib.redirect_stdout(io.StringIO()):
        out = ref.reads_counter(0, raw, feats, param, reads_stats)
    feats, _, st = out
    return {
        "rows": [[feats[k].name, k, int(feats[k].counts)] for k in feats],      # dict order
        "stats": [int(st["reads"]), int(st["perfect_counter"]), int(st["imperfect_counter"]),
                  int(st["non_aligned_counter"]), int(st["quality_failed"])],
    }


def rec(seq, qual=None, name="r"):
    qual = ("I" * len(seq)) if qual is None else qual
    return f"@{name}\n{seq}\n+\n{qual}\n"


def build_cases():
    cases = []

    def add(name, params, features, fastq, synth_spec=None, note=""):
        cases.append({"name": name, "params": params, "features": features, "fastq": fastq,
                      "synth": synth_spec, "note": note})

    # ---- edge set -----------------------------------------------
    lib5 = [["g1", "AAAAAAAAAA"], ["g2", "AAAAAAAATT"], ["g3", "CCCCCCCCCC"], ["g4", "GGGGGGGGGG"], ["g5", "GGGGGGGGGT"]]
    t = "CCCCCCCCCC"
    edge = "".join([
        rec("AAAAAAAAAA" + t), rec("AAAAAAAAAA" + t, ">" + "I" * 19), rec("AAAAAAAAAA" + t, "=" + "I" * 19),
        rec("AAAAAAAAAA" + t, "I" * 10 + "#" * 10), rec("CCCCCCCCCA" + t), rec("AAAAAAAAAT" + t),
        rec("GGGGGGGGGT" + t), rec("CCCCCCCCAA" + t), rec("aaaaaaaaaa" + t), rec("CCCCNCCCCC" + t),
        rec("CCCCCC"), rec("ACGTACGTAC" + t)])
    for m in (0, 1, 2, 3):
        add(f"edge_m{m}", {"miss": m, "length": 10}, lib5, edge)
    for ph in (-5, 0, 1, 2, 3, 29, 30, 31, 41, 42, 94, 95, 96, 200):
        add(f"edge_ph{ph}", {"miss": 1, "length": 10, "phred": ph}, lib5, edge)

    # ---- Phred sweep: every printable quality char once in the window ---------------------
    qs = "".join(chr(c) for c in range(33, 127))
    sweep = "".join(rec("AAAAAAAAAA" + t, ch + "I" * 19) for ch in qs) + rec("AAAAAAAAAA" + t, "\x1f" + "I" * 19) + \
        rec("AAAAAAAAAA" + t, "\x7f" + "I" * 19)
    for ph in (1, 10, 30, 60, 94, 95, 100):
        add(f"phred_sweep_ph{ph}", {"miss": 0, "length": 10, "phred": ph}, lib5, sweep)

    # ---- anchor set --------------------------------------------------------------------
    U, D = "ACCG", "GTTT"
    anchor = "".join([
        rec("TT" + U + "A" * 10 + D + "CC"), rec("TT" + "ACCT" + "A" * 10 + D + "CC"),
        rec("TT" + U + "A" * 10 + D + "CC", "II#III" + "I" * 16), rec("TT" + U + "A" * 10 + D + "CC", "I" * 6 + "#" + "I" * 15),
        rec("TT" + "accg" + "A" * 10 + D + "CC"), rec("TT" + U + "A" * 10 + "CCCCCC"),
        rec(U + "CC" + U + "GGGG" + D + "CCCCCC"), rec(U + "CCCCC" + D + "A" * 9), rec(U + D + "A" * 14)])
    add("anchor_ec_both_0", {"Running Mode": "EC", "upstream": U, "downstream": D}, None, anchor)
    add("anchor_ec_both_1", {"Running Mode": "EC", "upstream": U, "downstream": D, "miss_search_up": 1,
                             "miss_search_down": 1}, None, anchor)
    add("anchor_ec_up", {"Running Mode": "EC", "upstream": U, "length": 10}, None, anchor)
    add("anchor_ec_down", {"Running Mode": "EC", "downstream": D, "length": 10}, None, anchor)
    add("anchor_c_up_m1", {"upstream": U, "length": 10, "miss": 1}, lib5, anchor)
    add("anchor_ec_both_q", {"Running Mode": "EC", "upstream": U, "downstream": D, "qual_up": 1, "qual_down": 40},
        None, anchor)
    add("anchor_ec_lower_anchor", {"Running Mode": "EC", "upstream": "accg", "downstream": "gttt"}, None, anchor)

    # ---- multi-window and multi-anchor ---------------------------------------------------
    libmw = [["d1", "AAAAA:CCCCC"], ["s1", "AAAAA"], ["s2", "CCCCC"]]
    mw = "".join([rec("AAAAACCCCCGG"), rec("AAAAACCCCCGG", "IIIII#IIIIII"), rec("AAAAACCCCCGG", "#" + "I" * 11),
                  rec("AAAATCCCCCGG"), rec("AAAAACCCCTGG"), rec("TTTTTCCCCCGG")])
    for m in (0, 1, 2):
        add(f"multiwindow_m{m}", {"start": "0,5", "length": 5, "miss": m}, libmw, mw)
    add("multiwindow_ec", {"start": "0,5", "length": 5, "Running Mode": "EC"}, None, mw)
    ref_read = "AAAAAACACACACACACACACATTCAGGGGGGCCAAAAATAGAGAGAGAGAGACCGAGAGGGGGTTAGCATCG"
    ma = rec(ref_read, "B" * len(ref_read)) + rec(ref_read.lower(), "B" * len(ref_read)) + rec(ref_read[:40], "B" * 40)
    add("multianchor_ec", {"Running Mode": "EC", "upstream": "CACACATT,GAGACCGA", "downstream": "TAGAGAGA,TAGCATCG",
                           "phred": 30}, None, ma)
    add("multianchor_c", {"upstream": "CACACATT,GAGACCGA", "downstream": "TAGAGAGA,TAGCATCG", "miss": 2},
        [["x", "CAGGGGGGCCAAAAA:GAGGGGGT"], ["y", "CAGGGGGGCCAAAAA"], ["z", "GAGGGGGA"]], ma)

    # ---- framing oddities (A0) -------------------------------------------------------------
    base = rec("AAAAAAAAAA" + t) + rec("GGGGGGGGGT" + t)
    add("framing_crlf", {"length": 10}, lib5, base.replace("\n", "\r\n"))
    add("framing_no_final_newline", {"length": 10}, lib5, base[:-1])
    add("framing_partial_record", {"length": 10}, lib5, base + "@x\nAAAAAAAAAA\n+\n")
    add("framing_blank_line_shift", {"length": 10}, lib5, "\n" + base + base)
    add("framing_trailing_ws", {"length": 10}, lib5, rec("AAAAAAAAAA  \t", "IIIIIIIIII  ") + rec("AAAAAAAAAA", "IIIIIIIIII\t\t"))
    add("framing_empty", {"length": 10}, lib5, "")
    add("framing_qual_shorter", {"length": 10}, lib5, rec("AAAAAAAAAA" + t, "IIIII") + rec("AAAAAAAAAA" + t, ""))
    add("framing_qual_longer", {"length": 10}, lib5, rec("AAAAAAAA", "IIIIIIII#I") + rec("AAAAAAAAAA", "I" * 10 + "#"))
    add("framing_empty_seq", {"length": 10, "Running Mode": "EC"}, None, rec("", "") + rec("ACGT", "IIII"))
    add("window_past_end", {"length": 10, "start": "15", "Running Mode": "EC"}, None,
        rec("AAAAAAAAAA" + t) + rec("ACGTACGTACGTACGTAC") + rec("ACGT"))
    add("gz_input", {"length": 10}, lib5, edge)

    # ---- irregular libraries: mixed lengths, N / IUPAC symbols, lower case + blanks in csv -----
    libirr = [["a", "AAAAAAAAAA"], ["n", "AAAANAAAAA"], ["short", "AAAAAA"], ["low", "ccccc ccccc"], ["r", "GGGGRGGGGG"]]
    irr = "".join([rec("AAAAAAAAAA" + t), rec("AAAANAAAAA" + t), rec("AAAAnAAAAA" + t), rec("AAAATAAAAA" + t),
                   rec("AAAAAA"), rec("CCCCCCCCCC"), rec("GGGGRGGGGG"), rec("GGGGAGGGGG"), rec("AAAAAT")])
    for m in (0, 1, 2):
        add(f"irregular_m{m}", {"length": 10, "miss": m}, libirr, irr)
    libdup = [["a", "AAAAAAAAAA"], ["b", "aaaaaaaaaa"], ["c", "CCCCCCCCCC"], ["a", "GGGGGGGGGG"]]
    add("library_duplicates", {"length": 10, "miss": 1}, libdup, edge)

    # ---- seeded synthetic workloads (input regenerated from tests/synth.py by the tests) ----
    def synth_case(name, params, n_guides, glen, lib_seed, spec_kw, ec=False, mutate_lib=None):
        guides = synth.make_library(n_guides, glen, lib_seed)
        spec = synth.Spec(**spec_kw)
        fq = synth.make_fastq(spec, guides).decode()
        feats = None if ec else [[f"g{j:06d}", s] for j, s in enumerate(guides)]
        add(name, params, feats, None,
            synth_spec={"n_guides": n_guides, "glen": glen, "lib_seed": lib_seed, "spec": spec_kw}, note="synthetic")
        cases[-1]["_fastq_runtime"] = fq

    for m in (0, 1, 2, 3):
        synth_case(f"synth_fixed_m{m}", {"miss": m}, 300, 20, 0xF2A5 + 2, dict(seed=11 + m, n_reads=6000))
    synth_case("synth_fixed_st7_l12_m1", {"miss": 1, "start": "7", "length": 12}, 200, 12, 77,
               dict(seed=5, n_reads=4000, start=7))
    synth_case("synth_fixed_dense_m2", {"miss": 2, "length": 8}, 600, 8, 78, dict(seed=6, n_reads=4000, p_sub=0.3))
    synth_case("synth_fixed_short_reads", {"miss": 1, "start": "9"}, 100, 20, 79,
               dict(seed=7, n_reads=2000, read_len=32, start=9))
    synth_case("synth_fixed_clipped_reads", {"miss": 1, "start": "9"}, 100, 20, 79,
               dict(seed=7, n_reads=1000, read_len=27, start=9))
    synth_case("synth_fixed_clipped_reads_ec", {"Running Mode": "EC", "start": "9"}, 100, 20, 79,
               dict(seed=7, n_reads=1000, read_len=27, start=9), ec=True)
    up, down = "GTTTAAGAGCTA", "CGTTACCAGGTT"
    cas = dict(cassette=True, up=up, down=down, max_offset=100)
    synth_case("synth_anchor_both_c", {"miss": 1, "upstream": up, "downstream": down, "miss_search_up": 1,
                                       "miss_search_down": 1}, 300, 20, 80, dict(seed=8, n_reads=5000, **cas))
    synth_case("synth_anchor_both_ec", {"Running Mode": "EC", "upstream": up, "downstream": down, "miss_search_up": 1,
                                        "miss_search_down": 1}, 300, 20, 80, dict(seed=8, n_reads=5000, **cas), ec=True)
    synth_case("synth_anchor_up_c", {"miss": 1, "upstream": up}, 300, 20, 81, dict(seed=9, n_reads=5000, **cas))
    synth_case("synth_anchor_down_c", {"miss": 2, "downstream": down, "miss_search_down": 2}, 300, 20, 82,
               dict(seed=10, n_reads=5000, **cas))
    synth_case("synth_anchor_up_ec_ms2", {"Running Mode": "EC", "upstream": up, "miss_search_up": 2, "length": 15},
               300, 20, 83, dict(seed=12, n_reads=3000, **cas), ec=True)
    synth_case("synth_fixed_ec", {"Running Mode": "EC"}, 300, 20, 84, dict(seed=13, n_reads=5000), ec=True)
    return cases


def main():
    ref = import_reference()
    out_cases = []
    for c in build_cases():
        fq = c.pop("_fastq_runtime", None)
        fastq_text = fq if fq is not None else c["fastq"]
        fname = "s.fastq.gz" if c["name"] == "gz_input" else "s.fastq"
        exp = run_reference(ref, c["params"], c["features"], fastq_text.encode("latin-1"), fname)
        c["expected"] = exp
        out_cases.append(c)
        print(f"{c['name']:32s} stats={exp['stats']} keys={len(exp['rows'])}")
    with open(os.path.join(HERE, "reads_counter_cases.json"), "w") as f:
        json.dump({"generator": "tests/golden/make_golden.py", "reference": "afombravo/2FAST2Q v2.8.1 fast2q.py "
                   "(reads_counter :514, run JIT-less)", "cases": out_cases}, f, indent=0, separators=(",", ":"))

    # known-answer vectors for the two search helpers, straight from the reference functions
    kat = {"border_finder": [], "sequence_tinder": []}
    import random
    rng = random.Random(1234)
    for _ in range(300):
        r = "".join(rng.choice("ACGT") for _ in range(rng.randint(0, 40)))
        s = "".join(rng.choice("ACGT") for _ in range(rng.randint(1, 8)))
        if rng.random() < 0.5 and len(r) > len(s):
            p = rng.randint(0, len(r) - len(s))
            r = r[:p] + s + r[p + len(s):]
        m, sp = rng.randint(0, 2), rng.randint(0, max(0, len(r)))
        got = ref.border_finder(ref.seq2bin(s), ref.seq2bin(r), m, sp)
        kat["border_finder"].append([s, r, m, sp, got])
    with open(os.path.join(HERE, "search_kat.json"), "w") as f:
        json.dump(kat, f, separators=(",", ":"))
    print("wrote", len(out_cases), "cases")


if __name__ == "__main__":
    main()
