#!/opt/conda/bin/python3.9
"""tests/golden/make_golden_compiling.py -- fixtures for the OUTPUT CONTRACT of the path, made by RUNNING THE REFERENCE.

For a few multi-sample runs (Counter mode with numeric and with string feature names, Extract+Count with different
key sets per sample -> zero back-fill, a .fastq.gz sample) the reference's own `aligner` (fast2q.py:752) is run on every
FASTQ file and then its `compiling` (:1316) + `run_stats` (:1386); the bytes of `compiled.csv` and `compiled_stats.csv`
are stored next to the inputs in tests/golden/compiling_cases.json.  The reference is imported from /root/reference
exactly as make_golden.py does (JIT-less numba shim); no reference source is copied.

The only thing made deterministic is the clock: `time.perf_counter` of the reference module is replaced by a scripted
one, so that the "script ran in ..." texts (seconds / minutes / hours forms) are reproducible; the tests give the
harness the same clock.

Usage:  /opt/conda/bin/python3.9 tests/golden/make_golden_compiling.py
"""
import contextlib
import gzip
import io
import json
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import synth  # noqa: E402
from make_golden import cli_defaults, import_reference, rec  # noqa: E402

# elapsed seconds of the samples, in processing order (cycled): all three wordings of aligner's timing text
ELAPSED = [0.5249, 61.8, 3700.0, 12.0]


class Clock:
    """perf_counter stand-in: aligner reads it twice per sample (start, stop)"""

    def __init__(self):
        self.t, self.k, self.started = 100.0, 0, False

    def __call__(self):
        if not self.started:
            self.started = True
            return self.t
        self.started = False
        self.t += ELAPSED[self.k % len(ELAPSED)]
        self.k += 1
        return self.t


def build_cases():
    cases = []
    lib_num = [["10", "AAAAAAAAAA"], ["9", "AAAAAAAATT"], ["100", "CCCCCCCCCC"], ["2", "GGGGGGGGGG"], ["33", "GGGGGGGGGT"]]
    lib_str = [["gB", "AAAAAAAAAA"], ["gA", "AAAAAAAATT"], ["g10", "CCCCCCCCCC"], ["g9", "GGGGGGGGGG"], ["Zed", "GGGGGGGGGT"]]
    t = "CCCCCCCCCC"
    s1 = "".join([rec("AAAAAAAAAA" + t)] * 3 + [rec("GGGGGGGGGT" + t), rec("CCCCCCCCCA" + t), rec("ACGTACGTAC" + t),
                                               rec("AAAAAAAAAA" + t, "#" + "I" * 19)])
    s2 = "".join([rec("GGGGGGGGGG" + t)] * 2 + [rec("AAAAAAAATT" + t), rec("AAAAAAAATA" + t)])
    s3 = "".join([rec("CCCCCCCCCC" + t)] * 5)
    for name, lib in (("counter_numeric_names", lib_num), ("counter_string_names", lib_str)):
        cases.append({"name": name, "params": {"miss": 1, "length": 10}, "features": lib,
                      "files": [["b_sample.fastq", s1], ["a_sample.fastq.gz", s2], ["a_sample_2.fastq", s3]]})
    U, D = "ACCG", "GTTT"
    e1 = "".join([rec("TT" + U + "A" * 10 + D + "CC"), rec("TT" + U + "C" * 7 + D + "CC"), rec("TT" + U + "A" * 10 + D + "CC"),
                  rec(U + D + "A" * 14), rec("TTTTTTTTTTTTTTTTTTTT")])
    e2 = "".join([rec("TT" + U + "G" * 9 + D + "CC"), rec("TT" + U + "A" * 10 + D + "CC"), rec("TT" + U + "ACGTAC" + D)])
    e3 = "".join([rec("TT" + U + "C" * 7 + D + "CC")] * 4 + [rec("TT" + U + "TTTTT" + D + "CC")])
    cases.append({"name": "extract_count_backfill", "params": {"Running Mode": "EC", "upstream": U, "downstream": D},
                  "features": None, "files": [["x1.fastq", e1], ["x2.fastq", e2], ["x0.fastq", e3]]})
    guides = synth.make_library(40, 20, 4242)
    files = []
    for k in range(3):
        fq = synth.make_fastq(synth.Spec(seed=70 + k, n_reads=400 + 100 * k, read_len=60, start=5), guides).decode()
        files.append([f"synth_{k}.fastq", fq])
    cases.append({"name": "counter_synth_three_samples", "params": {"miss": 1, "start": "5", "used_cmd": "--c --s x --m 1"},
                  "features": [[f"sg{j:03d}", s] for j, s in enumerate(guides)], "files": files})
    return cases


def run_case(ref, case):
    tmp = tempfile.mkdtemp(prefix="f2q_goldc_")
    seqdir = os.path.join(tmp, "in")
    os.makedirs(seqdir)
    p = cli_defaults()
    p.update(case["params"])
    p["out"] = tmp
    p["seq_files"] = seqdir
    p["Progress bar"] = False
    with contextlib.redirect_stdout(io.StringIO()):
        param = ref.initializer(dict(p))
    os.makedirs(param["directory"], exist_ok=True)
    feats = {}
    if param["Running Mode"] == "C":
        csvp = os.path.join(tmp, "features.csv")
        with open(csvp, "w") as f:
            for name, seq in case["features"]:
                f.write(f"{name},{seq}\n")
        with contextlib.redirect_stdout(io.StringIO()):
            feats = ref.features_loader(csvp)
    ref.time.perf_counter = Clock()
    samples = []
    for i, (fname, text) in enumerate(case["files"]):
        raw = os.path.join(seqdir, fname)
        if fname.endswith(".gz"):
            with gzip.open(raw, "wb") as f:
                f.write(text.encode("latin-1"))
        else:
            with open(raw, "wb") as f:
                f.write(text.encode("latin-1"))
        # every sample starts from zeroed counts, as the per-process copy of `features` does upstream (:1646-1655)
        per_sample = {k: ref.Features(v.name, 0) for k, v in feats.items()} if param["Running Mode"] == "C" else {}
        reads_stats = {"failed_reads": set(), "passed_reads": {}}
        with contextlib.redirect_stdout(io.StringIO()):
            ref.aligner(i, raw, per_sample, param, reads_stats)
        samples.append({"file": fname, "rows": [[per_sample[k].name, k, int(per_sample[k].counts)] for k in per_sample]})
    reads_csv = {}
    for f in sorted(os.listdir(param["directory"])):
        if f.endswith("_reads.csv"):
            reads_csv[f] = open(os.path.join(param["directory"], f), "rb").read().decode("latin-1")
    with contextlib.redirect_stdout(io.StringIO()):
        ref.compiling(param)
    out = {"reads_csv": reads_csv, "samples": samples, "version": param["version"]}
    for f in ("compiled.csv", "compiled_stats.csv"):
        out[f] = open(os.path.join(param["directory"], f), "rb").read().decode("latin-1")
    out["files_left"] = sorted(os.listdir(param["directory"]))
    return out


def truncated_gzip_cases(ref):
    """a .fastq.gz cut off at several byte positions (fast2q.py:405-407, 577-582): what the reference still counts.
    Stored: the cut archive (base64) and the reference's counts/stats, or "none" when reads_counter gave up."""
    import base64
    from make_golden import run_reference
    guides = synth.make_library(30, 20, 99)
    fq = synth.make_fastq(synth.Spec(seed=31, n_reads=1500, read_len=50), guides)
    lib = [[f"t{j}", s] for j, s in enumerate(guides)]
    out = []
    for level in (1, 6):
        raw = gzip.compress(fq, compresslevel=level, mtime=0)
        for frac in (0.13, 0.5, 0.77, 0.999):
            cut = raw[: int(len(raw) * frac)]
            tmp = tempfile.mkdtemp(prefix="f2q_goldt_")
            path = os.path.join(tmp, "cut.fastq.gz")
            with open(path, "wb") as f:
                f.write(cut)
            p = cli_defaults(); p["out"] = tmp; p["seq_files"] = tmp
            with contextlib.redirect_stdout(io.StringIO()):
                param = ref.initializer(dict(p))
            csvp = os.path.join(tmp, "features.csv")
            with open(csvp, "w") as f:
                for name, seq in lib:
                    f.write(f"{name},{seq}\n")
            with contextlib.redirect_stdout(io.StringIO()):
                feats = ref.features_loader(csvp)
                res = ref.reads_counter(0, path, feats, param, {"failed_reads": set(), "passed_reads": {}})
            exp = "none" if res is None else {"counts": [int(v.counts) for v in res[0].values()],
                                              "stats": [int(res[2][k]) for k in ("reads", "perfect_counter", "imperfect_counter",
                                                                                 "non_aligned_counter", "quality_failed")]}
            out.append({"level": level, "frac": frac, "gz_b64": base64.b64encode(cut).decode(), "features": lib, "expected": exp})
            print("truncated gzip level", level, "cut at", frac, "->", exp if exp == "none" else exp["stats"])
    return out


def main():
    ref = import_reference()
    with open(os.path.join(HERE, "truncated_gzip_cases.json"), "w") as f:
        json.dump({"generator": "tests/golden/make_golden_compiling.py", "cases": truncated_gzip_cases(ref)}, f, separators=(",", ":"))
    out = []
    for case in build_cases():
        case["expected"] = run_case(ref, case)
        out.append(case)
        print(case["name"], "compiled.csv", len(case["expected"]["compiled.csv"]), "bytes;", case["expected"]["files_left"])
    with open(os.path.join(HERE, "compiling_cases.json"), "w") as f:
        json.dump({"generator": "tests/golden/make_golden_compiling.py", "elapsed": ELAPSED,
                   "reference": "afombravo/2FAST2Q v2.8.1 fast2q.py (aligner :752, compiling :1316, run_stats :1386)",
                   "cases": out}, f, indent=0, separators=(",", ":"))
    print("wrote", len(out), "cases")


if __name__ == "__main__":
    main()
