"""Host harness of the MI355X-native 2FAST2Q counting path.

This module mirrors the reference's interface for the path (afombravo/2FAST2Q v2.8.1,
fast2q/fast2q.py) -- same function names, argument meaning, return shapes, output files -- so that
an existing ``2fast2q -c ...`` command line, features CSV and ``compiled.csv`` consumer keep working:

    features_loader (:125)   reads_counter (:514, the drop-in seam)   aligner (:752)
    csv_writer (:803)        compiling (:1316)   run_stats (:1386)    input_parser (:1171)
    initializer (:1082)      file_sizer_split (:1657)   main (:1691)

What is different: reads_counter() does no per-read work in Python.  It hands the file to
libf2q_hip.so (include/f2q.h) through ctypes; FASTQ framing, Phred masking, window / anchored
extraction, <= m mismatch matching and count accumulation all run in the library (HIP kernels on
gfx950).  The reference's Numba helpers, multiprocessing pools, memo caches and chunk splitter
have no counterpart here (--cp = samples in flight, as threads; --fs is accepted and ignored).  There is no CPU fallback: without
the library or a GPU, reads_counter raises.

Multi-GPU: when launched with one process per GPU (torchrun; RANK/WORLD_SIZE set) every rank
counts its share of each file's record blocks and the int64 count vector is summed with one
all-reduce (sharding.py).
"""
import argparse
import atexit
import csv
import datetime
import glob
import os
import sys
import threading
import time
from dataclasses import dataclass
from pathlib import Path

from . import binding, sharding

version = "2.8.1+mi355x.0.1"


@dataclass
class Features:
    """name / read count of one feature; instances live in a dict keyed by sequence (:21-44)"""
    name: str
    counts: int


def colourful_errors(warning_type, error):
    """timestamped INFO / WARNING / FATAL line (:46-67); no colour codes (colorama is optional upstream only)"""
    print(f" {datetime.datetime.now().strftime('%c')} [{warning_type}] {error}")


def path_finder(folder_path, extension):
    found = []
    for ext in extension:
        for filename in glob.glob(os.path.join(folder_path, ext)):
            found.append([filename, os.path.getsize(filename)])
    return found


def path_parser(folder_path, extension):
    """files of the given patterns, by size; the *reads.csv temporaries by name (:91-123)"""
    pathing = path_finder(folder_path, extension)
    if extension != ['*reads.csv']:
        ordered = sorted(pathing, key=lambda e: e[-1])
        if not ordered:
            colourful_errors("FATAL", f"Check the path to the {extension} files folder. No files of this type found.\n")
            sys.exit()
        return ordered
    return [p[0] for p in sorted(pathing)]


def features_loader(guides):
    """{SEQUENCE: Features(name, 0)} from a 2-column csv (:125-186): separators ',' ';' tab are tried in
    turn; sequences are upper-cased with blanks removed; a repeated sequence keeps its first name."""
    colourful_errors("INFO", "Loading Features")
    if not os.path.isfile(guides):
        colourful_errors("FATAL", f"Check the path to the features file.\nNo .csv file found in the following path: {guides}\n")
        sys.exit()
    features = {}
    for sep in (",", ";", "\t"):
        names = set()
        try:
            with open(guides) as handle:
                for line in handle:
                    cols = line.rstrip().split(sep)
                    sequence = cols[1].upper().replace(" ", "")
                    name = cols[0]
                    if name in names:
                        colourful_errors("WARNING", f"The name {name} seems to appear at least twice. This MIGHT result in unexpected behaviour. Please have only unique name entries in your features.csv file.")
                    if sequence not in features:
                        features[sequence] = Features(name, 0)
                        names.add(name)
                    else:
                        colourful_errors("WARNING", f"{features[sequence].name} and {name} share the same sequence. Only {features[sequence].name} will be considered valid. {name} will be ignored.")
        except IndexError:
            pass
    if not features:
        colourful_errors("FATAL", "The given .csv file doesn't seem to be comma, semicolon, or tab separated. Please double check that the file's column separation\n")
        sys.exit()
    colourful_errors("INFO", f"{len(features)} different features were provided.")
    return features


def _counter_kwargs(param):
    return dict(mode=param['Running Mode'], miss=param['miss'], phred=param['phred'], length=param['length'],
                start=param['start'], upstream=param['upstream'], downstream=param['downstream'],
                miss_search_up=param['miss_search_up'], miss_search_down=param['miss_search_down'],
                qual_up=param['qual_up'], qual_down=param['qual_down'],
                device=int(param.get('device', os.environ.get("F2Q_DEVICE", os.environ.get("LOCAL_RANK", 0)))))


# ---- contexts are kept between samples ----------------------------------------------------------------------------
# A run counts many samples against ONE library with ONE set of parameters.  Creating a context per sample would build and
# upload the library index and allocate (and pin) the staging buffers every time -- 50 ms to several hundred per sample, as
# much as counting a small sample takes.  Each worker thread keeps its context; it is reset between samples
# (f2q_reset_counts: accumulators, Extract+Count tables, read numbering) and closed when the run ends.
_CTX_LOCK = threading.Lock()
_CTX_ALL = []
_CTX_TLS = threading.local()
_CTX_GEN = [0]


def _context_for(seqs, kwargs):
    key = (tuple(sorted((k, str(v)) for k, v in kwargs.items())), None if seqs is None else tuple(seqs))      # (compared by value: a few ms for 100 k features)
    cur = getattr(_CTX_TLS, "entry", None)
    if os.environ.get("F2Q_NO_CTX_CACHE") == "1":              # A/B runs: a fresh context per sample
        key = (key, object())
    if cur is not None and cur[0] == _CTX_GEN[0] and cur[1] == key:
        cur[2].reset()
        return cur[2]
    if cur is not None and cur[0] == _CTX_GEN[0]:
        _drop_context(cur[2])
    ctx = binding.Counter(features=seqs, **kwargs)
    _CTX_TLS.entry = (_CTX_GEN[0], key, ctx)
    with _CTX_LOCK:
        _CTX_ALL.append(ctx)
    return ctx


def _drop_context(ctx):
    with _CTX_LOCK:
        if ctx in _CTX_ALL:
            _CTX_ALL.remove(ctx)
    _CTX_TLS.entry = None
    ctx.close()


def close_contexts():
    """close every context kept by reads_counter (end of a run; also registered with atexit)"""
    with _CTX_LOCK:
        ctxs, _CTX_ALL[:] = list(_CTX_ALL), []
        _CTX_GEN[0] += 1
    for c in ctxs:
        c.close()


atexit.register(close_contexts)


def reads_counter(i, raw, features, param, reads_stats, preprocess=False):
    """Counts one FASTQ(.gz) file (:514-582).  Returns (features, reads_stats, local_read_stats) -- the reference's
    contract.  A cut-off or damaged .gz gives the counts of every complete record before the damage plus the
    reference's warning (its parser keeps what it counted when readline raises, :405-407).  `features` is updated in
    place: Counter mode adds to Features.counts, Extract+Count mode adds the de-novo keys in first-occurrence order."""
    if (param['upstream'] is not None) and (param['downstream'] is not None):
        if len(str(param['upstream']).split(",")) != len(str(param['downstream']).split(",")):
            colourful_errors("FATAL", "Up and Downstream sequences must be submitted in concurrent pairs, separated by ,.")
            sys.exit()
    local_read_stats = dict.fromkeys(binding.STAT_NAMES, 0)
    if preprocess:                                  # memo warm-up (:1593-1617) has nothing to warm here
        return features, reads_stats, local_read_stats
    counter_mode = param['Running Mode'] == 'C'
    seqs = list(features) if counter_mode else None
    ctx = _context_for(seqs, _counter_kwargs(param))
    try:
        world = sharding.world()
        if world.size > 1:
            truncated = sharding.count_file_sharded(ctx, raw, world)
            counts, stats, ec_rows = sharding.reduce_results(ctx, world)
        else:
            _, truncated = ctx.count_file(raw)
            counts, stats = ctx.read_counts()
            ec_rows = None if counter_mode else ctx.ec_results()
    except BaseException:
        _drop_context(ctx)                          # whatever state the failure left: the next sample starts afresh
        raise
    if truncated:
        colourful_errors("WARNING", f"{raw} is an incomplete or corrupted gzip file. Only partial processing might have occurred.")
    if counter_mode:
        for seq, n in zip(seqs, counts):
            features[seq].counts += int(n)
    else:
        for key, n, _first in ec_rows:
            if key in features:
                features[key].counts += n
            else:
                features[key] = Features(key, n)
    for k, v in zip(binding.STAT_NAMES, stats):
        local_read_stats[k] = int(v)
    return features, reads_stats, local_read_stats


@dataclass
class SampleResult:
    """What one sample contributes to the run's tables: kept in memory from aligner() to compiling()
    (the reference round-trips this through <sample>_reads.csv and an English sentence, :785-799, :1316-1406)."""
    name: str            # file stem without .fastq / .gz (:779-783)
    time_value: str      # "0.52" / "1.03" ...
    time_unit: str       # seconds / minutes / hours
    rows: list           # [feature name, reads] in the sample's row order (numeric or alphabetical, :790-793)
    stats: dict          # the five counters of reads_counter (:310-316)

    def sentence(self):
        """first line of <sample>_reads.csv, as upstream words it (:785)"""
        st = self.stats
        return (f'#script ran in {self.time_value} {self.time_unit} for file {self.name}. '
                f'{st["perfect_counter"] + st["imperfect_counter"]} reads out of {st["reads"]} were aligned. '
                f'{st["perfect_counter"]} were perfectly aligned. '
                f'{st["imperfect_counter"]} were aligned with mismatch. '
                f'{st["non_aligned_counter"]} passed quality filtering but were not aligned. '
                f'{st["quality_failed"]} did not pass quality filtering.')

    def reads_csv_rows(self):
        return [[self.sentence()], ["#Feature", "Reads"]] + [list(r) for r in self.rows]


def _sample_name(raw):
    name = Path(raw).stem
    return Path(name).stem if ".fastq" in name else name


def _elapsed_text(seconds):
    """(value, unit) of aligner's running time (:771-777)"""
    if seconds > 3600:
        return str(round(seconds / 3600, 2)), "hours"
    if seconds > 60:
        return str(round(seconds / 60, 2)), "minutes"
    return str(round(seconds, 2)), "seconds"


def aligner(i, raw, features, param, reads_stats):
    """One sample (:752-801): count it, order its rows, and keep the result for compiling() in param["samples"].
    <sample>_reads.csv is only written when the temporaries are kept (--k): nothing reads it back."""
    started = time.perf_counter()
    packed = reads_counter(i, raw, features, param, reads_stats)
    if packed is None:
        return reads_stats
    features, reads_stats, local = packed
    rows = [[f.name, f.counts] for f in features.values()]
    value, unit = _elapsed_text(time.perf_counter() - started)
    try:
        rows.sort(key=lambda row: int(row[0]))       # all names numeric: numerical order
    except ValueError:
        rows.sort(key=lambda row: row[0])
    sample = SampleResult(_sample_name(raw), value, unit, rows, dict(local))
    if not param['Progress bar']:
        colourful_errors("INFO", f"Sample {sample.name} was processed in {value} {unit}")
    param.setdefault("samples", {})[sample.name] = sample     # a later file of the same name replaces the earlier one,
    if not param.get("delete", True) and sharding.world().rank == 0:   # as its _reads.csv would upstream
        csv_writer(os.path.join(param["directory"], sample.name + "_reads.csv"), sample.reads_csv_rows())
    return reads_stats


def csv_writer(path, outfile):
    """csv.writer rows (CRLF line ends, like the reference :803-809)"""
    with open(path, "w", newline='') as output:
        csv.writer(output).writerows(outfile)


def initializer(cmd):
    """banner-less counterpart of :1082-1169: coerces the Phred inputs, creates the output directory name"""
    param = cmd
    if param is None:
        colourful_errors("FATAL", "Only the command line mode (-c) is provided by this build; the tkinter dialog of the reference is not.")
        sys.exit(2)
    print(f"\n 2FAST2Q (MI355X counting path)  Version: {version}")
    if param["test_mode"]:
        colourful_errors("WARNING", "Running test mode!\n")
    param["version"] = version
    for key in ("phred", "qual_up", "qual_down"):
        if int(param[key]) <= 0:            # :1118-1125
            param[key] = 1
    current_time = datetime.datetime.now().strftime('%Y_%m_%d_%H_%M_%S')
    param["directory"] = os.path.join(param['out'], f"2FAST2Q_output_{current_time}")
    print("\n -- Parameters -- ")
    if param['Running Mode'] == 'C':
        print("\n Mode: Align and count")
        print(f" Allowed mismatches per alignement: {param['miss']}")
    else:
        print("\n Mode: Extract and count")
    print(f" Minimal Phred Score per bp >= {param['phred']}")
    if param['upstream'] is not None:
        print(f" Upstream search sequence: {param['upstream']}")
        print(f" Mismatches allowed in the upstream search sequence: {param['miss_search_up']}")
        print(f" Minimal Phred-score in the upstream search sequence: {param['qual_up']}")
    if param['downstream'] is not None:
        print(f" Downstream search sequence: {param['downstream']}")
        print(f" Mismatches allowed in the downstream search sequence: {param['miss_search_down']}")
        print(f" Minimal Phred-score in the downstream search sequence: {param['qual_down']}")
    if (param['upstream'] is None) or (param['downstream'] is None):
        print(f" Finding features with the folowing length: {param['length']}bp")
    if (param['upstream'] is None) and (param['downstream'] is None):
        print(f" Read alignment start position: {param['start']}")
    print(f" All data will be saved into {param['directory']}")
    print("\n ---- ")
    param["cpu"] = param["cpu"] if isinstance(param["cpu"], int) and param["cpu"] > 0 else (os.cpu_count() or 1)
    return param


def _package_data(name):
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", name)


def ensure_example_fastq():
    """The upstream demo FASTQ is not redistributed (.MISSING_LARGE_BLOBS); test mode synthesises a
    stand-in of 20,000 reads from the packaged D39V library with the §8(d) generator."""
    import gzip
    path = _package_data("example.fastq.gz")
    if os.path.exists(path):
        return path
    feats = {}
    with open(_package_data("D39V_guides.csv")) as h:
        for line in h:
            cols = line.rstrip().split(",")
            feats.setdefault(cols[1].upper().replace(" ", ""), cols[0])
    with binding.Counter(features=list(feats), miss=1) as ctx:
        fq = bytes(ctx.synth_fastq(seed=0xD39, n_reads=20000, read_len=75))
    try:
        with gzip.open(path, "wb", compresslevel=6) as f:
            f.write(fq)
    except OSError:
        import tempfile
        path = os.path.join(tempfile.mkdtemp(prefix="f2q_demo_"), "example.fastq.gz")
        with gzip.open(path, "wb", compresslevel=6) as f:
            f.write(fq)
    return path


def input_parser(argv=None):
    """the reference's flag set (:1193-1216) and defaults (:1246-1309), plus --gpu (device ordinal)"""
    ap = argparse.ArgumentParser(prog="2fast2q")
    ap.add_argument("-c", nargs='?', const=True, help="cmd line mode.")
    ap.add_argument("-t", nargs='?', const=True, help="Runs 2FAST2Q in test mode with example data.")
    ap.add_argument("-v", nargs='?', const=True, help="Prints the current version.")
    ap.add_argument("--s", help="The full path to the directory with the sequencing files OR file.")
    ap.add_argument("--g", help="The full path to the .csv file with the sgRNAs.")
    ap.add_argument("--o", help="The full path to the output directory")
    ap.add_argument("--fn", nargs='?', const="compiled", help="Specify an output compiled file name (default is called compiled)")
    ap.add_argument("--pb", nargs='?', const=False, help="Adds progress bars (default is enabled)")
    ap.add_argument("--m", help="The number of allowed mismatches per feature (default = 1). Ignored in extract + Count mode.")
    ap.add_argument("--ph", help="Minimal Phred-score (default=30).")
    ap.add_argument("--st", help="The start position of the feature within the read (default = 0).")
    ap.add_argument("--l", help="The length of the feature in bp (default = 20).")
    ap.add_argument("--us", help="Upstream search sequence.")
    ap.add_argument("--ds", help="Downstream search sequence.")
    ap.add_argument("--msu", help="Upstream search sequence mismatches (default is 0).")
    ap.add_argument("--msd", help="Downstream search sequence mismatches (default is 0).")
    ap.add_argument("--qsu", help="Minimal Phred-score (default=30) in the upstream search sequence")
    ap.add_argument("--qsd", help="Minimal Phred-score (default=30) in the downstream search sequence")
    ap.add_argument("--mo", help="Running Mode (default=C) [Counter (C) / Extractor + Counter (EC)].")
    ap.add_argument("--cp", help="Number of cpus (accepted for compatibility; the counting runs on the GPU)")
    ap.add_argument("--fs", nargs='?', const=False, help="File Split mode (accepted for compatibility; ignored)")
    ap.add_argument("--k", nargs='?', const=False, help="If enabled, keeps all temporary files (default is disabled)")
    ap.add_argument("--gpu", help="HIP device ordinal (default: LOCAL_RANK or 0)")
    args = ap.parse_args(argv)
    if args.v is not None:
        print(f"\nVersion: {version}\n")
        sys.exit()
    if args.c is None:
        return None
    p = {"cmd": True, "big_file_split": args.fs is not None}
    p['used_cmd'] = " ".join(f"--{k}" if isinstance(v, bool) and v else f"--{k} {v}"
                             for k, v in vars(args).items() if v is not None)
    p['Running Mode'] = "EC" if (args.mo is not None and "EC" in args.mo.upper()) else "C"
    if args.t is None:
        p["test_mode"] = False
        paths = [[args.s, 'seq_files'], [args.g, 'feature'], [args.o, 'out']]
    else:
        p["test_mode"] = True
        paths = [[ensure_example_fastq(), 'seq_files'], [_package_data('D39V_guides.csv'), 'feature'], [os.getcwd(), 'out']]
    p['out_file_name'] = args.fn if args.fn is not None else "compiled"
    p['length'] = int(args.l) if args.l is not None else 20
    p['Progress bar'] = args.pb is None
    p['start'] = args.st if args.st is not None else "0"
    p['phred'] = int(args.ph) if args.ph is not None else 30
    p['miss'] = int(args.m) if args.m is not None else 1
    p['upstream'], p['downstream'] = args.us, args.ds
    p['miss_search_up'] = int(args.msu) if args.msu is not None else 0
    p['miss_search_down'] = int(args.msd) if args.msd is not None else 0
    p['qual_up'] = int(args.qsu) if args.qsu is not None else 30
    p['qual_down'] = int(args.qsd) if args.qsd is not None else 30
    p['delete'] = args.k is None
    p['cpu'] = int(args.cp) if args.cp is not None else False
    if args.gpu is not None:
        p['device'] = int(args.gpu)
    for value, key in paths:                                   # :1178-1191
        if value is None:
            p[key] = os.getcwd()
            if key == 'feature':
                found = path_finder(os.getcwd(), ["*.csv"])
                if p['Running Mode'] != "EC":
                    if len(found) > 1:
                        colourful_errors("FATAL", "There is more than one .csv in the current directory. If not directly indicating a path for the features .csv, please have only 1 .csv file in the directory.\n")
                        sys.exit()
                    if len(found) == 1:
                        p[key] = found[0][0]
        else:
            p[key] = value
    return p


def file_sizer_split(param):
    """sample discovery (:1657-1689): *.gz and *.fastq of the --s directory, smallest first"""
    if param["test_mode"]:
        param["sequencing_files"] = {"len_files": 1, "preprocess_files": [param["seq_files"]], "files": [param["seq_files"]]}
        return param
    files = [p[0] for p in path_parser(param["seq_files"], ["*.gz", "*.fastq"])]
    param["sequencing_files"] = {"len_files": len(files), "preprocess_files": files[:1], "files": files}
    return param


def aligner_mp_dispenser(features, param, start=0):
    """every sample through aligner() (:1619-1655); samples are independent, one GPU pass each"""
    if sharding.world().rank == 0:
        os.makedirs(param["directory"], exist_ok=True)
    sharding.barrier()
    reads_stats = {"failed_reads": set(), "passed_reads": {}}
    colourful_errors("INFO", f"Processing {param['sequencing_files']['len_files']} files. Please hold.")

    def one(i, raw):
        # Counter mode: each sample starts from zeroed counts, as the per-process `features` copy does upstream
        per_sample = {k: Features(v.name, 0) for k, v in features.items()} if param['Running Mode'] == 'C' else {}
        aligner(i, raw, per_sample, param, reads_stats)

    files = list(enumerate(param['sequencing_files']['files']))
    # samples in flight: two.  The GPU counts a sample faster than the host reads it, the file reader brings its own
    # worker pool (parallel pread / BGZF / gzip chunks), and every worker thread owns a context with pinned staging
    # buffers: more threads only add allocations and contention (16 samples of 1 M reads, --cp 1 / 2 / 4 / 16: 1.7 / 1.7 /
    # 2.1 / 2.4 s whole run; .gz: 2.4 / 2.0 / 2.4 / 3.1 s -- scripts/samples_threads.py).  F2Q_SAMPLES_IN_FLIGHT overrides.
    cap = int(os.environ.get("F2Q_SAMPLES_IN_FLIGHT", "2") or 2)
    workers = max(1, min(int(param.get("cpu") or 1), len(files), cap, 16))
    try:
        _dispense(files, workers, one)
    finally:
        close_contexts()                            # (the worker threads' contexts outlive the threads otherwise)


def _dispense(files, workers, one):
    if workers > 1 and sharding.world().size == 1:
        # --cp samples in flight, as upstream (:1646-1655) -- threads, not processes: the per-read work is in the
        # library (ctypes drops the GIL), each sample has its own context and HIP stream, and for .gz input the
        # single-threaded inflate of one file overlaps with the inflate and the GPU work of the others
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=workers) as pool:
            for fut in [pool.submit(one, i, raw) for i, raw in files]:
                fut.result()
    else:
        for i, raw in files:
            one(i, raw)


def _samples_from_directory(directory):
    """SampleResults of the <sample>_reads.csv files of a directory (a run whose temporaries were kept, or made by the
    reference itself): the file layout of aligner() read back, for callers that have no in-memory results."""
    found = {}
    for path in path_parser(directory, ['*reads.csv']):
        name = Path(path).stem[:-len("_reads")]
        with open(path, newline='') as handle:
            table = list(csv.reader(handle))
        words = table[0][0].split()
        numbers = [w for w in words if w.isdigit()]        # aligned, reads, perfect, imperfect, non aligned, failed
        stats = dict(zip(("reads", "perfect_counter", "imperfect_counter", "non_aligned_counter", "quality_failed"),
                         (int(numbers[-5]), int(numbers[-4]), int(numbers[-3]), int(numbers[-2]), int(numbers[-1]))))
        found[name] = SampleResult(name, words[3], words[4], [[r[0], int(r[1])] for r in table[2:]], stats)
    return found


def run_headers(param):
    """the '#key: value' lines that open <name>_stats.csv (:1323-1337)"""
    lines = [f"#2FAST2Q version: {param['version']}"]
    if "used_cmd" in param:
        lines.append(f"#cmd used: {param['used_cmd']}")
    lines += [f"#Mismatch: {param['miss']}",
              f"#Phred Score: {param['phred']}",
              f"#Feature Length: {param['length']}",
              f"#Feature start position in the read: {param['start']}",
              f"#Running mode: {param['Running Mode']}",
              f"#Upstream search sequence: {param['upstream']}",
              f"#Downstream search sequence: {param['downstream']}",
              f"#Mismatches in the upstream search sequence: {param['miss_search_up']}",
              f"#Mismatches in the downstream search sequence: {param['miss_search_down']}",
              f"#Minimal Phred-score in the upstream search sequence: {param['qual_up']}",
              f"#Minimal Phred-score in the downstream search sequence: {param['qual_down']}"]
    return lines


def compile_table(samples):
    """(head, {feature: [reads per sample]}) of the run (:1341-1364).  Samples in the order upstream meets their
    <sample>_reads.csv files (sorted by that file name); features in first-met order, a feature a sample lacks counts 0
    there (Extract+Count samples have different key sets).  A feature name holding '#' is left out of the table, as
    upstream's line filter does (:1350)."""
    ordered = sorted(samples.values(), key=lambda s: s.name + "_reads.csv")
    table = {}
    for column, sample in enumerate(ordered):
        for name, reads in sample.rows:
            name = str(name)
            if "#" in name:
                continue
            table.setdefault(name, [0] * column).append(int(reads))
        for counts in table.values():
            counts.extend([0] * (column + 1 - len(counts)))
    return ordered, ["#Feature"] + [s.name for s in ordered], table


def compiling(param):
    """<name>.csv and <name>_stats.csv (+ plots) from the samples aligner() collected (:1316-1384); without in-memory
    results the <sample>_reads.csv files of the output directory are read instead."""
    samples = param.get("samples") or _samples_from_directory(param["directory"])
    ordered, head, table = compile_table(samples)
    run_stats(run_headers(param), param, table, head, ordered)
    csv_writer(os.path.join(param["directory"], f"{param['out_file_name']}.csv"),
               [head] + [[feature] + counts for feature, counts in table.items()])
    if param["delete"]:
        for path in path_finder(param["directory"], ['*reads.csv']):
            os.remove(path[0])
    colourful_errors("INFO", "Analysis successfully completed")
    print("\n If you find 2FAST2Q useful, please consider citing:\n Bravo AM, Typas A, Veening J. 2022. \n 2FAST2Q: a general-purpose sequence search and counting program for FASTQ files. PeerJ 10:e14041\n DOI: 10.7717/peerj.14041\n")
    if param["test_mode"]:
        colourful_errors("WARNING", "Test successful. 2FAST2Q is working as intended!\n")


STATS_HEAD = ["#Sample name", "Running Time", "Running Time unit", "Total number of reads in sample",
              "Total number of reads that were aligned", "Number of reads that were aligned without mismatches",
              "Number of reads that were aligned with mismatches",
              "Number of reads that passed quality filtering but were not aligned",
              'Number of reads that did not pass quality filtering.']


def run_stats(headers, param, compiled, head, ordered):
    """<name>_stats.csv -- the run's '#key: value' lines, the column names, one row of numbers per sample (:1386-1412,
    taken from the counters themselves) -- plus the four overview plots (:1414-1527)"""
    rows = [[s.name, s.time_value, s.time_unit, s.stats["reads"], s.stats["perfect_counter"] + s.stats["imperfect_counter"],
             s.stats["perfect_counter"], s.stats["imperfect_counter"], s.stats["non_aligned_counter"],
             s.stats["quality_failed"]] for s in ordered]
    global_stat = [[line] for line in headers] + [STATS_HEAD] + rows
    csv_writer(os.path.join(param["directory"], f"{param['out_file_name']}_stats.csv"), global_stat)
    try:
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
        import numpy as np
    except Exception as exc:                                     # plots are cosmetic; say so and go on
        colourful_errors("WARNING", f"matplotlib unavailable ({exc}); plots skipped")
        return
    base = os.path.join(param["directory"], param['out_file_name'])
    labels = [r[0] for r in rows]

    def bars(relative, path, xlabel, legend):
        fig, ax = plt.subplots(figsize=(12, max(1, int(len(global_stat) / 4))))
        for i, r in enumerate(rows):
            total, aligned, not_aligned, qfail = int(r[3]), int(r[4]), int(r[7]), int(r[8])
            if relative:
                tot = max(total, 1)
                a, n, q = aligned / tot * 100, not_aligned / tot * 100, qfail / tot * 100
                ax.barh(i, a, .75, color="#6290C3", hatch="\\", edgecolor="black", linewidth=.7)
                ax.barh(i, n, .75, left=a, color="#F1FFE7", hatch="//", edgecolor="black", linewidth=.7)
                ax.barh(i, q, .75, left=a + n, color="#FB5012", hatch="||", edgecolor="black", linewidth=.7)
            else:
                ax.barh(i, total, .75, color="#FFD25A", hatch="//", edgecolor="black", linewidth=.7)
                ax.barh(i, aligned, .75, color="#FFAA5A", hatch="\\", edgecolor="black", linewidth=.7)
                ax.barh(i, not_aligned, .75, color="#F56416", hatch="x", edgecolor="black", linewidth=.7)
        ax.set_yticks(np.arange(len(labels)))
        ax.set_yticklabels(labels)
        ax.tick_params(axis='both', which='both', labelsize=16)
        ax.set_xlabel(xlabel, size=20)
        ax.spines['top'].set_visible(False)
        ax.spines['right'].set_visible(False)
        ax.set_xlim(xmin=1)
        ax.legend(legend, loc='right', bbox_to_anchor=(1.1, 1), ncol=3, prop={'size': 12})
        fig.tight_layout()
        fig.savefig(path, dpi=300, bbox_inches='tight')
        plt.close(fig)

    bars(False, base + "_reads_plot.png", 'Number of reads',
         ["Total reads in sample", "Aligned reads", "Reads that passed quality filtering but failed to align"])
    bars(True, base + "_reads_plot_percentage.png", '% of reads per sample',
         ["Aligned reads", "Reads that passed quality filtering but failed to align", "Reads that did not pass quality filtering"])

    per_sample = [[compiled[f][k] for f in compiled] for k in range(len(head) - 1)]

    def violin(data, path, title):
        fig, ax = plt.subplots(figsize=(12, max(1.0, len(global_stat) / 2)))
        ax.set_title(title, size=20)
        ax.set_xlabel('Reads per feature', size=20)
        if data and all(len(d) for d in data):
            parts = ax.violinplot(data, points=200, widths=1, showmeans=False, showmedians=False, showextrema=False, vert=False)
            for pc in parts['bodies']:
                pc.set_facecolor('#D43F3A'); pc.set_edgecolor('black'); pc.set_alpha(1)
            q1, med, q3 = np.percentile(np.array(data, dtype=float), [25, 50, 75], axis=1)
            inds = np.arange(1, len(med) + 1)
            ax.scatter(med, inds, marker='o', color='white', s=40, zorder=3)
            ax.hlines(inds, q1, q3, color='k', linestyle='-', lw=8)
        ax.set_yticks(np.arange(len(head[1:])) + 1)
        ax.set_yticklabels(head[1:])
        ax.tick_params(axis='both', which='major', labelsize=20)
        ax.spines['top'].set_visible(False)
        ax.spines['right'].set_visible(False)
        ax.set_xlim(xmin=1)
        fig.savefig(path, dpi=300, bbox_inches='tight')
        plt.close(fig)

    violin(per_sample, base + "_distribution_plot.png", 'Reads per feature distribution')
    totals = [sum(d) for d in per_sample]                      # (once per sample: the inner sum made this step quadratic in the library size)
    rpm = [[v / t * 1000000 for v in d] for d, t in zip(per_sample, totals) if t > 0]
    violin(rpm if len(rpm) == len(per_sample) else [], base + "_distribution_normalized_RPM_plot.png",
           'Reads per feature (RPM normalized) distribution')


def main(argv=None):
    param = file_sizer_split(initializer(input_parser(argv)))
    features = {}
    if param['Running Mode'] == 'C':
        features = features_loader(param["feature"])
    aligner_mp_dispenser(features, param)
    sharding.barrier()
    if sharding.world().rank == 0:
        compiling(param)
    sharding.barrier()


if __name__ == "__main__":
    main()
