"""oracle/oracle.py -- ctypes wrapper of the CPU restatement (TEST INFRASTRUCTURE).

Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import this module.  It is the checker for the HIP path, never a fallback.

The functions mirror the reference's names for the path (fast2q/fast2q.py):
``border_finder`` (:628), ``sequence_tinder`` (:215), and a ``count_fastq``
that plays the role of ``fastq_parser`` (:306) over an in-memory FASTQ buffer.
"""
import ctypes as C
import os
import subprocess
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libf2q_oracle.so")
MAX_ITER = 16
STAT_NAMES = ("reads", "perfect_counter", "imperfect_counter", "non_aligned_counter", "quality_failed")


class _Params(C.Structure):
    _fields_ = [
        ("mode", C.c_int32), ("miss", C.c_int32), ("phred", C.c_int32),
        ("qual_up", C.c_int32), ("qual_down", C.c_int32), ("length", C.c_int32),
        ("fixed", C.c_int32), ("n_iter", C.c_int32), ("starts", C.c_int32 * MAX_ITER),
        ("n_up", C.c_int32), ("n_down", C.c_int32), ("msu", C.c_int32), ("msd", C.c_int32),
        ("up", C.c_char_p * MAX_ITER), ("down", C.c_char_p * MAX_ITER),
    ]


def build(force=False):
    """Compile the C restatement (gcc). Building the checker is not using it."""
    src = os.path.join(_HERE, "f2q_oracle.c")
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libf2q_oracle.so"])
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB)
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.POINTER(_Params), C.c_int]
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_add_feature.restype = C.c_int64
        L.orc_add_feature.argtypes = [C.c_void_p, C.c_char_p, C.c_uint32]
        L.orc_border_finder.restype = C.c_long
        L.orc_border_finder.argtypes = [C.c_char_p, C.c_long, C.c_char_p, C.c_long, C.c_int, C.c_long]
        L.orc_sequence_tinder.restype = C.c_int
        L.orc_sequence_tinder.argtypes = [C.c_void_p, C.c_char_p, C.c_long, C.c_char_p, C.c_long, C.c_int,
                                          C.POINTER(C.c_long), C.POINTER(C.c_long)]
        L.orc_count_fastq.restype = C.c_int64
        L.orc_count_fastq.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        L.orc_get_stats.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
        L.orc_n_keys.restype = C.c_int64
        L.orc_n_keys.argtypes = [C.c_void_p]
        L.orc_key_bytes.restype = C.c_int64
        L.orc_key_bytes.argtypes = [C.c_void_p]
        L.orc_get_counts.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
        L.orc_get_keys.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_int64)]
        L.orc_reset.argtypes = [C.c_void_p]
        L.orc_merge_counts.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_skip_lines.restype = C.c_int64
        L.orc_skip_lines.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64]
        L.orc_count_lines.restype = C.c_int64
        L.orc_count_lines.argtypes = [C.c_void_p, C.c_int64]
        _lib = L
    return _lib


def _csv(x):
    if x is None:
        return []
    if isinstance(x, (list, tuple)):
        return [s for s in x]
    return str(x).split(",")


class Oracle:
    """One counting context.  Keyword names follow the reference's ``param`` dict
    (fast2q.py:1246-1309): miss, phred, length, start ("0" or "0,5"), upstream,
    downstream (comma separated or None), miss_search_up/down, qual_up/down,
    mode ("C" / "EC")."""

    def __init__(self, features=None, mode="C", miss=1, phred=30, length=20, start="0",
                 upstream=None, downstream=None, miss_search_up=0, miss_search_down=0,
                 qual_up=30, qual_down=30, use_memo=True):
        L = lib()
        p = _Params()
        p.mode = 0 if mode == "C" else 1
        p.miss, p.phred, p.length = int(miss), int(phred), int(length)
        p.qual_up, p.qual_down = int(qual_up), int(qual_down)
        p.msu, p.msd = int(miss_search_up), int(miss_search_down)
        ups, downs = _csv(upstream), _csv(downstream)
        self._keep = []
        if not ups and not downs:
            p.fixed = 1
            starts = [int(n) for n in str(start).split(",")]      # fast2q.py:539
            p.n_iter = len(starts)
            for i, s in enumerate(starts):
                p.starts[i] = s
        else:
            p.fixed = 0
            if ups and downs and len(ups) != len(downs):          # fast2q.py:553-556
                raise ValueError("Up and Downstream sequences must be submitted in concurrent pairs")
            p.n_up, p.n_down = len(ups), len(downs)
            for i, s in enumerate(ups):
                b = s.encode(); self._keep.append(b); p.up[i] = b
            for i, s in enumerate(downs):
                b = s.encode(); self._keep.append(b); p.down[i] = b
        self.mode = mode
        self._h = C.c_void_p(L.orc_create(C.byref(p), 1 if use_memo else 0))
        self.names = []
        if features is not None:
            # features: ordered iterable of (name, SEQ) or dict SEQ->name; sequences as the
            # loader leaves them (upper-cased, blanks removed, first duplicate wins :153-165)
            items = features.items() if isinstance(features, dict) else [(s, n) for (n, s) in features]
            for seq, name in items:
                b = seq.encode() if isinstance(seq, str) else bytes(seq)
                if L.orc_add_feature(self._h, b, len(b)) >= 0:
                    self.names.append(name)

    def close(self):
        if self._h:
            lib().orc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def count_fastq(self, data):
        """Feed a FASTQ byte buffer (accumulating). Returns bytes consumed."""
        buf = (C.c_char * len(data)).from_buffer_copy(data) if not isinstance(data, C.Array) else data
        return lib().orc_count_fastq(self._h, buf, len(data))

    def stats(self):
        out = (C.c_int64 * 5)()
        lib().orc_get_stats(self._h, out)
        return list(out)

    def stats_dict(self):
        return dict(zip(STAT_NAMES, self.stats()))

    def counts(self):
        n = lib().orc_n_keys(self._h)
        out = (C.c_int64 * n)()
        lib().orc_get_counts(self._h, out)
        return list(out)

    def keys(self):
        L = lib()
        n, nb = L.orc_n_keys(self._h), L.orc_key_bytes(self._h)
        raw = C.create_string_buffer(max(nb, 1))
        offs = (C.c_int64 * (n + 1))()
        L.orc_get_keys(self._h, raw, offs)
        return [raw.raw[offs[i]:offs[i + 1]].decode("latin-1") for i in range(n)]

    def result_dict(self):
        """{key: count}: Counter mode keys are feature sequences, EC mode the de-novo strings."""
        return dict(zip(self.keys(), self.counts()))

    def reset(self):
        lib().orc_reset(self._h)

    def merge(self, other):
        lib().orc_merge_counts(self._h, other._h)


def border_finder(seq, read, mismatch, start_place=0):
    """fast2q.py:628 -- returns the index or None."""
    s = seq if isinstance(seq, bytes) else bytes(seq)
    r = read if isinstance(read, bytes) else bytes(read)
    p = lib().orc_border_finder(s, len(s), r, len(r), int(mismatch), int(start_place))
    return None if p < 0 else int(p)


def sequence_tinder(read, qual, upstream=None, downstream=None, miss_search_up=0, miss_search_down=0,
                    qual_up=1, qual_down=1, length=20, i=0):
    """fast2q.py:215 -- returns (start, end) or (None, None). qual_up/qual_down are
    the --qsu/--qsd integers (1 => empty fail set)."""
    o = Oracle(mode="EC", upstream=upstream, downstream=downstream, miss_search_up=miss_search_up,
               miss_search_down=miss_search_down, qual_up=qual_up, qual_down=qual_down, length=length)
    a, b = C.c_long(), C.c_long()
    ok = lib().orc_sequence_tinder(o._h, read, len(read), qual, len(qual), i, C.byref(a), C.byref(b))
    o.close()
    return (int(a.value), int(b.value)) if ok else (None, None)


def split_fastq_on_records(data, parts):
    """Cut a FASTQ buffer into <=parts pieces on 4-line record boundaries (what the
    reference's chunking intends, fast2q.py:447-483, without its framing bug)."""
    L = lib()
    n = len(data)
    buf = (C.c_char * n).from_buffer_copy(data)
    lines = L.orc_count_lines(buf, n)
    recs = lines // 4
    per = max(1, -(-recs // max(1, parts)))
    cuts, pos = [0], 0
    while pos < n and len(cuts) < parts:
        pos = L.orc_skip_lines(buf, n, pos, per * 4)
        if pos < n:
            cuts.append(pos)
    cuts.append(n)
    return [data[cuts[i]:cuts[i + 1]] for i in range(len(cuts) - 1) if cuts[i + 1] > cuts[i]]


def count_fastq_parallel(data, threads, **kw):
    """CPU baseline driver: one Oracle per host thread over record-aligned pieces,
    results summed (Counter mode).  ctypes releases the GIL during the C call."""
    pieces = split_fastq_on_records(data, threads)
    workers = [Oracle(**kw) for _ in pieces]
    ts = [threading.Thread(target=w.count_fastq, args=(p,)) for w, p in zip(workers, pieces)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for w in workers[1:]:
        workers[0].merge(w)
    return workers[0]
