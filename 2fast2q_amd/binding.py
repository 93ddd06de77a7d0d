"""ctypes binding of libf2q_hip.so (include/f2q.h).

This is the only way the Python harness reaches the GPU: there is no Python or
CPU implementation of the counting path in this package.  If the shared library
is missing or no HIP device is usable, construction raises ``F2QError``.
"""
import ctypes as C
import os

import numpy as np

MAX_ITER = 16
_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("F2Q_LIB_PATH") or os.path.join(_HERE, "lib", "libf2q_hip.so")   # override: diagnostic builds
STAT_NAMES = ("reads", "perfect_counter", "imperfect_counter", "non_aligned_counter", "quality_failed")

EXPORTS = (
    "f2q_version", "f2q_build_id", "f2q_create", "f2q_destroy", "f2q_last_error", "f2q_set_features", "f2q_count_block",
    "f2q_count_file", "f2q_count_file_shard", "f2q_file_pieces", "f2q_census_pieces", "f2q_count_pieces", "f2q_synth_create", "f2q_block_from_fastq", "f2q_count_resident", "f2q_count_resident_queued", "f2q_queued_times", "f2q_block_info",
    "f2q_block_free", "f2q_synth_fastq", "f2q_synth_library", "f2q_reset_counts", "f2q_read_counts",
    "f2q_counts_device_ptr", "f2q_stream", "f2q_ec_size", "f2q_ec_fetch", "f2q_set_read_base", "f2q_synth_guides",
    "f2q_text_upload", "f2q_count_text", "f2q_text_free",
)

ERRORS = {-1: "EINVAL", -2: "ENODEVICE", -3: "EHIP", -4: "ENOMEM", -5: "EIO", -6: "ETRUNCATED", -7: "ESTATE",
          -8: "EUNSUPPORTED"}
F2Q_ETRUNCATED = -6


class F2QError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libf2q_hip: {ERRORS.get(code, code)}: {msg}")
        self.code = code


class Params(C.Structure):
    _fields_ = [
        ("mode", C.c_int32), ("miss", C.c_int32), ("phred", C.c_int32), ("length", C.c_int32),
        ("n_start", C.c_int32), ("start", C.c_int32 * MAX_ITER),
        ("n_upstream", C.c_int32), ("n_downstream", C.c_int32),
        ("upstream", C.c_char_p * MAX_ITER), ("downstream", C.c_char_p * MAX_ITER),
        ("miss_search_up", C.c_int32), ("miss_search_down", C.c_int32),
        ("qual_up", C.c_int32), ("qual_down", C.c_int32), ("device", C.c_int32), ("reserved", C.c_int32 * 7),
    ]


class Synth(C.Structure):
    _fields_ = [
        ("seed", C.c_uint64), ("n_reads", C.c_uint64), ("first_read", C.c_uint64),
        ("read_len", C.c_int32), ("start", C.c_int32), ("cassette", C.c_int32), ("max_offset", C.c_int32),
        ("up", C.c_char_p), ("down", C.c_char_p),
        ("t_sub", C.c_uint32), ("t_rand", C.c_uint32), ("t_n", C.c_uint32), ("t_lowq", C.c_uint32),
        ("t_q29", C.c_uint32), ("t_q28", C.c_uint32), ("reserved", C.c_int32 * 4),
    ]


class Timing(C.Structure):
    _fields_ = [("kernel_ms", C.c_double), ("total_ms", C.c_double), ("reads", C.c_uint64),
                ("fast_reads", C.c_uint64), ("general_reads", C.c_uint64), ("launches", C.c_uint32),
                ("path", C.c_uint32)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ if k != "reserved"}


def make_params(mode="C", miss=1, phred=30, length=20, start="0", upstream=None, downstream=None,
                miss_search_up=0, miss_search_down=0, qual_up=30, qual_down=30, device=0):
    """Build an f2q_params from the reference's ``param`` keys (fast2q.py:1246-1309).
    Returns (struct, keepalive list)."""
    p = Params()
    keep = []
    p.mode = 0 if str(mode).upper() == "C" else 1
    p.miss, p.phred, p.length = int(miss), int(phred), int(length)
    p.miss_search_up, p.miss_search_down = int(miss_search_up), int(miss_search_down)
    p.qual_up, p.qual_down, p.device = int(qual_up), int(qual_down), int(device)

    def split(x):
        if x is None:
            return []
        return list(x) if isinstance(x, (list, tuple)) else str(x).split(",")

    ups, downs = split(upstream), split(downstream)
    if not ups and not downs:
        starts = [int(n) for n in str(start).split(",")]            # fast2q.py:539
        if len(starts) > MAX_ITER:
            raise ValueError("at most 16 start positions")
        p.n_start = len(starts)
        for i, s in enumerate(starts):
            p.start[i] = s
    else:
        if len(ups) > MAX_ITER or len(downs) > MAX_ITER:
            raise ValueError("at most 16 search sequences")
        p.n_upstream, p.n_downstream = len(ups), len(downs)
        for i, s in enumerate(ups):
            b = s.encode(); keep.append(b); p.upstream[i] = b
        for i, s in enumerate(downs):
            b = s.encode(); keep.append(b); p.downstream[i] = b
    return p, keep


def make_synth(seed=0xF2A5, n_reads=1000, first_read=0, read_len=150, start=0, cassette=False, up="", down="",
               max_offset=100, p_sub=0.10, p_rand=0.05, p_n=0.005, p_lowq=0.06, p_q29=0.01, p_q28=0.01):
    """f2q_synth from the keyword set of tests/synth.py Spec."""
    def p32(x):
        return min(int(round(x * 4294967296.0)), 0xFFFFFFFF)
    s = Synth()
    s.seed, s.n_reads, s.first_read = seed, n_reads, first_read
    s.read_len, s.start, s.cassette, s.max_offset = read_len, start, 1 if cassette else 0, max_offset
    keep = [up.encode(), down.encode()]
    s.up, s.down = keep[0], keep[1]
    s.t_sub, s.t_rand, s.t_n = p32(p_sub), p32(p_sub + p_rand), p32(p_n)
    s.t_lowq, s.t_q29, s.t_q28 = p32(p_lowq), p32(p_lowq + p_q29), p32(p_lowq + p_q29 + p_q28)
    return s, keep


_lib = None


def load(path=None):
    """dlopen libf2q_hip.so and declare the prototypes.  Raises F2QError when it is not built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or LIB_PATH
    if not os.path.exists(path):
        raise F2QError(-2, f"{path} not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    L = C.CDLL(path)
    vp, u64p, i64p = C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_int64)
    L.f2q_version.restype = C.c_int
    L.f2q_build_id.restype = C.c_char_p
    L.f2q_create.argtypes = [C.POINTER(Params), C.POINTER(vp)]
    L.f2q_destroy.argtypes = [vp]; L.f2q_destroy.restype = None
    L.f2q_last_error.argtypes = [vp]; L.f2q_last_error.restype = C.c_char_p
    L.f2q_set_features.argtypes = [vp, C.c_char_p, C.POINTER(C.c_uint32), C.c_uint32]
    L.f2q_count_block.argtypes = [vp, vp, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(Timing)]
    L.f2q_count_file.argtypes = [vp, C.c_char_p, C.POINTER(Timing)]
    L.f2q_text_upload.argtypes = [vp, vp, C.c_size_t, C.POINTER(vp)]
    L.f2q_count_text.argtypes = [vp, vp, C.POINTER(C.c_size_t), C.POINTER(Timing)]
    L.f2q_text_free.argtypes = [vp, vp]; L.f2q_text_free.restype = None
    L.f2q_count_file_shard.argtypes = [vp, C.c_char_p, C.c_uint32, C.c_uint32, C.POINTER(Timing)]
    L.f2q_file_pieces.argtypes = [C.c_char_p, C.c_uint64, u64p, C.POINTER(C.c_int)]
    L.f2q_census_pieces.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint64, u64p, C.c_uint64]
    L.f2q_count_pieces.argtypes = [vp, C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint64, u64p, C.c_uint64, C.POINTER(Timing)]
    L.f2q_synth_create.argtypes = [vp, C.POINTER(Synth), C.POINTER(vp)]
    L.f2q_block_from_fastq.argtypes = [vp, vp, C.c_size_t, C.POINTER(vp)]
    L.f2q_count_resident.argtypes = [vp, vp, C.POINTER(Timing)]
    L.f2q_count_resident_queued.argtypes = [vp, vp]
    L.f2q_queued_times.argtypes = [vp, C.POINTER(C.c_float), C.c_uint32, C.POINTER(C.c_uint32)]
    L.f2q_block_info.argtypes = [vp, u64p, u64p, u64p]
    L.f2q_block_free.argtypes = [vp, vp]; L.f2q_block_free.restype = None
    L.f2q_synth_fastq.argtypes = [vp, C.POINTER(Synth), C.c_uint64, C.c_uint64, vp, C.POINTER(C.c_size_t)]
    L.f2q_synth_library.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_char_p]
    L.f2q_synth_guides.argtypes = [vp, C.c_char_p, C.c_uint32, C.c_uint32]
    L.f2q_reset_counts.argtypes = [vp]
    L.f2q_set_read_base.argtypes = [vp, C.c_uint64]
    L.f2q_read_counts.argtypes = [vp, i64p, i64p]
    L.f2q_counts_device_ptr.argtypes = [vp, C.POINTER(vp), u64p]
    L.f2q_stream.argtypes = [vp]; L.f2q_stream.restype = vp
    L.f2q_ec_size.argtypes = [vp, u64p, u64p]
    L.f2q_ec_fetch.argtypes = [vp, C.c_char_p, u64p, i64p, u64p]
    if path == LIB_PATH:
        _lib = L
    return L


def build_id():
    """the source hash the loaded library was compiled from (include/f2q.h: f2q_build_id)"""
    return (load().f2q_build_id() or b"").decode()


def synth_library(seed, n, length):
    """n unique uniform ACGT strings (host helper of the library; == tests/synth.py make_library)."""
    L = load()
    buf = C.create_string_buffer(n * length)
    rc = L.f2q_synth_library(seed, n, length, buf)
    if rc:
        raise F2QError(rc, "f2q_synth_library")
    raw = buf.raw
    return [raw[i * length:(i + 1) * length].decode() for i in range(n)]


class DeviceText:
    """FASTQ text resident in device memory (f2q_text)"""
    def __init__(self, counter, h):
        self._c, self._h = counter, h

    def free(self):
        if self._h:
            self._c._L.f2q_text_free(self._c._h, self._h)
            self._h = None


class Block:
    """A device-resident block of reads."""

    def __init__(self, owner, handle):
        self._o, self._h = owner, handle

    def info(self):
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        self._o._L.f2q_block_info(self._h, C.byref(a), C.byref(b), C.byref(c))
        return {"n_reads": a.value, "n_general": b.value, "device_bytes": c.value}

    def free(self):
        if self._h:
            self._o._L.f2q_block_free(self._o._h, self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Counter:
    """One counting context (f2q_ctx) bound to one GPU.

    ``features``: ordered list of sequences as features_loader leaves them
    (upper-case, blanks removed, unique); row i of the count vector is feature i.
    """

    def __init__(self, features=None, lib_path=None, **params):
        self._L = load(lib_path)
        self._p, self._keep = make_params(**params)
        self.mode = "C" if self._p.mode == 0 else "EC"
        h = C.c_void_p()
        rc = self._L.f2q_create(C.byref(self._p), C.byref(h))
        if rc:
            raise F2QError(rc, (self._L.f2q_last_error(None) or b"").decode())
        self._h = h
        self.n_features = 0
        if features is not None:
            self.set_features(features)

    # -- helpers --
    def _check(self, rc):
        if rc:
            raise F2QError(rc, (self._L.f2q_last_error(self._h) or b"").decode())

    def close(self):
        if getattr(self, "_h", None):
            self._L.f2q_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- library --
    def set_features(self, seqs):
        enc = [s.encode("latin-1") if isinstance(s, str) else bytes(s) for s in seqs]
        offs = np.zeros(len(enc) + 1, dtype=np.uint32)
        if enc:
            offs[1:] = np.cumsum([len(b) for b in enc], dtype=np.uint64).astype(np.uint32)
        blob = b"".join(enc)
        self._check(self._L.f2q_set_features(self._h, blob, offs.ctypes.data_as(C.POINTER(C.c_uint32)), len(enc)))
        self.n_features = len(enc)

    # -- counting --
    def count_block(self, data, want_timing=False):
        """fastq_parser over a bytes-like FASTQ buffer; returns bytes consumed (and timing)."""
        mv = memoryview(data)
        n = mv.nbytes
        if isinstance(data, bytes):
            ptr = C.cast(C.c_char_p(data), C.c_void_p)          # the bytes object's own buffer, no copy
        elif isinstance(data, bytearray):
            buf = (C.c_char * n).from_buffer(data)
            ptr = C.cast(buf, C.c_void_p)
        else:
            arr = np.frombuffer(mv, dtype=np.uint8)
            ptr = C.c_void_p(arr.ctypes.data)
        used, t = C.c_size_t(0), Timing()
        self._check(self._L.f2q_count_block(self._h, ptr, n, C.byref(used), C.byref(t)))
        return (used.value, t.as_dict()) if want_timing else used.value

    def text_upload(self, data):
        """copy at most 1 GiB of FASTQ text into device memory once -> DeviceText (count_text takes it from there)"""
        data = bytes(data)
        h = C.c_void_p()
        self._check(self._L.f2q_text_upload(self._h, C.cast(C.c_char_p(data), C.c_void_p), len(data), C.byref(h)))
        return DeviceText(self, h)

    def count_text(self, text, want_timing=False):
        """frame, pack and count FASTQ text that already sits in device memory; returns bytes consumed (and timing)"""
        used, t = C.c_size_t(0), Timing()
        self._check(self._L.f2q_count_text(self._h, text._h, C.byref(used), C.byref(t)))
        return (used.value, t.as_dict()) if want_timing else used.value

    def count_file(self, path):
        """Counts a whole .fastq / .gz file.  Returns (timing dict, truncated flag)."""
        t = Timing()
        rc = self._L.f2q_count_file(self._h, os.fsencode(path), C.byref(t))
        if rc == F2Q_ETRUNCATED:
            return t.as_dict(), True
        self._check(rc)
        return t.as_dict(), False

    def count_file_shard(self, path, rank, world):
        """This rank's share of a file that `world` processes count together.  Returns (timing dict, truncated flag)."""
        t = Timing()
        rc = self._L.f2q_count_file_shard(self._h, os.fsencode(path), rank, world, C.byref(t))
        if rc == F2Q_ETRUNCATED:
            return t.as_dict(), True
        self._check(rc)
        return t.as_dict(), False

    # -- a plain or BGZF file shared out by pieces: nobody reads or inflates foreign bytes (include/f2q.h, f2q_file_pieces ...) --
    def file_pieces(self, path, piece_bytes):
        """(number of pieces, shardable)"""
        n, ok = C.c_uint64(), C.c_int()
        rc = self._L.f2q_file_pieces(os.fsencode(path), piece_bytes, C.byref(n), C.byref(ok))
        if rc:
            raise F2QError(rc, (self._L.f2q_last_error(None) or b"").decode())
        return n.value, bool(ok.value)

    def census_pieces(self, path, rank, world, piece_bytes, n_pieces):
        """uint64[2 * n_pieces]: newline count and ends-with-newline flag of this rank's pieces (zeros elsewhere)"""
        census = np.zeros(2 * n_pieces, dtype=np.uint64)
        rc = self._L.f2q_census_pieces(os.fsencode(path), rank, world, piece_bytes, census.ctypes.data_as(C.POINTER(C.c_uint64)), n_pieces)
        if rc:
            raise F2QError(rc, (self._L.f2q_last_error(None) or b"").decode())
        return census

    def count_pieces(self, path, rank, world, piece_bytes, census):
        """counts the records that start in this rank's pieces; census = the ranks' census vectors summed"""
        census = np.ascontiguousarray(census, dtype=np.uint64)
        t = Timing()
        self._check(self._L.f2q_count_pieces(self._h, os.fsencode(path), rank, world, piece_bytes,
                                             census.ctypes.data_as(C.POINTER(C.c_uint64)), len(census) // 2, C.byref(t)))
        return t.as_dict()

    def block_from_fastq(self, data):
        data = bytes(data)
        h = C.c_void_p()
        self._check(self._L.f2q_block_from_fastq(self._h, C.cast(C.c_char_p(data), C.c_void_p), len(data), C.byref(h)))
        return Block(self, h)

    def synth_guides(self, guides):
        """guide set for the synthetic generator (needed by Extract+Count contexts, which take no library)"""
        glen = len(guides[0])
        self._check(self._L.f2q_synth_guides(self._h, "".join(guides).encode(), len(guides), glen))

    def synth_create(self, guides=None, **spec):
        if guides is not None:
            self.synth_guides(guides)
        s, keep = make_synth(**spec)
        h = C.c_void_p()
        self._check(self._L.f2q_synth_create(self._h, C.byref(s), C.byref(h)))
        return Block(self, h)

    def synth_fastq(self, lo=0, hi=None, guides=None, **spec):
        if guides is not None:
            self.synth_guides(guides)
        s, keep = make_synth(**spec)
        hi = s.n_reads if hi is None else hi
        n = C.c_size_t(0)
        self._check(self._L.f2q_synth_fastq(self._h, C.byref(s), lo, hi, None, C.byref(n)))
        buf = np.empty(n.value, dtype=np.uint8)
        self._check(self._L.f2q_synth_fastq(self._h, C.byref(s), lo, hi, C.c_void_p(buf.ctypes.data), C.byref(n)))
        return buf[:n.value]

    def count_resident(self, block):
        t = Timing()
        self._check(self._L.f2q_count_resident(self._h, block._h, C.byref(t)))
        return t.as_dict()

    def count_resident_queued(self, block):
        """queue the hot path over a resident block on the context's stream without waiting (see queued_times)"""
        self._check(self._L.f2q_count_resident_queued(self._h, block._h))

    def queued_times(self, cap=4096):
        """wait for the stream; kernel time (ms) of every step queued since the last call"""
        buf = (C.c_float * cap)()
        n = C.c_uint32()
        self._check(self._L.f2q_queued_times(self._h, buf, cap, C.byref(n)))
        return [float(buf[i]) for i in range(min(n.value, cap))]

    def set_read_base(self, first_read_index):
        self._check(self._L.f2q_set_read_base(self._h, int(first_read_index)))

    # -- results --
    def reset(self):
        self._check(self._L.f2q_reset_counts(self._h))

    def read_counts(self):
        counts = np.zeros(max(self.n_features, 1), dtype=np.int64)
        stats = np.zeros(5, dtype=np.int64)
        self._check(self._L.f2q_read_counts(self._h, counts.ctypes.data_as(C.POINTER(C.c_int64)),
                                            stats.ctypes.data_as(C.POINTER(C.c_int64))))
        return counts[:self.n_features], stats

    def counts_device_ptr(self):
        p, n = C.c_void_p(), C.c_uint64()
        self._check(self._L.f2q_counts_device_ptr(self._h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def stream(self):
        return self._L.f2q_stream(self._h)

    def ec_results(self):
        """[(key, count, first_read)] in first-occurrence order (== the reference's dict order)."""
        nk, nb = C.c_uint64(), C.c_uint64()
        self._check(self._L.f2q_ec_size(self._h, C.byref(nk), C.byref(nb)))
        n = nk.value
        keys = C.create_string_buffer(max(nb.value, 1))
        offs = np.zeros(n + 1, dtype=np.uint64)
        counts = np.zeros(max(n, 1), dtype=np.int64)
        first = np.zeros(max(n, 1), dtype=np.uint64)
        self._check(self._L.f2q_ec_fetch(self._h, keys, offs.ctypes.data_as(C.POINTER(C.c_uint64)),
                                         counts.ctypes.data_as(C.POINTER(C.c_int64)),
                                         first.ctypes.data_as(C.POINTER(C.c_uint64))))
        raw = keys.raw
        rows = [(raw[int(offs[i]):int(offs[i + 1])].decode("latin-1"), int(counts[i]), int(first[i])) for i in range(n)]
        rows.sort(key=lambda r: r[2])
        return rows
