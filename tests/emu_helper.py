"""tests/emu_helper.py -- builds and drives tests/emu/libf2q_emu.so: the product's per-lane device
logic compiled for the host (g++), used by the CPU test-suite to check that logic against the
oracle and the golden vectors without a GPU.  Test infrastructure only."""
import ctypes as C
import importlib
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "emu", "f2q_emu.cpp")
LIB = os.path.join(HERE, "emu", "libf2q_emu.so")
CSRC = os.path.join(os.path.dirname(HERE), "2fast2q_amd", "csrc")
binding = importlib.import_module("2fast2q_amd.binding")


def build():
    deps = [SRC] + [os.path.join(CSRC, f) for f in ("f2q_device.h", "f2q_host.h", "f2q_synth.h", "f2q_reader.h", "f2q_inflate.h", "f2q_pargz.h")]
    if not os.path.exists(LIB) or any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function", "-Wno-unknown-pragmas",
                               "-o", LIB, SRC, "-lz", "-lpthread"])
    return LIB


_L = None


def lib():
    global _L
    if _L is None:
        L = C.CDLL(build())
        vp = C.c_void_p
        L.emu_create.restype = vp
        L.emu_create.argtypes = [C.POINTER(binding.Params)]
        L.emu_destroy.argtypes = [vp]
        L.emu_set_features.argtypes = [vp, C.c_char_p, C.POINTER(C.c_uint32), C.c_uint32]
        L.emu_count_block.restype = C.c_size_t
        L.emu_count_block.argtypes = [vp, C.c_char_p, C.c_size_t]
        L.emu_read_counts.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_uint64),
                                      C.POINTER(C.c_uint64)]
        L.emu_set_read_base.argtypes = [vp, C.c_uint64]
        L.emu_reset.argtypes = [vp]
        L.emu_use_v2.argtypes = [vp, C.c_int]
        L.emu_v2_reads.restype = C.c_uint64
        L.emu_v2_reads.argtypes = [vp]
        L.emu_use_lt.argtypes = [vp, C.c_int]
        L.emu_use_pw.argtypes = [vp, C.c_int]
        L.emu_pw_ok.argtypes = [vp]
        L.emu_lt_reads.restype = C.c_uint64
        L.emu_lt_reads.argtypes = [vp]
        L.emu_lt_ok.argtypes = [vp]
        L.emu_pt_force.argtypes = [vp, C.c_int]
        L.emu_pt_reads.restype = C.c_uint64
        L.emu_pt_reads.argtypes = [vp]
        L.emu_pt_parts.argtypes = [vp]
        L.emu_anchor_reads.restype = C.c_uint64
        L.emu_anchor_reads.argtypes = [vp]
        L.emu_ec_n.restype = C.c_uint64
        L.emu_ec_n.argtypes = [vp]
        L.emu_ec_overflow.restype = C.c_uint64
        L.emu_ec_overflow.argtypes = [vp]
        L.emu_ec_get.argtypes = [vp, C.c_uint64, C.c_char_p, C.POINTER(C.c_uint32), C.POINTER(C.c_int64),
                                 C.POINTER(C.c_uint64)]
        L.emu_synth_fastq.restype = C.c_size_t
        L.emu_synth_fastq.argtypes = [vp, C.POINTER(binding.Synth), C.c_uint64, C.c_uint64, C.c_char_p]
        L.emu_crc32.restype = C.c_uint32
        L.emu_crc32.argtypes = [C.c_uint32, C.c_char_p, C.c_size_t]
        L.emu_read_file.restype = C.c_longlong
        L.emu_read_file.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_char_p, C.c_size_t, C.POINTER(C.c_int),
                                    C.POINTER(C.c_int)]
        _L = L
    return _L


def read_file(path, piece=1 << 16, threads=0, out_cap=1 << 26):
    """Decode `path` with the product's file reader (f2q_reader.h) -> (bytes, truncated, kind)."""
    out = C.create_string_buffer(out_cap)
    tr, kind = C.c_int(0), C.c_int(0)
    n = lib().emu_read_file(os.fsencode(path), piece, threads, out, out_cap, C.byref(tr), C.byref(kind))
    if n < 0:
        raise RuntimeError(f"emu_read_file rc {n}")
    return out.raw[:n], bool(tr.value), {1: "plain", 2: "gzip", 3: "bgzf"}.get(kind.value, "none")


class Emu:
    def __init__(self, features=None, v2=True, lt=True, pt_parts=0, pw=True, **params):
        self._p, self._keep = binding.make_params(**params)
        self._h = C.c_void_p(lib().emu_create(C.byref(self._p)))
        if not self._h:
            raise ValueError("emu_create failed")
        lib().emu_use_v2(self._h, 1 if v2 else 0)
        lib().emu_use_lt(self._h, 1 if lt else 0)
        lib().emu_use_pw(self._h, 1 if pw else 0)      # pair tables (two-pair runs, A:B libraries); False: the string index
        if pt_parts:
            lib().emu_pt_force(self._h, pt_parts)      # partitioned tables whatever the library's size (F2Q_PT_PARTS)
        self.n = 0
        if features is not None:
            enc = [s.encode("latin-1") for s in features]
            offs = np.zeros(len(enc) + 1, dtype=np.uint32)
            if enc:
                offs[1:] = np.cumsum([len(b) for b in enc])
            lib().emu_set_features(self._h, b"".join(enc), offs.ctypes.data_as(C.POINTER(C.c_uint32)), len(enc))
            self.n = len(enc)

    def count_block(self, data):
        return lib().emu_count_block(self._h, data, len(data))

    def reset(self):
        """f2q_reset_counts (Counter mode): counts and the five counters back to zero"""
        lib().emu_reset(self._h)

    def v2_reads(self):
        return lib().emu_v2_reads(self._h)

    def lt_reads(self):
        """reads decided by the LDS-table logic (k_count_fixed4_lds)"""
        return lib().emu_lt_reads(self._h)

    def pt_reads(self):
        """reads decided by the partitioned-table logic (k_part_scatter / k_part_count)"""
        return lib().emu_pt_reads(self._h)

    def pt_parts(self):
        """partitions the library was dealt into (0: no partitioned tables)"""
        return lib().emu_pt_parts(self._h)

    def lt_ok(self):
        """the cuckoo build of the LDS tables succeeded for the library"""
        return bool(lib().emu_lt_ok(self._h))

    def pw_ok(self):
        """pair tables were built (a pure A:B library) and are in use"""
        return bool(lib().emu_pw_ok(self._h))

    def anchor_reads(self):
        return lib().emu_anchor_reads(self._h)

    def read(self):
        counts = (C.c_int64 * max(self.n, 1))()
        stats = (C.c_int64 * 5)()
        fast, gen = C.c_uint64(), C.c_uint64()
        lib().emu_read_counts(self._h, counts, stats, C.byref(fast), C.byref(gen))
        return list(counts)[:self.n], list(stats), fast.value, gen.value

    def ec_rows(self):
        L = lib()
        assert L.emu_ec_overflow(self._h) == 0
        rows = []
        for e in range(L.emu_ec_n(self._h)):
            key = C.create_string_buffer(4096)
            ln, cnt, first = C.c_uint32(), C.c_int64(), C.c_uint64()
            L.emu_ec_get(self._h, e, key, C.byref(ln), C.byref(cnt), C.byref(first))
            rows.append((key.raw[:ln.value].decode("latin-1"), cnt.value, first.value))
        rows.sort(key=lambda r: r[2])
        return rows

    # -- the subset of binding.Counter's interface that sharding.py drives --
    @property
    def mode(self):
        return "C" if self._p.mode == 0 else "EC"

    def set_read_base(self, b):
        lib().emu_set_read_base(self._h, int(b))

    def read_counts(self):
        counts, stats, _, _ = self.read()
        return np.array(counts, dtype=np.int64), np.array(stats, dtype=np.int64)

    def ec_results(self):
        return self.ec_rows()

    def synth_fastq(self, lo, hi, **spec):
        s, keep = binding.make_synth(**spec)
        buf = C.create_string_buffer((hi - lo) * (2 * s.read_len + 40) + 64)
        n = lib().emu_synth_fastq(self._h, C.byref(s), lo, hi, buf)
        return buf.raw[:n]

    def close(self):
        if self._h:
            lib().emu_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
