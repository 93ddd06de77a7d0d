"""pytest configuration: the `gpu` marker and shared helpers.

`-m "not gpu"` runs everywhere (oracle vs golden vectors, host logic, C-ABI symbol
checks, gloo multi-process path).  `-m gpu` tests call the HIP path through the
C-ABI and are the parity tests proper.
"""
import importlib
import json
import os
import struct
import zlib
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
TESTS = os.path.dirname(os.path.abspath(__file__))
if TESTS not in sys.path:
    sys.path.insert(0, TESTS)
GOLDEN = os.path.join(TESTS, "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: a minute or more (full-size runs against the oracle); part of -m gpu, deselect with -m 'gpu and not slow'")


def load_cases():
    with open(os.path.join(GOLDEN, "reads_counter_cases.json")) as f:
        return json.load(f)["cases"]


def case_fastq(case):
    """bytes of the FASTQ input of a golden case (inline, or regenerated from tests/synth.py)"""
    import synth
    if case.get("synth"):
        s = case["synth"]
        guides = synth.make_library(s["n_guides"], s["glen"], s["lib_seed"])
        return synth.make_fastq(synth.Spec(**s["spec"]), guides)
    return case["fastq"].encode("latin-1")


def loader_view(features):
    """What features_loader (fast2q.py:148-166) keeps of [[name, seq-as-written]...]:
    upper-cased, blanks removed, first occurrence of a sequence wins."""
    out, seen = [], set()
    for name, seq in features:
        s = seq.upper().replace(" ", "")
        if s not in seen:
            seen.add(s)
            out.append((name, s))
    return out


def pkg():
    """the product package (its directory name starts with a digit, so importlib)"""
    return importlib.import_module("2fast2q_amd")


@pytest.fixture(scope="session")
def cases():
    return load_cases()


def sprinkle_symbols(fastq, seed, rate=0.08, symbols=b"NnRacgtY."):
    """Rewrite random bases of a FASTQ buffer with lower-case / N / IUPAC / junk symbols (sequence lines only)."""
    import random
    rng = random.Random(seed)
    lines = fastq.split(b"\n")
    for i in range(1, len(lines), 4):
        b = bytearray(lines[i])
        for j in range(len(b)):
            if rng.random() < rate:
                c = symbols[rng.randrange(len(symbols))]
                b[j] = ord(chr(b[j]).lower()) if c in b"acgt" and rng.random() < 0.5 else c
        lines[i] = bytes(b)
    return b"\n".join(lines)


def bgzf_bytes(data, block=0xFF00, level=6, eof_marker=True, extra_subfield=False):
    """BGZF as bgzip writes it: gzip members with a 'BC' extra subfield holding the member size - 1."""
    out = bytearray()
    pieces = [data[i:i + block] for i in range(0, len(data), block)]
    if eof_marker:
        pieces.append(b"")
    for p in pieces:
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        body = co.compress(p) + co.flush()
        extra = (b"XY" + struct.pack("<H", 3) + b"abc" if extra_subfield else b"") + b"BC" + struct.pack("<H", 2)
        xlen = len(extra) + 2
        bsize = 12 + xlen + len(body) + 8
        out += b"\x1f\x8b\x08\x04" + b"\0\0\0\0" + b"\x00\xff" + struct.pack("<H", xlen) + extra + struct.pack("<H", bsize - 1)
        out += body + struct.pack("<II", zlib.crc32(p) & 0xFFFFFFFF, len(p))
    return bytes(out)
