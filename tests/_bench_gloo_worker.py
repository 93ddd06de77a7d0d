"""worker of tests/test_bench_gloo.py: one rank of `bench.py --gpus N --dist-backend gloo` with a stand-in engine (the
product's lane logic on the host, tests/emu) in place of the HIP library, so that the N > 1 path of bench.py -- the
slicing of the stream, one all-reduce per step, max-over-ranks timing, the whole-job checks, rank 0's single JSON line --
runs on a machine without GPUs.  The counting engine under bench.py is test infrastructure here; bench.py itself is the
file under test."""
import importlib.util
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import numpy as np                      # noqa: E402
import synth                            # noqa: E402
from emu_helper import Emu              # noqa: E402


class Block:
    def __init__(self, fq, n):
        self.fq, self.n = fq, n

    def info(self):
        return {"n_reads": self.n, "n_general": 0}

    def free(self):
        self.fq = None


class Counter:
    """the methods bench.py uses of 2fast2q_amd.Counter, on the emulator"""
    def __init__(self, features=None, **kw):
        kw.pop("device", None)
        self._kw, self._features = kw, features
        self._emu = Emu(features=features, **kw)
        self._queued = []
        self.mode = kw.get("mode", "C")

    def synth_create(self, guides=None, seed=0, n_reads=0, first_read=0, read_len=150, p_n=0.005, **_):
        spec = synth.Spec(seed=seed, n_reads=first_read + n_reads, read_len=read_len, p_n=p_n)
        return Block(synth.make_fastq(spec, guides or self._features, lo=first_read), n_reads)

    def synth_fastq(self, seed=0, n_reads=0, first_read=0, read_len=150, p_n=0.005, **_):
        spec = synth.Spec(seed=seed, n_reads=first_read + n_reads, read_len=read_len, p_n=p_n)
        return np.frombuffer(synth.make_fastq(spec, self._features, lo=first_read), dtype=np.uint8)

    def reset(self):
        self._emu.reset()

    def count_resident(self, blk):
        self._emu.count_block(blk.fq)
        return {"kernel_ms": 1.0}

    def count_resident_queued(self, blk):
        self._emu.count_block(blk.fq)
        self._queued.append(1.0)

    def queued_times(self):
        q, self._queued = self._queued, []
        return q

    def read_counts(self):
        counts, stats, _, _ = self._emu.read()
        return np.array(counts, dtype=np.int64), np.array(stats, dtype=np.int64)

    def counts_device_ptr(self):
        return 0, self._emu.n + 5

    def close(self):
        pass


class binding:
    @staticmethod
    def synth_library(seed, n, length):
        return synth.make_library(n, length, seed)


if __name__ == "__main__":
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    bench.main(sys.argv[1:], engine=sys.modules[__name__])
