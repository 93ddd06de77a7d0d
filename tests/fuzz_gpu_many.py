"""one-off (GPU box): many fuzz seeds through the C ABI against the oracle -- usage: python tests/fuzz_gpu_many.py LO HI"""
import importlib, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))   # R = repo root (this file sits in tests/)
from fuzz_cases import make_case
from oracle import oracle as O
P = importlib.import_module("2fast2q_amd")
lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for seed in range(lo, hi):
    kw, feats, fq = make_case(seed)
    o = O.Oracle(features=[(str(i), s) for i, s in enumerate(feats)] if feats is not None else None, **kw)
    uo = o.count_fastq(fq)
    with P.Counter(features=feats, **kw) as c:
        u = c.count_block(fq)
        counts, stats = c.read_counts()
        ok = u == uo and list(stats) == o.stats() and (list(counts) == o.counts() if feats is not None else
                                                        [(k, n) for k, n, _ in c.ec_results()] == list(zip(o.keys(), o.counts())))
    if not ok:
        bad += 1
        print("MISMATCH", seed, kw, list(stats), o.stats())
print("seeds", lo, hi, "bad", bad)
