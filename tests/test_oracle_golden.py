"""Oracle vs the golden vectors captured from the reference's reads_counter()
(tests/golden/make_golden.py).  Bit-exact: per-key counts in dict order + 5 stats."""
import gzip

import pytest

from conftest import case_fastq, load_cases, loader_view
from oracle import oracle as O

CASES = load_cases()


def oracle_for(case, **extra):
    p = case["params"]
    kw = dict(mode=p.get("Running Mode", "C"), miss=p.get("miss", 1), phred=p.get("phred", 30),
              length=p.get("length", 20), start=p.get("start", "0"), upstream=p.get("upstream"),
              downstream=p.get("downstream"), miss_search_up=p.get("miss_search_up", 0),
              miss_search_down=p.get("miss_search_down", 0), qual_up=p.get("qual_up", 30),
              qual_down=p.get("qual_down", 30))
    kw.update(extra)
    feats = loader_view(case["features"]) if case["features"] is not None else None
    return O.Oracle(features=feats, **kw)


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
@pytest.mark.parametrize("memo", [True, False], ids=["memo", "nomemo"])
def test_oracle_matches_reference(case, memo):
    o = oracle_for(case, use_memo=memo)
    o.count_fastq(case_fastq(case))
    exp = case["expected"]
    assert o.stats() == exp["stats"]
    keys, counts = o.keys(), o.counts()
    assert keys == [r[1] for r in exp["rows"]]
    assert counts == [r[2] for r in exp["rows"]]
    if case["features"] is not None:
        assert o.names == [r[0] for r in exp["rows"]]


def test_streaming_blocks_equal_whole():
    case = next(c for c in CASES if c["name"] == "synth_fixed_m1")
    data = case_fastq(case)
    whole = oracle_for(case)
    whole.count_fastq(data)
    parts = O.split_fastq_on_records(data, 7)
    assert b"".join(parts) == data and len(parts) > 1
    inc = oracle_for(case)
    for p in parts:
        assert inc.count_fastq(p) == len(p)
    assert inc.counts() == whole.counts() and inc.stats() == whole.stats()
    par = O.count_fastq_parallel(data, 3, features=loader_view(case["features"]), miss=1)
    assert par.counts() == whole.counts() and par.stats() == whole.stats()
