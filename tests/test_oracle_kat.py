"""The reference's own known-answer tests (reference tests/test_mainfunctions.py:4-78),
restated against the oracle -- this is what pins oracle/f2q_oracle.c to upstream."""
import json
import os

from conftest import GOLDEN
from oracle import oracle as O


def test_seq2bin_equivalent():
    # reference test_seq2bin (:4-8): seq2bin is the ASCII byte view; the oracle works on bytes directly
    assert list(b"GATTACA") == [71, 65, 84, 84, 65, 67, 65]


def test_border_finder():
    # reference test_borderFinder (:10-16)
    assert O.border_finder(b"GATTACA", b"TACTGATTACAGCAC", 1) == 4


def test_sequence_tinder():
    # reference test_sequenceTinder (:18-51).  quality_set_up/down = set("") <=> --qsu/--qsd 1
    read, qual = b"TACTGATTACAGCAC", b"AAII$%&#III/(&/"
    assert O.sequence_tinder(read, qual, "TACT", "GCAC", 1, 1, qual_up=1, qual_down=1) == (4, 11)
    # quality_set_down = {"/"}: the reference test injects a one-character set, which no --qsd value
    # produces (sets are prefixes of '!'..'~').  --qsd 16 fails '!'..'/' and, like {"/"}, rejects
    # the downstream anchor's qualities "/(&/" -> (None, None)
    assert O.sequence_tinder(read, qual, "TACT", "GCAC", 1, 1, qual_up=1, qual_down=16) == (None, None)
    assert O.sequence_tinder(read, qual, "TACT", "GCAC", 1, 2, qual_up=1, qual_down=1) == (4, 6)


def test_sequence_tinder_multi_search():
    # reference test_sequenceTinderMultiSearch (:53-78)
    read = b"AAAAAACACACACACACACACATTCAGGGGGGCCAAAAATAGAGAGAGAGAGACCGAGAGGGGGTTAGCATCG"
    qual = b"B" * 90
    got = []
    for i in range(2):
        s, e = O.sequence_tinder(read, qual, "CACACATT,GAGACCGA", "TAGAGAGA,TAGCATCG", 0, 0, i=i)
        got.append(read[s:e])
    assert got == [b"CAGGGGGGCCAAAAA", b"GAGGGGGT"]


def test_border_finder_reference_run_vectors():
    # 300 random (seq, read, mismatch, start_place) -> answers captured from the reference function
    with open(os.path.join(GOLDEN, "search_kat.json")) as f:
        kat = json.load(f)
    for s, r, m, sp, want in kat["border_finder"]:
        assert O.border_finder(s.encode(), r.encode(), m, sp) == want, (s, r, m, sp)
