"""N>1 path on CPU: world_size-2 and -3 gloo jobs run 2fast2q_amd/sharding.py (record-block sharding
+ one all-reduce of the int64 vector / gather-merge of Extract+Count tables).  Every rank must end
with the whole-sample result, equal to the oracle's single pass."""
import gzip
import importlib
import json
import os
import socket
import subprocess
import sys

import pytest

import synth
from conftest import ROOT, TESTS
from oracle import oracle as O

sharding = importlib.import_module("2fast2q_amd.sharding")


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run_world(tmp_path, world, cfg):
    cfg["out"] = str(tmp_path / "res")
    cfgp = tmp_path / "cfg.json"
    cfgp.write_text(json.dumps(cfg))
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), PYTHONPATH=ROOT)
        procs.append(subprocess.Popen([sys.executable, os.path.join(TESTS, "_gloo_worker.py"), str(cfgp)], env=env))
    for p in procs:
        assert p.wait(timeout=300) == 0
    return [json.load(open(f"{cfg['out']}.{r}")) for r in range(world)]


@pytest.mark.parametrize("world,gz", [(2, False), (2, True), (3, False)])
def test_counter_mode_sharded(tmp_path, world, gz):
    guides = synth.make_library(200, 20, 31)
    fq = synth.make_fastq(synth.Spec(seed=9, n_reads=3000, read_len=60), guides)
    path = tmp_path / ("s.fastq.gz" if gz else "s.fastq")
    (gzip.open(path, "wb") if gz else open(path, "wb")).write(fq)
    res = run_world(tmp_path, world, {"features": guides, "params": {"miss": 1}, "path": str(path), "block_bytes": 40000})
    orc = O.Oracle(features=[(str(i), g) for i, g in enumerate(guides)], miss=1)
    orc.count_fastq(fq)
    for r in res:
        assert r["stats"] == orc.stats() and r["counts"] == orc.counts()
        assert 0 < r["own_reads"] < 3000            # every rank did part of the work, none did all
    assert sum(r["own_reads"] for r in res) == 3000


@pytest.mark.parametrize("fault", ["census", "unsupported", "device"])
def test_pieces_protocol_with_a_rank_that_cannot(tmp_path, fault):
    """count_file_sharded's pieces protocol when ONE rank fails: a failed census (a BGZF member damaged past its header)
    or a line beyond the look-ahead send every rank down the streaming path together -- nobody waits in an all-reduce
    the failed rank never enters, and the result is the oracle's; any other error is raised on every rank instead of
    being hidden behind a recount"""
    guides = synth.make_library(200, 20, 31)
    fq = synth.make_fastq(synth.Spec(seed=9, n_reads=3000, read_len=60), guides)
    path = tmp_path / "s.fastq"
    path.write_bytes(fq)
    res = run_world(tmp_path, 2, {"features": guides, "params": {"miss": 1}, "path": str(path), "block_bytes": 40000,
                                  "pieces_fault": fault})
    if fault == "device":
        assert all("raised" in r for r in res) and "code -4" in res[1]["raised"] and "another rank failed" in res[0]["raised"]
        return
    orc = O.Oracle(features=[(str(i), g) for i, g in enumerate(guides)], miss=1)
    orc.count_fastq(fq)
    for r in res:
        assert r["stats"] == orc.stats() and r["counts"] == orc.counts()


def test_extract_count_sharded(tmp_path):
    guides = synth.make_library(100, 20, 32)
    up, down = "GTTTAAGAGCTA", "CGTTACCAGGTT"
    fq = synth.make_fastq(synth.Spec(seed=10, n_reads=2000, cassette=True, up=up, down=down), guides)
    path = tmp_path / "s.fastq"
    path.write_bytes(fq)
    kw = {"mode": "EC", "upstream": up, "downstream": down, "miss_search_up": 1, "miss_search_down": 1}
    res = run_world(tmp_path, 2, {"features": None, "params": kw, "path": str(path), "block_bytes": 100000})
    orc = O.Oracle(**kw)
    orc.count_fastq(fq)
    want = [[k, n] for k, n in zip(orc.keys(), orc.counts())]
    for r in res:
        assert r["stats"] == orc.stats()
        assert [[k, n] for k, n, _ in r["ec"]] == want          # dict order restored from first-read indices


def test_record_blocks_are_record_aligned(tmp_path):
    guides = synth.make_library(10, 20, 33)
    fq = synth.make_fastq(synth.Spec(seed=11, n_reads=500, read_len=37), guides) + b"@partial\nACGT\n"
    path = tmp_path / "s.fastq"
    path.write_bytes(fq)
    blocks = list(sharding.iter_record_blocks(str(path), block_bytes=1000))
    assert b"".join(b for _, _, b, _ in blocks) == fq
    reads = 0
    for idx, first, b, trunc in blocks[:-1]:
        assert b.count(b"\n") % 4 == 0 and b.startswith(b"@r") and first == reads and not trunc
        reads += b.count(b"\n") // 4
    assert reads + blocks[-1][2].count(b"\n") // 4 == 500


def test_truncated_gzip_keeps_partial_counts(tmp_path):
    guides = synth.make_library(10, 20, 34)
    fq = synth.make_fastq(synth.Spec(seed=12, n_reads=4000, read_len=50), guides)
    path = tmp_path / "s.fastq.gz"
    raw = gzip.compress(fq)
    path.write_bytes(raw[: len(raw) // 2])
    blocks = list(sharding.iter_record_blocks(str(path), block_bytes=1 << 16))
    assert any(t for _, _, _, t in blocks)
    got = b"".join(b for _, _, b, _ in blocks)
    assert fq.startswith(got) and 0 < len(got) < len(fq)


def test_merge_ec_tables():
    a = [("AAA", 2, 5), ("CCC", 1, 9)]
    b = [("CCC", 4, 3), ("GGG", 1, 7)]
    assert sharding.merge_ec_tables([a, b]) == [("CCC", 5, 3), ("AAA", 2, 5), ("GGG", 1, 7)]
