"""bench.py's N > 1 path on a machine without GPUs: two ranks over gloo, the counting engine replaced by the host
emulation of the product's lane logic (tests/_bench_gloo_worker.py).  What is checked is bench.py: every rank takes its
slice of the stream, each step ends with one all-reduce, rank 0 prints ONE JSON line whose whole-job numbers equal the
oracle's single pass, and the line carries the N > 1 fields (scaling, collective, cpu_baseline on rank 0)."""
import json
import os
import socket
import subprocess
import sys

import pytest

import synth
from conftest import ROOT, TESTS
from oracle import oracle as O


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("scaling", ["strong", "weak"])
def test_bench_two_ranks_gloo(tmp_path, scaling):
    port = free_port()
    args = ["--gpus", "2", "--steps", "2", "--warmup", "1", "--dist-backend", "gloo", "--workload", "cfg3_50M_10k_m1", "--reads", "20000",
            "--guides", "300", "--no-pmc", "--no-extras", "--scaling", scaling]
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PYTHONPATH=ROOT)
        procs.append(subprocess.Popen([sys.executable, os.path.join(TESTS, "_bench_gloo_worker.py")] + args, env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1].decode()[-800:] for o in outs]
    assert outs[1][0].strip() == b""                                  # only rank 0 prints
    lines = [ln for ln in outs[0][0].decode().splitlines() if ln.strip()]
    assert len(lines) == 1                                            # ONE JSON line and nothing else on stdout
    d = json.loads(lines[0])
    total = 20000 * (2 if scaling == "weak" else 1)
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == scaling and d["unit"] == "Mreads/s"
    assert d["config"]["reads_total"] == total and d["config"]["reads_per_gpu"] == total // 2
    assert "gloo all_reduce(int64[305])" in d["config"]["collective"]
    v = d["verify"]
    assert v["reads"] == total and v["identity_reads_eq_sum_of_outcomes"] and v["all_ranks_hold_the_same_result"]
    # the whole-job counters equal the oracle's single pass over the same stream
    guides = synth.make_library(300, 20, 0xF2A5 + 3)
    fq = synth.make_fastq(synth.Spec(seed=0xBEEF, n_reads=total, read_len=150), guides)
    orc = O.Oracle(features=[(str(i), g) for i, g in enumerate(guides)], miss=1, phred=30, length=20, start="0")
    orc.count_fastq(fq)
    assert v["stats"] == orc.stats()
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 0 and v["prefix_vs_oracle"]["equals_oracle"]
    assert abs(d["value"] - total * 2 / (d["ms_per_step"] * 2e-3) / 1e6) < 1e-6 * d["value"] + 1e-9
