#!/usr/bin/env python3
"""bench.py -- Mreads/s of the 2FAST2Q counting hot path on MI355X.

A step = one whole job over one device-resident batch of synthetic 150-bp reads: zeroed
accumulators -> Phred mask -> window -> <= m-mismatch match -> counts, followed by the all-reduce of the int64
count vector when N > 1.

  N = 1   workload cfg3_50M_10k_m1 = BASELINE.json configs[2], the configuration the metric is quoted on
          (50 M x 150 bp reads vs 10 k x 20 bp guides, --m 1 --ph 30).
  N > 1   workload cfg4_400M_100k_m1 = configs[3]: 400 M reads x 100 k guides IN TOTAL, split N ways
          ("scaling": "strong"); one RCCL all-reduce of int64[100 005] ends every step.  `--scaling weak` keeps
          the per-GPU load of the chosen workload fixed instead.

`--gpus N` with N > 1 outside torchrun starts the N ranks itself (python -m torch.distributed.run, before this
process touches the GPU); under the driver's torchrun launch it only checks that WORLD_SIZE == N.

The JSON line also carries: `roofline` (bytes the dominant kernel must touch / HIP-event kernel time, measured
HBM traffic from two rocprofv3 --pmc passes of a child of this script), `verify` (the timed result checked
against the oracle and the counter identities), `cpu_baseline` / `cpu_baseline_1core` (the oracle on host cores,
same sample) and `end_to_end` (FASTQ text in host memory / a file -> counts, PCIe inclusive; never `value`).
"""
import argparse
import csv
import glob
import importlib
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: total reads (per GPU under weak scaling), guides, --m
    "cfg2_10M_1k_m0": dict(n_reads=10_000_000, n_guides=1000, miss=0, lib_seed=0xF2A5 + 2),
    "cfg3_50M_10k_m1": dict(n_reads=50_000_000, n_guides=10000, miss=1, lib_seed=0xF2A5 + 3),
    "cfg4_50M_100k_m1": dict(n_reads=50_000_000, n_guides=100000, miss=1, lib_seed=0xF2A5 + 4),
    "cfg4_400M_100k_m1": dict(n_reads=400_000_000, n_guides=100000, miss=1, lib_seed=0xF2A5 + 4),
    # config 3's reads as a two-window run: --st 0,10 --l 10 against the guides written as first10:last10
    "cfg3_2win_50M_10k_m1": dict(n_reads=50_000_000, n_guides=10000, miss=1, lib_seed=0xF2A5 + 3, windows=2),
    # config 5: up+guide+down cassette at a uniform offset in [0,100]; --us/--ds anchored search
    "cfg5a_50M_10k_anchor_m1": dict(n_reads=50_000_000, n_guides=10000, miss=1, lib_seed=0xF2A5 + 5, anchored=True),
    # two --us/--ds pairs (here: the same cassette twice, so every read gives the key guide:guide) against 10 k two-part features
    "cfg5c_2pair_50M_10k_m1": dict(n_reads=50_000_000, n_guides=10000, miss=1, lib_seed=0xF2A5 + 5, anchored=True, pairs=2),
    # Extract+Count with a fixed window on the config-3 reads
    "cfg3b_50M_fixed_ec": dict(n_reads=50_000_000, n_guides=10000, miss=1, lib_seed=0xF2A5 + 3, ec=True),
    "cfg5b_50M_anchor_ec": dict(n_reads=50_000_000, n_guides=10000, miss=1, lib_seed=0xF2A5 + 5, anchored=True, ec=True),
}
UP, DOWN = "GTTTAAGAGCTA", "CGTTACCAGGTT"
B_CONTRACT = 188          # 38 B of 2-bit bases + 150 quality bytes per 150-bp read
HBM_PEAK_GBS = 8000.0
HBM_STREAM_COPY_GBS = 6290.0        # what a streaming copy reaches on MI355X (MI355X_MICROARCH.md: ~6.3 TB/s achievable)
SEED = 0xBEEF
PMC_CHILD_STEPS = 4


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS))
    ap.add_argument("--scaling", default=None, choices=["weak", "strong"])
    ap.add_argument("--reads", type=int, default=0, help="override the workload's read count (debug)")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the oracle legs (cpu_baseline, verify.prefix)")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 --pmc passes (roofline.traffic = null)")
    ap.add_argument("--no-extras", action="store_true", help="skip end_to_end and the strong-scaling N = 1 leg")
    ap.add_argument("--force-dist", action="store_true", help="run the RCCL all-reduce path even with one rank (rehearsal)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: ranks may share one GPU, the reduction goes through host tensors (rehearsal on a 1-GPU box)")
    ap.add_argument("--miss", type=int, default=None, help="override --m (experiments)")
    ap.add_argument("--guides", type=int, default=0, help="override the library size (experiments)")
    ap.add_argument("--phred", type=int, default=30, help="override --ph (experiments)")
    ap.add_argument("--ms", type=int, default=1, help="--msu/--msd of the anchored workloads (experiments)")
    ap.add_argument("--read-len", type=int, default=150, help="override the read length (experiments)")
    ap.add_argument("--p-n", type=float, default=0.005, help="share of reads with an N in the window (experiments)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    return ap.parse_args(argv)


def resolve(a, world):
    """workload dict + reads this rank holds"""
    name = a.workload or ("cfg3_50M_10k_m1" if world == 1 else "cfg4_400M_100k_m1")
    scaling = a.scaling or ("weak" if world == 1 or a.workload else "strong")
    w = dict(WORKLOADS[name])
    if a.reads:
        w["n_reads"] = a.reads
    if a.miss is not None:
        w["miss"] = a.miss
    if a.guides:
        w["n_guides"] = a.guides
    return name, scaling, w


def make_job(pkg, w, a, device, n, first_read):
    """(context, resident block) of reads [first_read, first_read + n) of the workload's stream"""
    guides = pkg.binding.synth_library(w["lib_seed"], w["n_guides"], 20)
    if w.get("anchored") and w.get("pairs") == 2:
        feats = [g + ":" + g for g in guides]
        c = pkg.Counter(features=feats, miss=w["miss"], phred=a.phred, device=device, upstream=UP + "," + UP, downstream=DOWN + "," + DOWN,
                        miss_search_up=a.ms, miss_search_down=a.ms)
        spec = dict(seed=SEED, n_reads=n, first_read=first_read, read_len=a.read_len, p_n=a.p_n, cassette=True, up=UP,
                    down=DOWN, max_offset=100)
        blk = c.synth_create(guides=guides, **spec)
        return c, blk, feats, spec
    if w.get("anchored"):
        akw = dict(upstream=UP, downstream=DOWN, miss_search_up=a.ms, miss_search_down=a.ms)
        c = pkg.Counter(features=None if w.get("ec") else guides, mode="EC" if w.get("ec") else "C", miss=w["miss"],
                        phred=a.phred, device=device, **akw)
        spec = dict(seed=SEED, n_reads=n, first_read=first_read, read_len=a.read_len, p_n=a.p_n, cassette=True, up=UP,
                    down=DOWN, max_offset=100)
        blk = c.synth_create(guides=guides, **spec) if w.get("ec") else c.synth_create(**spec)
    elif w.get("windows") == 2:
        feats = [g[:10] + ":" + g[10:] for g in guides]
        c = pkg.Counter(features=feats, miss=w["miss"], phred=a.phred, length=10, start="0,10", device=device)
        spec = dict(seed=SEED, n_reads=n, first_read=first_read, read_len=a.read_len, p_n=a.p_n)
        blk = c.synth_create(guides=guides, **spec)
        return c, blk, feats, spec
    elif w.get("ec"):
        c = pkg.Counter(features=None, mode="EC", phred=a.phred, length=20, start="0", device=device)
        spec = dict(seed=SEED, n_reads=n, first_read=first_read, read_len=a.read_len, p_n=a.p_n)
        blk = c.synth_create(guides=guides, **spec)
    else:
        c = pkg.Counter(features=guides, miss=w["miss"], phred=a.phred, length=20, start="0", device=device)
        spec = dict(seed=SEED, n_reads=n, first_read=first_read, read_len=a.read_len, p_n=a.p_n)
        blk = c.synth_create(**spec)
    return c, blk, guides, spec


def required_bytes_per_read(w, read_len):
    """bytes of a read the dominant kernel cannot avoid fetching.  Fixed --st 0 --l 20: the 2 base words (8 B) and
    5 quality words (20 B) under the window + the 2-byte length = 30 B (the tile layout keeps the other rows
    untouched); anchored runs need every base and every quality byte: SURVEY §8(d)'s 188 B at 150 bp."""
    if w.get("anchored"):
        return (2 * read_len + 7) // 8 + read_len
    st, length = 0, 20
    nb = ((st + length - 1) >> 4) - (st >> 4) + 1
    nq = ((st + length - 1) >> 2) - (st >> 2) + 1
    return 4 * nb + 4 * nq + 2


# ---- rocprofv3 PMC passes (child of this script, started before this process touches the GPU) -------------------
def pmc_child(a):
    """what the profiler wraps: no torch, one context, one resident block, four launches"""
    pkg = importlib.import_module("2fast2q_amd")
    name, _, w = resolve(a, 1)
    c, blk, _, _ = make_job(pkg, w, a, 0, w["n_reads"], 0)
    for _ in range(PMC_CHILD_STEPS):
        c.reset()
        c.count_resident(blk)
    c.read_counts()
    blk.free(); c.close()


def pmc_traffic(a, dominant):
    """HBM bytes per launch of the dominant kernel: FETCH_SIZE and WRITE_SIZE (KiB) from two separate rocprofv3 --pmc
    passes; gfx950 counts a 128-B request of a wide coalesced stream as 64 B in FETCH_SIZE, so the read side is doubled
    (MI355X_MICROARCH.md, HBM section).  Returns (bytes or None, detail dict)."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, {"error": "rocprofv3 not found"}
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None, {"error": "already running under a profiler"}
    out = {}
    tmp = tempfile.mkdtemp(prefix="f2q_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    child = [sys.executable, os.path.abspath(__file__), "--pmc-child", "--workload", resolve(a, 1)[0], "--phred", str(a.phred),
             "--ms", str(a.ms), "--read-len", str(a.read_len), "--p-n", str(a.p_n)]
    if a.reads:
        child += ["--reads", str(a.reads)]
    if a.miss is not None:
        child += ["--miss", str(a.miss)]
    if a.guides:
        child += ["--guides", str(a.guides)]
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, counter)
            cmd = [exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "--"] + child
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=240)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None, {"error": f"rocprofv3 --pmc {counter} failed (exit {r.returncode})", "stderr": r.stderr.decode(errors="replace")[-300:]}
            vals = []
            for row in csv.DictReader(open(files[0])):
                if row["Counter_Name"] == counter and any(k in row["Kernel_Name"] for k in dominant.split("|")):
                    vals.append(float(row["Counter_Value"]))
            if not vals:
                return None, {"error": f"no {dominant} dispatch in the {counter} pass"}
            out[counter + "_KiB"] = sum(vals) / PMC_CHILD_STEPS          # per step: a step may launch the kernel several times
            out["dispatches"] = len(vals)
            out["steps"] = PMC_CHILD_STEPS
    except (subprocess.TimeoutExpired, OSError, KeyError, ValueError) as e:
        return None, {"error": f"{type(e).__name__}: {e}"}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    traffic = (2.0 * out["FETCH_SIZE_KiB"] + out["WRITE_SIZE_KiB"]) * 1024.0
    out["formula"] = "(2*FETCH_SIZE + WRITE_SIZE) KiB, separate --pmc passes, all launches of " + dominant + " in a step"
    return traffic, out


# ---- oracle legs -----------------------------------------------------------------------------------------------
def oracle_kwargs(w, a):
    kw = dict(miss=w["miss"], phred=a.phred)
    if w.get("anchored"):
        rep = w.get("pairs", 1)
        kw.update(upstream=",".join([UP] * rep), downstream=",".join([DOWN] * rep), miss_search_up=a.ms, miss_search_down=a.ms)
        if w.get("ec"):
            kw["mode"] = "EC"
    elif w.get("windows") == 2:
        kw.update(length=10, start="0,10")
    else:
        kw.update(length=20, start="0")
        if w.get("ec"):
            kw["mode"] = "EC"
    return kw


def cpu_legs(pkg, c, w, a, guides, spec):
    """The oracle (C port of the reference algorithm) on this host's cores over a bounded sample = the first S reads
    of the bench stream as FASTQ text, and the device result for the same S reads checked against it."""
    from oracle import oracle as O
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    threads = max(1, min(avail, 16))              # the GPU box's CPU share for one GPU is 16 cores
    S = 4_000_000 if w["n_guides"] <= 10000 else 1_000_000
    S = min(S, spec["n_reads"])
    fq = bytes(c.synth_fastq(**dict(spec, n_reads=S, first_read=0)))
    feats = None if w.get("ec") else [(str(i), s) for i, s in enumerate(guides)]
    kw = oracle_kwargs(w, a)
    t0 = time.perf_counter()
    orc = O.count_fastq_parallel(fq, threads, features=feats, **kw) if not w.get("ec") else None
    if orc is None:
        orc = O.Oracle(features=None, **kw)
        orc.count_fastq(fq)
        threads = 1
    dt = time.perf_counter() - t0
    base = {"value": S / dt / 1e6, "unit": "Mreads/s", "cores": threads, "kind": "port", "host_cpus": os.cpu_count(),
            "sample": f"first {S} reads of the bench stream as FASTQ text in memory, {threads} threads, oracle/f2q_oracle.c "
                      f"(dict hit + early-exit all-vs-all with memo caches)", "seconds": dt}
    S1 = max(1, S // 16)
    fq1 = bytes(c.synth_fastq(**dict(spec, n_reads=S1, first_read=0)))
    t0 = time.perf_counter()
    o1 = O.Oracle(features=feats, **kw)
    o1.count_fastq(fq1)
    dt1 = time.perf_counter() - t0
    one = {"value": S1 / dt1 / 1e6, "unit": "Mreads/s", "cores": 1, "kind": "port", "sample": f"first {S1} reads, one thread", "seconds": dt1}
    # the device on the same S reads (resident block generated by the device twin of the generator)
    c.reset()
    pre = c.synth_create(**dict(spec, n_reads=S, first_read=0))
    c.count_resident(pre)
    counts, stats = c.read_counts()
    pre.free()
    ok = list(stats) == orc.stats()
    if w.get("ec"):
        ok = ok and [(k, n) for k, n, _ in c.ec_results()] == list(zip(orc.keys(), orc.counts()))
    else:
        ok = ok and list(counts) == orc.counts()
    prefix = {"reads": S, "stats": [int(v) for v in stats], "equals_oracle": bool(ok)}
    return base, one, prefix, fq


def end_to_end(pkg, c, fq, n_fq):
    """PCIe-inclusive rates of the host entry points on the same FASTQ text (never `value`)"""
    out = {"sample_reads": n_fq, "unit": "Mreads/s"}
    c.reset(); c.count_block(fq[: 1 << 20]); c.reset()
    for key in ("host_text_first", "host_text_to_counts"):             # second pass: the device buffers of that size exist
        c.reset(); t0 = time.perf_counter(); c.count_block(fq); c.read_counts(); dt = time.perf_counter() - t0
        out[key] = n_fq / dt / 1e6
    # the same text already resident in HBM -> framed, packed and counted by the device: what producing the tile layout
    # costs next to the counting kernel, without PCIe in the way (at most 1 GiB of text per f2q_text)
    cut = fq[: min(len(fq), (1 << 30) - (1 << 20))]
    if len(cut) < len(fq):
        cut = cut[: cut.rfind(b"\n@r") + 1]
    n_cut = cut.count(b"\n") // 4
    txt = c.text_upload(cut)
    try:
        c.reset(); c.count_text(txt)
        c.reset(); t0 = time.perf_counter(); _, tt = c.count_text(txt, want_timing=True); _, st = c.read_counts(); dt = time.perf_counter() - t0
        out["device_text_to_counts"] = n_cut / dt / 1e6
        out["device_text"] = {"reads": int(st[0]), "text_bytes": len(cut), "wall_ms": dt * 1e3, "stream_ms": tt["total_ms"],
                              "counting_kernels_ms": tt["kernel_ms"], "text_GBps": len(cut) / dt / 1e9,
                              "note": "FASTQ text resident in HBM -> k_nl_count .. k_pack -> counting kernels (f2q_count_text); no "
                                      "host-to-device copy inside the timed call, three host waits for sizes"}
    finally:
        txt.free()
    d = tempfile.mkdtemp(prefix="f2q_bench_")
    try:
        p = os.path.join(d, "x.fastq")
        with open(p, "wb") as f:
            f.write(fq)
        for key in ("plain_file_first", "plain_file_to_counts"):       # second pass: pinned staging buffers already exist
            c.reset(); t0 = time.perf_counter(); c.count_file(p); c.read_counts(); dt = time.perf_counter() - t0
            out[key] = n_fq / dt / 1e6
        import gzip
        pz = os.path.join(d, "x.fastq.gz")
        cut = fq[: len(fq) // 2]                                       # (half of the sample: compressing it here takes ~6 s)
        cut = cut[: cut.rfind(b"\n@r") + 1]
        nz = cut.count(b"\n") // 4
        with gzip.open(pz, "wb", compresslevel=1) as f:               # ONE deflate stream, as `gzip -1` writes it
            f.write(cut)
        for key in ("gzip_file_first", "gzip_file_to_counts"):         # second pass: the decoder's buffers exist
            c.reset(); t0 = time.perf_counter(); c.count_file(pz); _, st = c.read_counts(); dt = time.perf_counter() - t0
            out[key] = nz / dt / 1e6
        out["gzip_sample_reads"] = int(st[0])
        out["gzip_note"] = "ordinary single-member .gz; the worker pool (F2Q_IO_THREADS, default min(cores, 16)) decodes chunks of the one deflate stream"
    finally:
        shutil.rmtree(d, ignore_errors=True)
    c.reset()
    return out


def timed_steps(c, blk, steps, warmup, allreduce=None, barrier=None, sync=None):
    path = 0
    for _ in range(warmup):
        c.reset(); path = c.count_resident(blk).get("path", 0)
        if allreduce:
            allreduce()
    if barrier:
        barrier()
    if sync:
        sync()
    c.queued_times()                               # (forget steps queued earlier)
    t0 = time.perf_counter()
    for _ in range(steps):
        c.reset()                                  # each step is a whole job: zeroed accumulators -> count -> all-reduce
        c.count_resident_queued(blk)               # queued on the context's stream, no host wait between steps; every step
        if allreduce:                              # is stamped with its own pair of HIP events on that stream
            allreduce()
    if sync:
        sync()
    if barrier:
        barrier()
    dt = time.perf_counter() - t0
    kern = c.queued_times()                        # kernel time of each of the K steps, read after the timed region
    assert len(kern) == steps, (len(kern), steps)
    return dt, sum(kern) / len(kern), path


def main(argv=None, engine=None):
    """engine: None = the product (2fast2q_amd on a HIP device).  tests/test_bench_gloo.py passes a stand-in with the same
    methods so that the N > 1 plumbing (slicing of the stream, the all-reduce per step, max-over-ranks timing, the checks,
    rank 0's line) runs in the CPU test-suite; nothing in this file knows what the stand-in is."""
    a = parse_args(argv)
    if a.pmc_child:
        return pmc_child(a)
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and a.gpus > 1:
        # not under a launcher: start the ranks ourselves, before anything here touches the GPU
        port = os.environ.get("MASTER_PORT", "29511")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr",
               "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + (sys.argv[1:] if argv is None else list(argv))
        raise SystemExit(subprocess.call(cmd))
    # stdout carries the ONE JSON line and nothing else: libraries that print there (RCCL writes a version banner at its
    # first collective) are sent to stderr for the rest of the run
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    world = int(env_world or "1")
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but the launcher started WORLD_SIZE={world} ranks")
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    name, scaling, w = resolve(a, world)
    dominant = "k_extract_fixed4" if w.get("ec") and not w.get("anchored") else "k_count_anchor_pairs" if w.get("pairs") else "k_extract_anchor" if w.get("anchored") and w.get("ec") else "k_count_anchor" if w.get("anchored") else "k_count_multi4|k_count_fixed4_lds" if w.get("windows") else "k_count_fixed4|k_part_"
    # (a fixed window is counted by k_count_fixed4[_lds] or, large libraries, by the k_part_scatter / k_part_count / k_part_reduce
    # sequence: the PMC passes sum whichever family the step launched; `roofline.kernel` names the one that ran, from f2q_timing.path)

    traffic, traffic_detail = None, {"error": "skipped"}
    if world == 1 and not a.no_pmc:
        traffic, traffic_detail = pmc_traffic(a, dominant)      # child processes; this one has not touched the GPU yet

    import torch
    import torch.distributed as dist
    device = 0
    if engine is None:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a HIP device: the counting path has no CPU fallback")
        ndev = torch.cuda.device_count()
        device = local if a.dist_backend == "nccl" else local % ndev
        torch.cuda.set_device(device)
    use_dist = world > 1 or a.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if a.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = importlib.import_module("2fast2q_amd") if engine is None else engine
    total = w["n_reads"] * (world if scaling == "weak" else 1)        # reads of the whole job
    lo, hi = total * rank // world, total * (rank + 1) // world        # this rank's slice of the stream
    n = hi - lo
    c, blk, guides, spec = make_job(pkg, w, a, device, n, lo)
    info = blk.info()

    allreduce = barrier = None
    stream = None
    if use_dist:
        ptr, n64 = c.counts_device_ptr()
        if a.dist_backend == "nccl":
            class _Arr:                            # the library's device accumulator as a tensor: reduced in place by RCCL
                __cuda_array_interface__ = {"shape": (n64,), "typestr": "<i8", "data": (ptr, False), "version": 3}
            acc = torch.as_tensor(_Arr(), device=torch.device("cuda", device))
            stream = torch.cuda.ExternalStream(c.stream(), device=torch.device("cuda", device))

            def allreduce():
                with torch.cuda.stream(stream):    # the stream the counting kernels were launched on
                    dist.all_reduce(acc)
        else:
            holder = {}

            def allreduce():
                counts, stats = c.read_counts()
                t = torch.tensor(list(counts) + list(stats), dtype=torch.int64)
                dist.all_reduce(t)
                holder["t"] = t
        barrier = dist.barrier

    def sync():
        if stream is not None:
            stream.synchronize()
        if engine is None:
            torch.cuda.synchronize()

    dt, k_ms, path = timed_steps(c, blk, a.steps, a.warmup, allreduce, barrier, sync)
    if "|" in dominant:
        dominant = {4: "k_part_scatter + k_part_count + k_part_reduce", 3: "k_count_fixed4_lds", 2: "k_count_fixed4", 5: "k_count_multi4",
                    10: "k_count_fixed4_lds<.., MW> (windows back to back on the tiles)"}.get(path, dominant.split("|")[0])
    if use_dist:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda" if a.dist_backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # ---- the timed result is checked: whole-job counters on every rank, oracle on a prefix at N = 1 ---------------
    if use_dist and a.dist_backend == "gloo":
        t = holder["t"]
        counts, stats = t[:-5].numpy(), t[-5:].numpy()
    else:
        counts, stats = c.read_counts()
    stats = [int(v) for v in stats]
    verify = {"reads": stats[0], "stats": stats, "whole_job_reads_expected": total,
              "identity_reads_eq_sum_of_outcomes": stats[0] == sum(stats[1:]),
              "identity_counts_eq_aligned": (w.get("ec") or int(counts.sum()) == stats[1] + stats[2])}
    assert stats[0] == total, (stats, total)
    assert verify["identity_reads_eq_sum_of_outcomes"] and verify["identity_counts_eq_aligned"], verify
    if use_dist:
        chk = torch.tensor(stats + [int(counts.sum())], dtype=torch.int64, device="cuda" if a.dist_backend == "nccl" else "cpu")
        mx = chk.clone(); dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        assert bool((mx == chk).all()), "ranks disagree on the all-reduced result"
        verify["all_ranks_hold_the_same_result"] = True

    if rank == 0:
        b_req = required_bytes_per_read(w, a.read_len)
        achieved = b_req * n / (k_ms * 1e-3) / 1e9
        contract = B_CONTRACT * a.read_len / 150.0 * n / (k_ms * 1e-3) / 1e9
        assert achieved <= HBM_PEAK_GBS, (achieved, "required bytes / kernel time exceeds the HBM peak: the byte model is wrong")
        out = {
            "metric": "Mreads/sec matched (150 bp, 20 bp feature, m=%d)" % w["miss"],
            "value": total * a.steps / dt / 1e6, "unit": "Mreads/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None, "dtype": "u64 (2-bit packed bases) / u8 qualities", "data": "synthetic",
            "config": {"workload": name, "reads_total": total, "reads_per_gpu": n, "read_len": a.read_len,
                       "guides": w["n_guides"], "guide_len": 20, "miss": w["miss"], "phred": a.phred, "start": 0,
                       "general_path_reads_per_gpu": info["n_general"], "sharding": f"dp{world}",
                       "collective": (f"{a.dist_backend} all_reduce(int64[{w['n_guides'] + 5}]) per step" if use_dist else None)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "frac_of_stream_copy": achieved / HBM_STREAM_COPY_GBS, "traffic": traffic,
                         "kernel": dominant, "kernel_ms": k_ms, "bytes_per_read": b_req,
                         "achieved_contract": contract, "frac_contract": contract / HBM_PEAK_GBS,
                         "bytes_per_read_contract": B_CONTRACT * a.read_len / 150.0,
                         "traffic_gbs": (traffic / (k_ms * 1e-3) / 1e9) if traffic else None,
                         "traffic_bytes_per_read": (traffic / n) if traffic else None,
                         "traffic_detail": traffic_detail,
                         "note": ("achieved = bytes the kernel must touch per read x reads / HIP-event time of the step's counting "
                                  "launches on the library's stream (dominant kernel + histogram reduction); achieved_contract uses "
                                  "SURVEY 8(d)'s 188 B/read, which a fixed-offset kernel does not need to fetch (the tile layout "
                                  "leaves the rows outside the window untouched) and may therefore exceed the HBM peak")},
            "verify": verify,
        }
        if not a.no_cpu_baseline:                   # rank 0 of any world size (the other ranks wait at the closing barrier)
            base, one, prefix, fq = cpu_legs(pkg, c, w, a, guides, spec)
            out["cpu_baseline"], out["cpu_baseline_1core"] = base, one
            out["verify"]["prefix_vs_oracle"] = prefix
            assert prefix["equals_oracle"], "device result differs from the oracle on the prefix sample"
            if world == 1 and not a.no_extras and not w.get("ec"):
                try:
                    out["end_to_end"] = end_to_end(pkg, c, fq, prefix["reads"])
                except Exception as e:              # an optional leg must not take the bench line down
                    out["end_to_end"] = {"error": f"{type(e).__name__}: {e}"}
            del fq
        if world == 1 and not a.no_extras and not a.workload and not a.reads:
            # the N = 1 leg of the strong-scaling workload the N > 1 runs use, so that the curve has its anchor
            blk.free(); c.close()
            try:
                w4 = dict(WORKLOADS["cfg4_400M_100k_m1"])
                c4, b4, _, _ = make_job(pkg, w4, a, device, w4["n_reads"], 0)
                dt4, k4, _ = timed_steps(c4, b4, 3, 1, None, None, sync)
                _, s4 = c4.read_counts()
                out["strong_scaling_n1"] = {"workload": "cfg4_400M_100k_m1", "n_gpus": 1, "steps": 3, "value": w4["n_reads"] * 3 / dt4 / 1e6,
                                            "unit": "Mreads/s", "ms_per_step": dt4 / 3 * 1e3, "kernel_ms": k4,
                                            "reads_check": int(s4[0]) == w4["n_reads"] and int(s4[0]) == int(sum(s4[1:]))}
                b4.free(); c4.close()
            except Exception as e:
                out["strong_scaling_n1"] = {"error": f"{type(e).__name__}: {e}"}
            blk = c = None
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if blk is not None:
        blk.free(); c.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
