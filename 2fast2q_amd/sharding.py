"""One process per GPU: record-block sharding of a FASTQ file and the reduction of the results.

The counting path shards trivially -- Counter-mode counts are a sum over reads of a pure per-read
function -- so there is no data-path collective: rank r counts the record blocks
b with b % world == r on its own GPU, and ONE all-reduce of the int64[n_features + 5] vector
(counts then the 5 stats) finishes the sample.  Under torchrun on a GPU node the backend is "nccl"
(= RCCL over xGMI) and the reduction runs in place on the library's device accumulator; on a
machine without GPUs (the CPU test-suite) it is gloo on host tensors.  Extract+Count tables are
keyed by string, so they are gathered and merged by key instead (sum of counts, min of first-read).
This replaces the reference's chunk pool and dict merge (fast2q.py:411-512, :439-445, :487-495).
"""
import gzip
import os
from collections import namedtuple

World = namedtuple("World", "rank size backend")
BLOCK_BYTES = 64 << 20
_state = {"world": None}


def world():
    """(rank, size, backend); initialises torch.distributed on first use when WORLD_SIZE > 1"""
    if _state["world"] is not None:
        return _state["world"]
    size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    backend = None
    if size > 1:
        import torch
        import torch.distributed as dist
        if not dist.is_initialized():
            backend = os.environ.get("F2Q_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            if backend == "nccl":
                torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", rank)))
            dist.init_process_group(backend, rank=rank, world_size=size)
        else:
            backend = dist.get_backend()
    _state["world"] = World(rank, size, backend)
    return _state["world"]


def reset_world():
    _state["world"] = None


def barrier():
    w = world()
    if w.size > 1:
        import torch.distributed as dist
        dist.barrier()


def shard_of(block_index, w):
    return block_index % w.size


def iter_record_blocks(path, block_bytes=BLOCK_BYTES):
    """Yield (block_index, first_read_index, bytes, truncated) with every block cut on a 4-line record
    boundary (the framing of fastq_parser, fast2q.py:324-328).  gzip by extension, like :567."""
    opener = gzip.open if os.path.splitext(path)[1] == ".gz" else open
    carry, index, first_read, truncated, eof = b"", 0, 0, False, False
    with opener(path, "rb") as f:
        while not eof:
            try:
                chunk = f.read(block_bytes)
            except (EOFError, OSError, gzip.BadGzipFile):          # :405-407 -- keep what was readable
                chunk, truncated = b"", True
            eof = not chunk
            buf = carry + chunk
            if not buf:
                break
            if eof:
                cut = len(buf)                                      # the tail goes as it is: a partial
            else:                                                   # record at the very end is dropped by the framer
                k = buf.count(b"\n")
                p = buf.rfind(b"\n")
                for _ in range(k % 4):
                    p = buf.rfind(b"\n", 0, p)
                cut = p + 1
            block, carry = buf[:cut], buf[cut:]
            if block:
                n_lines = block.count(b"\n") + (0 if block.endswith(b"\n") else 1)
                yield index, first_read, block, truncated
                first_read += n_lines // 4
                index += 1
    if truncated and index == 0:
        yield 0, 0, b"", True


PIECE_BYTES = 256 << 20
MAX_PIECE_BYTES = (1 << 30) - (2 << 20)       # the library frames a piece + its look-ahead with 32-bit offsets (include/f2q.h)
F2Q_EUNSUPPORTED = -8


def _piece_bytes():
    """F2Q_PIECE_BYTES, kept inside what the library accepts"""
    return max(4096, min(int(os.environ.get("F2Q_PIECE_BYTES", PIECE_BYTES)), MAX_PIECE_BYTES))


def count_file_sharded(ctx, path, w, block_bytes=BLOCK_BYTES):
    """rank w.rank counts its blocks of `path` into ctx; returns the truncated-gzip flag"""
    if hasattr(ctx, "count_pieces"):
        # plain and BGZF files: every rank reads (and inflates) only its own pieces.  The 4-line framing is global, so the ranks first pool the
        # line counts of their pieces (one small all-reduce), then each frames its share on its own (include/f2q.h)
        piece = _piece_bytes()
        n_pieces, ok = ctx.file_pieces(path, piece)
        if ok and n_pieces >= 1:
            import numpy as np
            import torch
            import torch.distributed as dist
            # A rank whose share cannot be read (a BGZF member damaged past its header, a pread error) must not leave
            # the others waiting in the all-reduce: the census carries one more element, "this rank failed", and when
            # it comes back non-zero every rank takes the streaming path, which keeps what is readable and reports
            # `truncated` the way the reference does (fast2q.py:405-407)
            census_np, failed = np.zeros(2 * n_pieces, dtype=np.uint64), 0
            try:
                census_np = ctx.census_pieces(path, w.rank, w.size, piece, n_pieces)
            except Exception:
                failed = 1
            census = torch.from_numpy(np.concatenate([census_np.astype("int64"), np.array([failed], dtype="int64")]))
            if w.backend == "nccl":
                census = census.cuda()
            dist.all_reduce(census)
            if int(census[-1].item()) == 0:
                error = None
                try:
                    ctx.count_pieces(path, w.rank, w.size, piece, census[:-1].cpu().numpy().astype("uint64"))
                    failed = 0
                except Exception as e:              # F2Q_EUNSUPPORTED: a line longer than the look-ahead behind a piece
                    failed = 1
                    if getattr(e, "code", None) != F2Q_EUNSUPPORTED:
                        error = e                   # a device error, an out-of-memory, a sizing bug: not to be papered over
                flag = torch.tensor([failed, 0 if error is None else 1], dtype=torch.int64, device=census.device)
                dist.all_reduce(flag)               # (every rank gets here, whatever happened to it)
                if error is not None:
                    raise error
                if int(flag[1].item()):
                    raise RuntimeError("another rank failed while counting its pieces of " + str(path))
                if int(flag[0].item()) == 0:
                    return False
            ctx.reset()                            # somebody could not: all ranks take the streaming path below
    if hasattr(ctx, "count_file_shard"):           # the library streams, frames and deals the pieces itself
        return ctx.count_file_shard(path, w.rank, w.size)[1]
    truncated = False                              # contexts without it (the CPU test-suite's stand-in): Python framing
    for index, first_read, block, trunc in iter_record_blocks(path, block_bytes):
        truncated = truncated or trunc
        if block and shard_of(index, w) == w.rank:
            ctx.set_read_base(first_read)
            ctx.count_block(block)
    return truncated


def merge_ec_tables(tables):
    """[(key, count, first_read)] lists -> one list in first-occurrence order"""
    merged = {}
    for rows in tables:
        for key, n, first in rows:
            if key in merged:
                m = merged[key]
                m[0] += n
                m[1] = min(m[1], first)
            else:
                merged[key] = [n, first]
    out = [(k, v[0], v[1]) for k, v in merged.items()]
    out.sort(key=lambda r: r[2])
    return out


def reduce_results(ctx, w):
    """all ranks end with the whole-sample (counts, stats, ec_rows)"""
    import torch
    import torch.distributed as dist
    ec_rows = None
    if w.backend == "nccl" and hasattr(ctx, "counts_device_ptr"):
        ptr, n64 = ctx.counts_device_ptr()

        class _Arr:
            __cuda_array_interface__ = {"shape": (n64,), "typestr": "<i8", "data": (ptr, False), "version": 3}
        dev = torch.device("cuda", torch.cuda.current_device())
        acc = torch.as_tensor(_Arr(), device=dev)
        # RCCL, in place on the library's accumulator and on the library's own stream: ordered after the counting
        # kernels without any host synchronisation (f2q_counts_device_ptr does none)
        stream = torch.cuda.ExternalStream(ctx.stream(), device=dev)
        with torch.cuda.stream(stream):
            dist.all_reduce(acc)
        stream.synchronize()
        counts, stats = ctx.read_counts()
    else:
        counts, stats = ctx.read_counts()
        t = torch.tensor(list(counts) + list(stats), dtype=torch.int64)
        dist.all_reduce(t)
        counts, stats = t[:-5].numpy(), t[-5:].numpy()
    if ctx.mode == "EC":
        gathered = [None] * w.size
        dist.all_gather_object(gathered, ctx.ec_results())
        ec_rows = merge_ec_tables(gathered)
    return counts, stats, ec_rows
