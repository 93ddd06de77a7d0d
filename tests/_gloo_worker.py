"""worker of tests/test_sharding_gloo.py: one rank of a world_size-N gloo job.  The per-rank counting
engine is tests/emu (the product's lane logic on the host) because this machine has no GPU; the
sharding / reduction code under test is the product's 2fast2q_amd/sharding.py."""
import importlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


def main():
    cfg = json.load(open(sys.argv[1]))
    os.environ["F2Q_DIST_BACKEND"] = "gloo"
    sharding = importlib.import_module("2fast2q_amd.sharding")
    from emu_helper import Emu
    w = sharding.world()
    eng = Emu(features=cfg["features"], **cfg["params"])
    fault = cfg.get("pieces_fault")
    if fault:
        # the pieces protocol of the library (file_pieces / census_pieces / count_pieces) with one rank that cannot do its
        # share: the census of rank 1 fails ("census"), its counting meets a line beyond the look-ahead ("unsupported") or a
        # device error ("device")
        class Err(RuntimeError):
            def __init__(self, code):
                super().__init__(f"code {code}")
                self.code = code

        def census_pieces(path, rank, world, piece, n_pieces):
            import numpy as np
            if fault == "census" and rank == 1:
                raise Err(-5)
            return np.zeros(2 * n_pieces, dtype=np.uint64)

        def count_pieces(path, rank, world, piece, census):
            if rank == 1:
                raise Err(-8 if fault == "unsupported" else -4)
        eng.file_pieces = lambda path, piece: (3, True)
        eng.census_pieces = census_pieces
        eng.count_pieces = count_pieces
    try:
        trunc = sharding.count_file_sharded(eng, cfg["path"], w, block_bytes=cfg["block_bytes"])
    except Exception as e:
        json.dump({"raised": type(e).__name__ + ": " + str(e)}, open(f"{cfg['out']}.{w.rank}", "w"))
        sharding.barrier()
        return
    counts, stats, ec = sharding.reduce_results(eng, w)
    own = eng.read()[1][0]
    json.dump({"counts": [int(x) for x in counts], "stats": [int(x) for x in stats], "ec": ec, "own_reads": own,
               "truncated": trunc}, open(f"{cfg['out']}.{w.rank}", "w"))
    sharding.barrier()


if __name__ == "__main__":
    main()
