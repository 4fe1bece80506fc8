"""world_size-2 gloo test of the multi-GPU path (runs on CPU): shard -> solve -> gather to rank 0.

The per-rank "solve" is the CPU oracle standing in for the HIP kernel (this is a test of the sharding and
of the single gather, the only communication of the job); rank 0 must end up with exactly the results of
solving the whole batch in one process, in global problem order."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _problem(B, n):
    rng = np.random.default_rng(42)
    kappa = np.exp(rng.uniform(np.log(10), np.log(100), B))
    d = 1.0 + (kappa[:, None] - 1.0) * (np.arange(n) / (n - 1))[None, :]
    b = rng.uniform(-1, 1, (B, n))
    return d, b


def _worker(rank, world, port, B, n, ret):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "fortran-library_amd", "FortranLibrary"))
    import oracle_lib as O
    import distributed as D  # FortranLibrary/distributed.py (imported alone: the package itself needs libFL.so)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    d, b = _problem(B, n)
    lo, hi = D.shard_bounds(B, rank, world)
    r = O.solve_batch(O.LBFGS, O.DIAGQUAD, np.zeros((hi - lo, n)), d=d[lo:hi], b=b[lo:hi],
                      opts=O.defaults(precision=1e-6), nthreads=1)
    # pad the last shard so that every rank contributes the same shape (gather needs equal sizes)
    per = -(-B // world)
    def pad(a):
        t = torch.from_numpy(np.ascontiguousarray(a))
        if t.shape[0] < per:
            t = torch.cat([t, torch.zeros((per - t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype)])
        return t
    res = D.gather_results({"x": pad(r["x"]), "f": pad(r["f"]), "iters": pad(r["iters"]), "status": pad(r["status"])})
    dist.barrier()
    if rank == 0:
        ret["x"] = res["x"][:B].numpy()
        ret["f"] = res["f"][:B].numpy()
        ret["iters"] = res["iters"][:B].numpy()
    else:
        assert res is None
    dist.destroy_process_group()


def test_shard_bounds_cover_batch_exactly():
    sys.path.insert(0, os.path.join(ROOT, "fortran-library_amd", "FortranLibrary"))
    import distributed as D
    for B in (1, 7, 8, 65536, 65537):
        for w in (1, 2, 4, 8):
            cuts = [D.shard_bounds(B, r, w) for r in range(w)]
            assert cuts[0][0] == 0 and cuts[-1][1] == B
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in cuts) == -(-B // w)


def test_two_rank_gloo_shard_solve_gather():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    B, n, world = 7, 64, 2  # ragged: ranks own 4 and 3 problems
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(world, 29533, B, n, ret), nprocs=world, join=True)
        d, b = _problem(B, n)
        ref = O.solve_batch(O.LBFGS, O.DIAGQUAD, np.zeros((B, n)), d=d, b=b, opts=O.defaults(precision=1e-6), nthreads=1)
        assert np.array_equal(ret["x"], ref["x"])
        assert np.array_equal(ret["f"], ref["f"])
        assert np.array_equal(ret["iters"], ref["iters"])
