"""world_size-2 gloo test of the multi-GPU path (runs on CPU): shard -> solve -> gather to rank 0
(contiguous and interleaved assignment, ragged shards padded inside the gatherer, buffers reused).

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


def _worker(rank, world, port, B, n, ret, interleaved):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "fortran-library_amd", "FortranLibrary"))
    import oracle_lib as O
    import distributed as D  # FortranLibrary/distributed.py (imported alone: the package itself needs libFL.so)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    d, b = _problem(B, n)
    idx = D.shard_indices(B, rank, world, interleaved).numpy()
    assert len(idx) == D.shard_size(B, rank, world, interleaved)
    r = O.solve_batch(O.LBFGS, O.DIAGQUAD, np.zeros((len(idx), n)), d=d[idx], b=b[idx],
                      opts=O.defaults(precision=1e-6), nthreads=1)
    # ragged shards go in as they are: the gatherer pads to ceil(B / world) rows and trims on rank 0
    G = D.Gatherer(B, {"x": ((n,), torch.float64), "f": ((), torch.float64), "iters": ((), torch.int32),
                       "status": ((), torch.int32)}, torch.device("cpu"), dst=0, interleaved=interleaved)
    mine = {k: torch.from_numpy(np.ascontiguousarray(r[k])) for k in ("x", "f", "iters", "status")}
    for _ in range(2):  # the buffers are reused from step to step
        res = G.gather(mine)
    # the overlapped form bench.py's timed step uses: enqueue, (next solve would run here), finish, read
    assert G.gather(mine, overlap=True) is None
    assert G.gather(mine, overlap=True) is None  # a second exchange first waits for the one in flight
    G.finish()
    if rank == 0:
        late = G.assembled()
        assert all(torch.equal(late[k], res[k]) for k in ("x", "f", "iters", "status"))
    dist.barrier()
    if rank == 0:
        assert res["x"].shape == (B, n)
        ret["x"] = res["x"].numpy().copy()
        ret["f"] = res["f"].numpy().copy()
        ret["iters"] = res["iters"].numpy().copy()
        one = D.gather_results(mine, batch=B, interleaved=interleaved)  # the one-shot form gives the same
        assert torch.equal(one["x"], res["x"]) and torch.equal(one["iters"], res["iters"])
    else:
        assert res is None
        assert D.gather_results(mine, batch=B, interleaved=interleaved) is None
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


def test_shard_indices_partition_the_batch():
    sys.path.insert(0, os.path.join(ROOT, "fortran-library_amd", "FortranLibrary"))
    import distributed as D
    for B in (1, 7, 8, 1000):
        for w in (1, 2, 3, 8):
            for inter in (False, True):
                parts = [D.shard_indices(B, r, w, inter) for r in range(w)]
                assert sorted(torch.cat(parts).tolist()) == list(range(B))
                assert [len(p_) for p_ in parts] == [D.shard_size(B, r, w, inter) for r in range(w)]
                assert max(len(p_) for p_ in parts) <= -(-B // w)


import pytest


@pytest.mark.parametrize("interleaved", [False, True])
def test_two_rank_gloo_shard_solve_gather(interleaved):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    B, n, world = 7, 64, 2  # ragged: ranks own 4 and 3 problems
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(world, 29533 + int(interleaved), B, n, ret, interleaved), nprocs=world, join=True)
        d, b = _problem(B, n)
        ref = O.solve_batch(O.LBFGS, O.DIAGQUAD, np.zeros((B, n)), d=d, b=b, opts=O.defaults(precision=1e-6), nthreads=1)
        assert np.array_equal(ret["x"], ref["x"])
        assert np.array_equal(ret["f"], ref["f"])
        assert np.array_equal(ret["iters"], ref["iters"])
