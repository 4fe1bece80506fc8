"""-m gpu: bench.py itself, small -- one process, and two ranks through torch.distributed.run on the one GPU of the
box (gloo, FL_BENCH_ONE_DEVICE=1: the N > 1 orchestration of the product -- shard, solve with libFL.so, gather -- with
everything but RCCL).  Strong scaling solves the same global problems however they are sharded, so the total number of
L-BFGS iterations must not depend on the number of ranks or on the assignment."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMON = ["--steps", "1", "--warmup", "1", "--batch", "2048", "--scaling", "strong", "--no-two-loop"]


def _run(cmd, env=None):
    e = dict(os.environ)
    e.update(env or {})
    p = subprocess.run(cmd, cwd=ROOT, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    line = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")][-1]
    return json.loads(line)


def test_bench_line_one_gpu_and_two_ranks_on_one_gpu():
    one = _run([sys.executable, "bench.py", "--gpus", "1", "--cpu-sample", "64"] + COMMON)
    assert one["n_gpus"] == 1 and one["scaling"] == "strong" and one["converged_fraction"] == 1.0
    r = one["roofline"]
    assert r["bound"] == "hbm" and 0.0 < r["frac"] <= 1.0 and r["model_bytes_per_launch"] > 0
    assert r["algorithmic_bytes_per_launch"] > r["model_bytes_per_launch"]
    assert one["parity"]["ok"] and all(one["parity"]["bit_exact_vs_oracle_kernel_order"][k] for k in ("x", "f", "iterations"))
    assert one["cpu_baseline"]["kind"] == "port" and one["cpu_baseline"]["cores"] >= 1
    for extra in ([], ["--interleaved"]):
        two = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                    "127.0.0.1", "--master-port", "29611", "bench.py", "--gpus", "2", "--backend", "gloo", "--cpu-sample",
                    "0"] + COMMON + extra, env={"FL_BENCH_ONE_DEVICE": "1"})
        assert two["n_gpus"] == 2 and two["ranks"]["world_size"] == 2 and two["scaling"] == "strong"
        assert two["iterations_per_step"] == one["iterations_per_step"]
        assert sum(two["ranks"]["iterations_per_rank"]) == two["iterations_per_step"]
        assert two["ranks"]["gather_ms"] is not None and two["config"]["global_batch"] == 2048


def test_bench_refuses_n_gpus_without_a_launcher():
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "2"], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode != 0 and b"WORLD_SIZE" in p.stderr


def test_bench_through_rccl_with_a_world_of_one():
    """The N > 1 code path of bench.py -- process group on RCCL ("nccl"), device-resident gather buffers, the gather
    inside the timed step, the all_reduce / all_gather of the report -- executed on the one GPU of the box with a
    world of one (FL_BENCH_FORCE_DIST=1 under torch.distributed.run): everything but a second peer."""
    line = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
                 "127.0.0.1", "--master-port", "29613", "bench.py", "--gpus", "1", "--backend", "nccl", "--cpu-sample",
                 "0"] + COMMON, env={"FL_BENCH_FORCE_DIST": "1"})
    assert line["ranks"]["backend"] == "nccl" and line["ranks"]["world_size"] == 1
    assert line["ranks"]["gather_ms"] is not None and line["config"]["exchange"].startswith("gather")
    assert line["converged_fraction"] == 1.0 and line["iterations_per_step"] == sum(line["ranks"]["iterations_per_rank"])
