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
COMMON = ["--steps", "1", "--warmup", "1", "--batch", "2048", "--scaling", "strong", "--no-two-loop", "--bar-problems", "512"]


def _run(cmd, env=None):
    e = dict(os.environ)
    e.update(env or {})
    p = subprocess.run(cmd, cwd=ROOT, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    line = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")][-1]
    return json.loads(line)


def test_bench_line_one_gpu_and_two_ranks_on_one_gpu():
    one = _run([sys.executable, "bench.py", "--gpus", "1", "--cpu-sample", "64", "--configs", "none"] + COMMON)
    assert one["n_gpus"] == 1 and one["scaling"] == "strong" and one["converged_fraction"] == 1.0
    r = one["roofline"]
    assert r["bound"] == "hbm" and 0.0 < r["frac"] <= 1.0 and r["model_bytes_per_launch"] > 0
    assert r["algorithmic_bytes_per_launch"] > r["model_bytes_per_launch"]
    assert one["parity"]["ok"] and all(one["parity"]["bit_exact_vs_oracle_kernel_order"][k] for k in ("x", "f", "iterations"))
    bar = one["parity"]["minimiser_bar"]  # the north-star minimiser bar where it is defined: kappa <= 100
    assert one["parity"]["ok_x"] and bar["comparable_problems"] >= bar["comparable_problems_required"]
    assert bar["x_err_max_where_both_met_the_gradient_test"] <= 1e-8 and bar["f_rel_err_max"] <= 1e-10
    assert bar["gpu_to_exact_minimiser_err_max"] <= 3e-7
    assert one["cpu_baseline"]["kind"] == "port" and one["cpu_baseline"]["cores"] >= 1
    for extra in (["--contiguous", "--configs", "none"], ["--configs", "c5"]):  # (interleaved shards are the default)
        two = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                    "127.0.0.1", "--master-port", "29611", "bench.py", "--gpus", "2", "--backend", "gloo", "--cpu-sample",
                    "0"] + COMMON + extra, env={"FL_BENCH_ONE_DEVICE": "1"})
        assert two["n_gpus"] == 2 and two["ranks"]["world_size"] == 2 and two["scaling"] == "strong"
        assert two["iterations_per_step"] == one["iterations_per_step"]
        assert sum(two["ranks"]["iterations_per_rank"]) == two["iterations_per_step"]
        assert two["ranks"]["gather_ms"] is not None and two["config"]["global_batch"] == 2048
        # both legs in the one invocation: the other one is the weak leg (2048 problems per rank)
        w = two["other_leg"]
        assert w["scaling"] == "weak" and w["config"]["global_batch"] == 4096 and w["n_gpus"] == 2
        if "c5" in extra:  # BASELINE config 5 sharded over the two ranks: the same 8192 problems as on one rank
            c5 = two["configs"]["C5"]
            assert c5["n_gpus"] == 2 and c5["scaling"] == "strong" and sum(c5["iterations_per_rank"]) == c5["iterations"]
            c5_two_ranks = c5["iterations"]
    solo = _run([sys.executable, "bench.py", "--cpu-sample", "0", "--configs", "c5", "--config-cpu-seconds", "1"] + COMMON)
    c5 = solo["configs"]["C5"]
    assert c5["iterations"] == c5_two_ranks and c5["parity"]["ok"] and c5["roofline"]["bound"] == "valu_issue"
    assert c5["cpu_baseline"]["kind"] == "port" and c5["cpu_baseline"]["value"] > 0


def test_bench_spawns_its_own_launcher_for_n_gpus():
    """`python bench.py --gpus 2` without torch.distributed.run: bench.py starts the launcher as a child process before
    anything touches the GPU (here: both ranks on the one GPU of the box, gloo)"""
    two = _run([sys.executable, "bench.py", "--gpus", "2", "--backend", "gloo", "--cpu-sample", "0", "--configs", "none",
                "--single-leg"] + COMMON, env={"FL_BENCH_ONE_DEVICE": "1"})
    assert two["n_gpus"] == 2 and two["ranks"]["world_size"] == 2 and "other_leg" not in two


def test_bench_every_baseline_config_in_the_one_line():
    """the driver's command (fewer steps): the headline plus C2, C3, C4, C4_gemm, C5, each with roofline, cpu_baseline, parity"""
    line = _run([sys.executable, "bench.py", "--gpus", "1", "--steps", "2", "--warmup", "1", "--config-cpu-seconds", "1.5"])
    assert line["parity"]["ok"] and line["parity"]["ok_x"]
    cf = line["configs"]
    assert set(cf) == {"C2", "C3", "C4", "C4_gemm", "C5"}
    for name, c in cf.items():
        assert c["parity"]["ok"], (name, c["parity"])
        assert c["cpu_baseline"]["value"] > 0 and c["cpu_baseline"]["cores"] >= 1
        assert c["roofline"]["bound"] in ("hbm", "mfma", "valu_issue") and c["ms"] > 0
    assert cf["C4"]["roofline"]["bound"] == "hbm" and cf["C4_gemm"]["roofline"]["bound"] == "mfma"
    assert 0.3 < cf["C4_gemm"]["roofline"]["frac"] <= 1.0 and 0.2 < cf["C4"]["roofline"]["frac"] <= 1.0


def test_bench_through_rccl_with_a_world_of_one():
    """The N > 1 code path of bench.py -- process group on RCCL ("nccl"), device-resident gather buffers, the gather
    inside the timed step, the all_reduce / all_gather of the report -- executed on the one GPU of the box with a
    world of one (FL_BENCH_FORCE_DIST=1 under torch.distributed.run): everything but a second peer."""
    line = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
                 "127.0.0.1", "--master-port", "29613", "bench.py", "--gpus", "1", "--backend", "nccl", "--cpu-sample",
                 "0", "--configs", "c5"] + COMMON, env={"FL_BENCH_FORCE_DIST": "1"})
    assert line["configs"]["C5"]["scaling"] == "strong" and line["configs"]["C5"]["n_gpus"] == 1  # the sharded form of a config through RCCL
    assert line["ranks"]["backend"] == "nccl" and line["ranks"]["world_size"] == 1
    assert line["ranks"]["gather_ms"] is not None and line["config"]["exchange"].startswith("gather")
    assert line["converged_fraction"] == 1.0 and line["iterations_per_step"] == sum(line["ranks"]["iterations_per_rank"])
