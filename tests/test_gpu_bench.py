"""-m gpu: bench.py as the driver calls it - the JSON contract, the RCCL path on one GPU (a process group
of one rank: init_process_group("nccl"), the all-reduce of the real model's flat gradient buffer, the MAX /
SUM reductions of the timing), and the self-launch of `--gpus N` without a launcher."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, timeout=300):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)


def test_bench_line_with_rccl_group_of_one(device):
    r = _run(["--gpus", "1", "--dist-single", "--steps", "3", "--warmup", "2", "--no-cpu-baseline"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(r.stdout.strip().splitlines()) == 1, r.stdout[:500]  # ONE line: RCCL's banner must not land on stdout
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["unit"] == "edge-layers/s" and line["n_gpus"] == 1 and line["steps"] == 3
    # what a scaling run needs to explain itself, and the training-loop numbers beside the replay
    assert line["per_rank"][0]["edges"] == line["config"]["edges_per_gpu"] and line["per_rank"][0]["allreduce_ms"] > 0
    assert line["exposed_collective_ms"] > 0 and line["eager_ms_per_step"] > 0 and line["lib_sha256_16"]
    fb = line["fresh_batch"]
    assert fb and "error" not in fb and fb["ms_per_step"] > 0 and len(set(fb["edges"])) > 1 and fb["padded_edges"] >= max(fb["edges"])
    assert fb["replay_equals_eager"], fb  # a batch the graph was not captured on, at the benchmark's own size: same bits
    assert 0 < line["mfma_roofline"]["self_connection"]["frac"] < 1  # (executed flops of the low-rank route)
    assert abs(line["model_edges_per_s"] * line["config"]["conv_layers"] - line["value"]) < 1e-6 * line["value"]
    assert "all-reduce (nccl)" in line["config"]["step"]
    assert line["radial_mlp_launches_per_step"] == line["config"]["conv_layers"]  # no prefetch across the end of a pass
    roof = line["roofline"]
    assert roof["bound"] == "hbm" and 0.05 < roof["frac"] < 1.0 and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9
    e, n = line["config"]["edges_per_gpu"], line["config"]["nodes_per_gpu"]
    assert abs(line["config"]["avg_num_neighbors"] - e / n) < 1e-9  # computed from the batch, not assumed


def test_bench_allreduce_inside_the_graph(device):
    """The RCCL all-reduce captured as a node of the HIP graph (group of one rank)."""
    r = _run(["--gpus", "1", "--dist-single", "--graph-allreduce", "1", "--steps", "3", "--warmup", "2", "--no-cpu-baseline", "--fresh", "0"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["allreduce_in_graph"] is True and line["config"]["launch"] == "hip_graph_replay"


def test_bench_gpus_n_without_launcher(device):
    have = torch.cuda.device_count()
    r = _run(["--gpus", str(have + 1), "--steps", "1", "--warmup", "0", "--no-cpu-baseline"], timeout=120)
    assert r.returncode != 0 and f"needs {have + 1} devices" in r.stderr and "Traceback" not in r.stderr
