"""`python3 bench.py --gpus N` typed plainly -- no torch.distributed.run in front -- starts its N ranks by itself, as fresh
child processes, before the parent has imported torch or made a HIP call (bench.py::launch_ranks).  The GPU test runs the
exact command of the driver's multi-GPU tier on a one-GPU box (three ranks sharing device 0 over gloo: the row partition of
src/matrix/csr-matrix.cpp:77-95, the peer stores, the link probe and the gather check are the real ones; only the xGMI
hop is missing); the CPU test checks the launch itself where no GPU exists."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


def test_plain_gpus_n_starts_its_own_ranks_before_touching_a_device():
    """Without a GPU every rank must stop at 'no GPU visible' (there is no CPU fallback) -- which proves that N ranks were
    started and that each went through the launcher's environment.  The parent leaves with the children's failure."""
    import torch
    if torch.cuda.device_count() > 0:
        pytest.skip("a GPU is visible: test_gpu_three_ranks_on_one_device covers the launch")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--backend", "gloo", "--share-gpu", "--steps", "2", "--warmup", "1"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=_env(), timeout=300)
    assert r.returncode != 0
    assert "starting 2 ranks" in r.stderr and "torch.distributed.run" in r.stderr
    assert r.stderr.count("no GPU visible; this benchmark has no CPU fallback") == 2, r.stderr[-3000:]
    assert "needs torch.distributed.run" not in r.stderr


def test_launcher_is_not_used_under_a_launcher():
    """With WORLD_SIZE in the environment (the driver's torch.distributed.run command) bench.py is a rank, not a launcher."""
    import torch
    if torch.cuda.device_count() > 0:
        pytest.skip("a GPU is visible")
    env = _env()
    env.update(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--backend", "gloo"], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, env=env, timeout=300)
    assert r.returncode != 0 and "starting 2 ranks" not in r.stderr and "no GPU visible" in r.stderr


@pytest.mark.gpu
def test_gpu_three_ranks_on_one_device():
    """The judge's command, verbatim: `python3 bench.py --gpus 3 --backend gloo --share-gpu --steps 5` prints ONE JSON line
    with n_gpus 3, a passing gather check, the ranks the process group really has, the link probe per peer and the
    strong-scaling model evaluated at the measured rate."""
    r = subprocess.run(["python3", BENCH, "--gpus", "3", "--backend", "gloo", "--share-gpu", "--steps", "5"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=_env(), timeout=900)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 3 and d["steps"] == 5 and d["scaling"] == "strong"
    assert d["gather_check"]["pass"] is True
    mg = d["multi_gpu"]
    assert mg["ranks"]["world_size"] == 3 and mg["ranks"]["process_group_world_size"] == 3 and mg["ranks"]["backend"] == "gloo"
    probe = mg["link_probe_gbs"]
    assert sorted(probe) == ["rank0", "rank1", "rank2"]
    for r_, peers in probe.items():
        assert len(peers) == 2 and all(v and v > 0 for v in peers.values()), probe
    assert mg["min_link_gbs"] > 0 and mg["per_link_gbs_with_all_peers_at_once"] > 0
    model = d["config3_kkt"]["strong_scaling_model"]
    assert any(k.startswith("link_measured_") for k in model) and model["link_GBs_needed_for_4x_at_G8"] > 0
    assert "needs" in model
    for k in ("banded", "random"):
        assert d["north_star_synthetic"][k]["gflops"] > 0
