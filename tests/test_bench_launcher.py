"""`python bench.py --gpus N` without a launcher around it must BECOME N ranks: bench.py re-runs itself
under torch.distributed.run.  CPU: the whole distributed flow (rendezvous on 127.0.0.1, barrier, all-reduces, ragged all-gather
of per-trajectory results) over gloo with --plumbing-only (no solve: the product has no CPU path).  GPU: the same launcher path
with real solves, two ranks sharing the box's one GPU; and the refusal to run fewer ranks than asked for."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _line(stdout):
    return json.loads([ln for ln in stdout.splitlines() if ln.startswith('{"metric"')][-1])


def _clean_env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "MPCG_DIST_FORCE")}
    env.update(kw)
    return env


@pytest.mark.parametrize("scaling,batch,total", [("strong", 5, 5), ("weak", 3, 6)])
def test_bare_gpus2_becomes_two_gloo_ranks(scaling, batch, total):
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--plumbing-only", "--batch", str(batch), "--scaling", scaling],
                       capture_output=True, text=True, timeout=600, env=_clean_env(MPCG_DIST_BACKEND="gloo"))
    assert r.returncode == 0, r.stdout + r.stderr
    out = _line(r.stdout)
    assert out["n_gpus"] == 2 and out["self_launched"] is True and out["plumbing_only"] is True and out["value"] is None
    g = out["results_gather"]
    assert g["backend"] == "gloo" and g["trajectories"] == total and g["consistent_with_allreduce_sum"] is True
    assert out["per_rank"] == [0.5, 1.5]


def test_world_size_mismatch_is_an_error():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--plumbing-only"], capture_output=True, text=True, timeout=300,
                       env=_clean_env(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"))
    assert r.returncode != 0 and "--gpus 4 but WORLD_SIZE=1" in r.stderr


@pytest.mark.gpu
def test_more_ranks_than_devices_is_refused_not_degraded():
    import torch
    n = torch.cuda.device_count() + 1
    r = subprocess.run([sys.executable, BENCH, "--gpus", str(n), "--steps", "1", "--warmup", "0", "--no-extras", "--no-cpu-baseline", "--full-json", ""],
                       capture_output=True, text=True, timeout=300, env=_clean_env())
    assert r.returncode == 2 and "refusing to run fewer ranks" in r.stderr and '{"metric"' not in r.stdout


@pytest.mark.gpu
def test_self_launched_two_ranks_solve_on_the_gpu():
    # two ranks through bench.py's own launcher; on a one-GPU box both use device 0 and the collectives go over gloo
    # (RCCL cannot put two ranks on one device); on a multi-GPU node this is the real thing
    import torch
    env = _clean_env()
    if torch.cuda.device_count() < 2:
        env.update(MPCG_FORCE_DEVICE="0", MPCG_DIST_BACKEND="gloo")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "96", "--knots", "64", "--scaling", "strong",
                        "--no-extras", "--no-cpu-baseline", "--full-json", ""], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    out = _line(r.stdout)
    assert out["n_gpus"] == 2 and out["self_launched"] is True and out["scaling"] == "strong"
    assert out["results_gather"]["trajectories"] == 96 and out["results_gather"]["consistent_with_allreduce_sum"] is True
    assert len(out["per_rank_kernel_ms"]) == 2 and out["value"] > 0


@pytest.mark.gpu
def test_native_multi_device_driver():
    """examples/multi_gpu_pcg.cpp: C++ host threads (one per device) over the C ABI + one RCCL all-gather in a single-process
    communicator — on every device of the box (one here, eight on the scaling node), weak and strong partitioning; asking for more
    devices than there are is refused."""
    import torch
    from mpcgpu_amd import build
    exe = build.MULTI_BIN if os.path.exists(build.MULTI_BIN) else build.build_multi_gpu()
    nd = torch.cuda.device_count()
    for extra, total in ((["--batch", "40"], 40 * nd), (["--batch", "37", "--strong"], 37)):
        r = subprocess.run([exe, "--knots", "64", "--steps", "2", "--warmup", "1", *extra], capture_output=True, text=True, timeout=600, env=_clean_env())
        assert r.returncode == 0, r.stdout + r.stderr
        out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')][-1])
        assert out["n_gpus"] == nd and out["global_batch"] == total and out["value"] > 0 and len(out["per_device_ms_per_step"]) == nd
        assert out["results_gather"]["consistent_with_shard_sums"] is True and out["results_gather"]["trajectories"] == total
        assert 0 < out["mean_pcg_iters"] <= 167
    r = subprocess.run([exe, "--gpus", str(nd + 1)], capture_output=True, text=True, timeout=120, env=_clean_env())
    assert r.returncode == 2 and "refusing to run fewer" in r.stderr
