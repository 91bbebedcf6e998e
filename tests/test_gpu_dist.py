"""GPU: the multi-GPU flow of bench.py on ONE GPU — the process group is created even at world size 1
(MPCG_DIST_FORCE=1), so the RCCL (backend "nccl") barrier, all-reduce and the all-gather of per-trajectory results
run on real hardware instead of being dead code on a single-GPU box.  Strong scaling mode shards a global
batch with shard_range."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_runs_its_collectives_over_rccl_at_world_size_one(scaling):
    env = dict(os.environ, MPCG_DIST_FORCE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "96",
                        "--knots", "64", "--scaling", scaling, "--no-extras", "--no-cpu-baseline", "--full-json", ""],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')][-1])
    assert out["scaling"] == scaling and out["n_gpus"] == 1
    g = out["results_gather"]
    assert g["backend"] == "nccl (RCCL)" and g["trajectories"] == 96 and g["consistent_with_allreduce_sum"] is True
    assert out["value"] > 0 and out["config"]["kernel_family"] in ("pcg_lqb_kernel", "pcg_lpk_kernel", "pcg_rpl_kernel")
