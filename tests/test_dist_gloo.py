"""CPU, 2 processes over gloo: the N>1 path of bench.py — shard trajectories, solve the shard,
gather (iters, exit flags) into global order, max-over-ranks timing.  On the GPU box the same
code runs with backend "nccl" (RCCL); here the shard solve is stood in for by the oracle."""
import os
import socket
import sys

import numpy as np
import torch
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, N, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import oracle as orc
    from mpcgpu_amd import dist as D
    from mpcgpu_amd import synth
    r, lr, w = D.init("gloo")
    assert (r, w) == (rank, world)
    lo, hi = D.shard_range(total, rank, world)
    k = synth.make_kkt(N, total, 77)
    S, P, g = synth.form_schur(k)
    it = np.zeros(hi - lo, np.int32)
    ex = np.zeros(hi - lo, np.uint8)
    for j, b in enumerate(range(lo, hi)):
        res = orc.pcg(S[b], P[b], g[b], np.zeros(14 * N, np.float32), N, 30, 1e-3, "ss")
        it[j], ex[j] = res["iters"], res["max_iter_exit"]
    D.barrier()
    git, gex = D.gather_results(torch.from_numpy(it), torch.from_numpy(ex), total)
    tmax = D.max_over_ranks(1.0 + rank)
    tsum = D.sum_over_ranks(float(it.sum()))
    q.put((rank, git.numpy().copy(), gex.numpy().copy(), tmax, tsum))
    torch.distributed.destroy_process_group()


def test_two_rank_shard_and_gather(orc):
    from mpcgpu_amd import synth
    total, N, world = 5, 4, 2           # ragged: shards of 3 and 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, world, port, total, N, q)) for r in range(world)]
    for p in ps:
        p.start()
    outs = sorted([q.get(timeout=180) for _ in ps], key=lambda t: t[0])
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    k = synth.make_kkt(N, total, 77)
    S, P, g = synth.form_schur(k)
    want = [orc.pcg(S[b], P[b], g[b], np.zeros(14 * N, np.float32), N, 30, 1e-3, "ss") for b in range(total)]
    wit = np.array([w["iters"] for w in want])
    wex = np.array([w["max_iter_exit"] for w in want], np.uint8)
    for rank, git, gex, tmax, tsum in outs:
        np.testing.assert_array_equal(git, wit)
        np.testing.assert_array_equal(gex, wex)
        assert tmax == 2.0 and tsum == wit.sum()
