"""World-size-2 and world-size-8 gloo tests (CPU) of the candidate-sharding PROTOCOL (dliom/sharded.py): N ranks, each
owning a contiguous share of the candidate rotations, reach the unsharded winner through the two MAX all-reduces of the
two-phase protocol and through the ONE exchange of the one-collective protocol.

What this file does NOT cover: the product's C entry points.  There is no GPU in this container, so the local shard
here is an oracle-backed stand-in (OracleShard: the oracle's exact per-candidate scores behind the interface of
dliom.RtcsmShard) -- the packing, the tie rule, the failing-rank word and the collectives' control flow are what is
tested.  dliom_rtcsm3d_match_sharded / _sharded_rccl themselves run in tests/test_gpu_sharded.py (eight processes
through the C entry point on one GPU, a failing rank, a one-rank RCCL communicator) and
tests/test_gpu_parity.py::test_sharded_match_single_process; bench.py --gpus N adds the per-rank winner check against
rank 0's unsharded match (benchlib/multigpu.py)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    for p in (ROOT, os.path.join(ROOT, "d-liom_amd"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as orc
    from dliom import sharded, synth
    from helpers import DEFAULT_RTCSM, build_oracle_submap
    og = build_oracle_submap(orc, 0.1, num_scans=4, beams=16, azimuths=128)
    truth = synth.trajectory_pose(0.4)
    pts, _ = synth.scan(truth, 8, 64)
    pts = synth.range_filter(pts, 15.0)
    init = synth.perturb_pose(truth, 0.1, 0.5, seed=21)
    ref = orc.rtcsm3d_match(DEFAULT_RTCSM, init, pts, og, want_scores=True)
    w = orc.rtcsm3d_window(DEFAULT_RTCSM, 0.1, pts)
    R = (2 * w["angular_window"] + 1) ** 3
    _, cand = orc.rtcsm3d_candidates(DEFAULT_RTCSM, 0.1, pts, init)

    class OracleShard:
        """begin/finish/decode over this rank's rotations, exact scores as (trivial) bounds."""

        def begin(self, init_, cloud, grid):
            r = np.arange(len(ref["scores"])) % R
            self.mine = (r >= R * rank // world) & (r < R * (rank + 1) // world)
            s = np.where(self.mine, ref["scores"], np.float32(0))
            return int(np.float32(s.max()).view(np.uint32))

        def finish(self, global_lo_bits):
            lo = np.uint32(global_lo_bits).view(np.float32)
            alive = self.mine & (ref["scores"] >= lo)
            if not alive.any():
                return 0
            s = np.where(alive, ref["scores"], np.float32(-1))
            best = int(np.argmax(s))  # first maximum = lowest index
            return sharded.pack_winner(s[best], best)

        def decode(self, packed):
            score, index = sharded.unpack_winner(packed)
            return score, cand[index].astype(np.float64), index

        def match(self, init_, cloud, grid, exchange):
            """dliom_rtcsm3d_match_sharded's protocol: own exact winner, ONE exchange (a rank whose shard holds no
            candidate -- more ranks than rotations -- contributes 0)."""
            self.begin(init_, cloud, grid)
            local = self.finish(np.float32(0).view(np.uint32)) if self.mine.any() else 0
            score_, pose_, index_ = self.decode(exchange(local))
            return score_, pose_, index_

    score, pose, index = sharded.sharded_match_two_phase(OracleShard(), init, pts, og, dist=dist)
    ok = (index == ref["best_index"] and np.float32(score) == np.float32(ref["score"]) and
          np.array_equal(pose, ref["pose"]))
    # ... and the one-collective protocol the C entry points implement
    score1, pose1, index1 = sharded.sharded_match(OracleShard(), init, pts, og, dist=dist)
    ok = ok and index1 == index and np.float32(score1) == np.float32(score) and np.array_equal(pose1, pose)
    ret[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_sharded_protocol_world_size(world):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert all(ret.get(r) is True for r in range(world)), dict(ret)


def test_pack_winner_orders_like_the_reference():
    from dliom import sharded
    a = sharded.pack_winner(0.5, 10)
    b = sharded.pack_winner(0.5, 3)      # same score, lower index wins
    c = sharded.pack_winner(0.50000006, 99)  # strictly greater score wins regardless of index
    assert b > a and c > b
    assert sharded.unpack_winner(b) == (0.5, 3)
