"""Search-window sharding of RealTimeCorrelativeScanMatcher3D::Match across the GPUs of a node
(BASELINE config 4).

Every rank holds the scan and the submap grid (replicated: a 64x1024 scan is 786 KB) and owns a
contiguous range of the candidate rotations.  ONE 8-byte MAX all-reduce over RCCL/xGMI makes the result
identical to the unsharded match on every rank: each rank finds its own winner exactly and contributes the
packed word (score_bits << 32 | 0xFFFFFFFF - candidate_index) -- positive floats order like their bit
patterns and the complemented index makes the LOWER index win ties, the reference's "first strictly greater
score in generation order" (rtcsm_3d.cc:46-51).  The C entry points are dliom_rtcsm3d_match_sharded (the
collective is a callback) and dliom_rtcsm3d_match_sharded_rccl (an ncclComm_t); this module is the Python side
used by bench.py and the tests.  `sharded_match_two_phase` is the older protocol (global lower bound first:
less rescoring per rank, two collectives); `shard` there only needs begin / finish / decode.
"""
import numpy as np


def all_reduce_max_int(value, dist=None, device=None):
    """MAX all-reduce of one non-negative integer < 2**63 (an int64 tensor on `device`)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return int(value)
    import torch
    t = torch.tensor([int(value)], dtype=torch.int64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return int(t.item())


def sharded_match(shard, initial_pose_estimate, cloud, hybrid_grid, dist=None, device=None):
    """One sharded match through dliom_rtcsm3d_match_sharded (one collective); the same result on every rank."""
    return shard.match(initial_pose_estimate, cloud, hybrid_grid, lambda v: all_reduce_max_int(v, dist, device))


def sharded_match_two_phase(shard, initial_pose_estimate, cloud, hybrid_grid, dist=None, device=None):
    """The two-collective protocol over begin / finish / decode; returns what decode returns."""
    local_lo = shard.begin(initial_pose_estimate, cloud, hybrid_grid)
    global_lo = all_reduce_max_int(local_lo, dist, device)
    local_best = shard.finish(global_lo)
    global_best = all_reduce_max_int(local_best, dist, device)
    return shard.decode(global_best)


def pack_winner(score, index):
    """(score_bits << 32) | (0xFFFFFFFF - index) for a positive float32 score."""
    bits = int(np.float32(score).view(np.uint32))
    return (bits << 32) | (0xFFFFFFFF - int(index))


def unpack_winner(packed):
    bits = np.uint32(packed >> 32)
    return float(bits.view(np.float32)), 0xFFFFFFFF - (packed & 0xFFFFFFFF)
