"""Search-window sharding of RealTimeCorrelativeScanMatcher3D::Match across the GPUs of a node
(BASELINE config 4).

Every rank holds the scan and the submap grid (replicated: a 64x1024 scan is 786 KB) and owns a
contiguous range of the candidate rotations.  Two 8-byte MAX all-reduces over RCCL/xGMI make the
result identical to the unsharded match on every rank:

  1. the best score lower bound  -> every rank prunes against the GLOBAL bound
  2. the packed winner (score_bits << 32 | 0xFFFFFFFF - candidate_index): positive floats order
     like their bit patterns and the complemented index makes the LOWER index win ties -- the
     reference's "first strictly greater score in generation order" (rtcsm_3d.cc:46-51).

Both messages are latency bound (8 B); link bandwidth is irrelevant.  `shard` only needs
begin(...) -> int, finish(int) -> int and decode(int) -> (score, pose): dliom.RtcsmShard on a GPU.
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
    """Runs one sharded match; returns (score, pose_estimate[7]) -- the same on every rank."""
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
