"""What bench.py --gpus N (N > 1) adds to its sharded lines so that the FIRST run on real multi-GPU hardware is
informative (VERDICT r5 item 8 -- until the driver finds an 8-GPU node this code has only seen one device: the N-rank
control flow runs under DLIOM_BENCH_BACKEND=gloo with every rank on GPU 0):

  ranks_seen           every rank's own view of its communicator (RCCL: ncclCommCount; gloo: the process group), all ranks
  winner check         every rank's SHARDED match of every distinct scan against rank 0's UNSHARDED match of the same scan
                       on its own replica of the submap: score bits and pose bits equal, on every rank
  replica consistency  CRC-32 of both grids' cells after the timed steps, equal on all ranks (the replicas ran the same
                       deterministic stream; the sharded stream inserts at the all-reduced winner)
  all-reduce time      the collective's own time per match on each rank's stream (HIP events inside the library around
                       copy + ncclAllReduce + copy, DLIOM_KERNEL_ALLREDUCE; the gloo callback: host clock around it)

Nothing here is timed as part of `value`; the oracle is not involved (rank 0's unsharded device match is the reference)."""
import time
import zlib

import numpy as np


def gather_objects(dist, obj, world):
    if dist is None or world == 1:
        return [obj]
    out = [None] * world
    dist.all_gather_object(out, obj)
    return out


def grid_crc(dg):
    """CRC-32 over the grid's non-zero cells in a canonical order (leaf origin, then cell)."""
    origins, values = dg.download_blocks()
    if len(origins) == 0:
        return 0
    order = np.lexsort((origins[:, 0], origins[:, 1], origins[:, 2]))
    crc = zlib.crc32(np.ascontiguousarray(origins[order]).tobytes())
    return int(zlib.crc32(np.ascontiguousarray(values[order]).tobytes(), crc))


def checks(dl, ctx, dist, world, rank, rt, shard, scans, g_hi, g_lo, sharded_rtcsm, ranks_seen_here):
    """Collective: every rank calls it with the same scans after the timed regions.  Returns the dict rank 0 prints."""
    same, detail = True, []
    for k, sc in enumerate(scans):
        s_sh, p_sh = sharded_rtcsm(shard, sc, g_hi)                  # collective: all ranks, same scan
        ref = None
        if rank == 0:
            s_un, p_un = rt.Match(sc["init"], sc["cloud"], g_hi)     # rank 0 only: the unsharded match
            ref = (np.float32(s_un).tobytes(), np.asarray(p_un, np.float64).tobytes())
        box = [ref]
        if dist is not None and world > 1:
            dist.broadcast_object_list(box, src=0)
        mine = (np.float32(s_sh).tobytes(), np.asarray(p_sh, np.float64).tobytes())
        ok = mine == box[0]
        same = same and ok
        detail.append(bool(ok))
    crcs = (grid_crc(g_hi), grid_crc(g_lo))
    everyone = gather_objects(dist, {"rank": rank, "ranks_seen": int(ranks_seen_here), "winners_equal_rank0_unsharded": bool(same),
                                     "per_scan": detail, "grid_crc32": crcs}, world)
    return {"ranks_seen_by_rank": [e["ranks_seen"] for e in everyone],
            "sharded_winner_bit_equal_to_rank0_unsharded_on_every_rank": bool(all(e["winners_equal_rank0_unsharded"] for e in everyone)),
            "scans_checked": len(scans),
            "replica_grids_crc32_equal_on_all_ranks": bool(all(tuple(e["grid_crc32"]) == tuple(everyone[0]["grid_crc32"]) for e in everyone)),
            "grid_crc32_rank0": list(everyone[0]["grid_crc32"]),
            "failing_ranks": [e["rank"] for e in everyone if not e["winners_equal_rank0_unsharded"] or
                              tuple(e["grid_crc32"]) != tuple(everyone[0]["grid_crc32"])]}


class CollectiveClock:
    """Time spent in the data-path collective per sharded match on this rank: HIP events inside the library for the RCCL
    entry point (ctx.kernel_time(KERNEL_ALLREDUCE)), the host clock around the callback otherwise."""

    def __init__(self, dl, ctx, uses_rccl):
        self.dl, self.ctx, self.uses_rccl = dl, ctx, uses_rccl
        self.host_s, self.calls = 0.0, 0

    def wrap(self, all_reduce):
        def timed(v):
            t0 = time.perf_counter()
            r = all_reduce(v)
            self.host_s += time.perf_counter() - t0
            self.calls += 1
            return r
        return timed

    def reset(self):
        self.host_s, self.calls = 0.0, 0

    def ms_per_match(self, matches):
        if self.uses_rccl:
            ms, n = self.ctx.kernel_time(self.dl.KERNEL_ALLREDUCE)
            return (ms / n) if n else None
        return (1e3 * self.host_s / self.calls) if self.calls else None
