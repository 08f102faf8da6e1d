"""BASELINE config 4 through the C entry points: dliom_rtcsm3d_match_sharded with a gloo collective between two and
between EIGHT processes (all on GPU 0 -- the collective is the caller's, so a one-GPU box can run the N-rank protocol,
including the rank that fails before the exchange), and
dliom_rtcsm3d_match_sharded_rccl on a one-rank RCCL communicator (ncclAllReduce really runs)."""
import ctypes as C
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _scene(dl, ctx):
    from oracle import oracle as orc
    from dliom import synth
    from helpers import DEFAULT_RTCSM, build_oracle_submap, to_device_grid
    og = build_oracle_submap(orc, 0.1, num_scans=6, beams=16, azimuths=256)
    truth = synth.trajectory_pose(0.6)
    pts, _ = synth.scan(truth, 32, 512)
    init = synth.perturb_pose(truth, 0.1, 0.5, seed=13)
    return orc, og, to_device_grid(dl, ctx, og), pts, init, DEFAULT_RTCSM


def _worker(rank, world, port, ret):
    for p in (ROOT, os.path.join(ROOT, "d-liom_amd"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import dliom as dl
    from dliom import sharded
    ctx = dl.Context(0)
    orc, og, dg, pts, init, opts = _scene(dl, ctx)
    cloud = dl.PointCloud(ctx, pts)
    shard = dl.RtcsmShard(ctx, opts, rank, world)
    score, pose = sharded.sharded_match(shard, init, cloud, dg, dist=dist)  # -> dliom_rtcsm3d_match_sharded
    ref = orc.rtcsm3d_match(opts, init, pts, og)
    ret[rank] = bool(np.float32(score) == np.float32(ref["score"]) and np.array_equal(pose, ref["pose"]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_ranks_one_collective_through_the_c_entry_point(world):
    """world = 8 is the control flow of the driver's 8-GPU run (eight shards of the rotations, one exchange) with every
    rank on GPU 0: the collective is the caller's, so a one-GPU box can run it."""
    import torch.multiprocessing as mp
    mpc = mp.get_context("spawn")
    ret = mpc.Manager().dict()
    port = _free_port()
    procs = [mpc.Process(target=_worker, args=(r, world, port, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    assert all(ret.get(r) is True for r in range(world)), dict(ret)


def _failing_worker(rank, world, port, ret):
    """The last rank hands in an empty cloud (DLIOM_ERR_EMPTY_CLOUD before any kernel): it must still join the collective,
    and every other rank must come back with DLIOM_ERR_PEER_FAILED instead of hanging in the all-reduce."""
    for p in (ROOT, os.path.join(ROOT, "d-liom_amd"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import dliom as dl
    from dliom import sharded
    ctx = dl.Context(0)
    orc, og, dg, pts, init, opts = _scene(dl, ctx)
    cloud = dl.PointCloud(ctx, pts if rank != world - 1 else pts[:0])  # the LAST rank fails
    shard = dl.RtcsmShard(ctx, opts, rank, world)
    try:
        sharded.sharded_match(shard, init, cloud, dg, dist=dist)
        ret[rank] = "no error"
    except dl.DliomError as e:
        ret[rank] = e.status
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_a_failing_rank_still_joins_the_collective(world):
    import torch.multiprocessing as mp
    import dliom as dl
    mpc = mp.get_context("spawn")
    ret = mpc.Manager().dict()
    port = _free_port()
    procs = [mpc.Process(target=_failing_worker, args=(r, world, port, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0, "a rank hung or crashed"
    assert all(ret.get(r) == dl.ERR_PEER_FAILED for r in range(world - 1)) and ret.get(world - 1) == dl.ERR_EMPTY_CLOUD, dict(ret)


def _rccl_one_rank_main():
    """Body of the RCCL test; runs in a process of its own (see the test)."""
    for p in (ROOT, os.path.join(ROOT, "d-liom_amd"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    import dliom as dl
    dl.load_library()
    rccl = C.CDLL("librccl.so.1")

    class UniqueId(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]

    uid = UniqueId()
    rc = rccl.ncclGetUniqueId(C.byref(uid))
    assert rc == 0, "ncclGetUniqueId -> %d" % rc
    comm = C.c_void_p()
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    ctx = dl.Context(0)
    rc = rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0)
    assert rc == 0, "ncclCommInitRank -> %d" % rc
    orc, og, dg, pts, init, opts = _scene(dl, ctx)
    cloud = dl.PointCloud(ctx, pts)
    score, pose = dl.RtcsmShard(ctx, opts, 0, 1).match_rccl(init, cloud, dg, comm.value)
    ref = orc.rtcsm3d_match(opts, init, pts, og)
    assert np.float32(score) == np.float32(ref["score"]) and np.array_equal(pose, ref["pose"])
    rccl.ncclCommDestroy.argtypes = [C.c_void_p]
    rccl.ncclCommDestroy(comm)
    cloud.close()
    dg.close()
    ctx.close()
    print("rccl-one-rank ok")


def test_rccl_entry_point_on_a_one_rank_communicator():
    """In a fresh process: RCCL's bootstrap (sockets, its own HIP streams and IPC probing) must not depend on
    what the 100+ earlier tests left behind in the pytest process.  RCCL's own log is part of the failure text."""
    import subprocess
    env = dict(os.environ, NCCL_DEBUG="WARN", HSA_ENABLE_IPC_MODE_LEGACY="0", NCCL_SOCKET_IFNAME="lo",
               MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "rccl-one-rank"], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0 and "rccl-one-rank ok" in r.stdout, r.stdout[-3000:] + "\n" + r.stderr[-3000:]


if __name__ == "__main__" and sys.argv[1:] == ["rccl-one-rank"]:
    _rccl_one_rank_main()
