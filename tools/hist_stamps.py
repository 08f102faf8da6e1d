"""Cycle stamps of the histogram's big-slice kernels (experiments build: make -C d-liom_amd experiments; DLIOM_LIB is set
here).  Prints, for the level yard scan (one floor slice of ~10 000 returns), the cycles between the DLIOM_BSTAMP /
DLIOM_SSTAMP marks of big_prepare_kernel, big_sort_order and big_slice_kernel (block 0).  s_memtime: shader clock cycles (~2.4 GHz)."""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("DLIOM_LIB", os.path.join(ROOT, "d-liom_amd", "ab", "libdliom_exp.so"))
sys.path[:0] = [ROOT, os.path.join(ROOT, "d-liom_amd")]
import numpy as np  # noqa: E402

import dliom as dl  # noqa: E402
from dliom import synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402


def main():
    lib = dl.load_library()
    ctx = dl.Context(0)
    out = {}
    for name, beams, azimuths, size in (("yard_64x1024_level", 64, 1024, 0.15), ("yard_128x2048", 128, 2048, 0.15)):
        with synth.scene("ground"):
            raw, _ = synth.scan(synth.trajectory_pose(0.5), beams, azimuths)
        pts = raw[orc.voxel_filter(size, raw)]
        cloud = dl.PointCloud(ctx, pts)
        for _ in range(4):
            dl.cloud_rotational_histogram(ctx, cloud, 120, None)
        ctx.synchronize()
        buf = (ctypes.c_ulonglong * (64 * 16))()
        lib.dliom_exp_rothist_big_stamps.argtypes = [ctypes.c_void_p]
        lib.dliom_exp_rothist_big_stamps.restype = ctypes.c_int
        rc = lib.dliom_exp_rothist_big_stamps(ctypes.cast(buf, ctypes.c_void_p))
        a = np.array(buf[:], dtype=np.uint64).reshape(64, 16).astype(np.int64)
        rec = {"rc": rc, "points": int(len(pts))}
        for label, row, marks in (("big_prepare", 0, 5), ("big_slice", 4, 7), ("sort_order", 8, 15)):
            st = a[row]
            rec[label] = {"ticks_from_first": [int(st[k] - st[0]) if st[k] else None for k in range(marks)]}
        es = (ctypes.c_ulonglong * 16)()
        lib.dliom_exp_exact_sum_stamps.argtypes = [ctypes.c_void_p]
        lib.dliom_exp_exact_sum_stamps.restype = ctypes.c_int
        lib.dliom_exp_exact_sum_stamps(ctypes.cast(es, ctypes.c_void_p))
        e = [int(v) for v in es]
        # the last call of block 0 (big_slice_kernel's centroid): 0 start, 10 / 11 / 12 loads+prefix / block scans /
        # element functions of the LAST array, 1 + k array k classified, 8 walk done
        rec["exact_sum_last_call"] = {str(k): (e[k] - e[0]) for k in (10, 11, 12, 1, 2, 8)}
        rec["exact_sum_walk_array0"] = {"unsafe_chunks": e[13], "steps": e[14], "one_by_one": e[15] & 0xffffffff, "from_memory": e[15] >> 32}
        rec["slice_count"] = int(a[4][10])
        rec["m"] = int(a[4][11])
        out[name] = rec
        cloud.close()
    # slice_kernel (every slice in LDS): the cube scene's scan, per workgroup (= slice) the cycles between the marks
    with synth.scene("cube"):
        raw, _ = synth.scan(synth.trajectory_pose(0.7), 64, 1024)
    pts = raw[orc.voxel_filter(0.15, raw)]
    cloud = dl.PointCloud(ctx, pts)
    for _ in range(4):
        dl.cloud_rotational_histogram(ctx, cloud, 120, None)
    ctx.synchronize()
    buf = (ctypes.c_ulonglong * (64 * 16))()
    lib.dliom_exp_rothist_stamps.argtypes = [ctypes.c_void_p]
    lib.dliom_exp_rothist_stamps.restype = ctypes.c_int
    lib.dliom_exp_rothist_stamps(ctypes.cast(buf, ctypes.c_void_p))
    a = np.array(buf[:], dtype=np.uint64).reshape(64, 16).astype(np.int64)
    rows = []
    for b in range(64):
        st = a[b]
        if st[7] == 0:
            continue
        order = [0, 8, 9, 1, 2, 3, 13, 14, 15, 4, 5, 6, 7]
        names = ["find", "compact_count", "compact_offsets", "compact_copy", "centroid", "angles", "bitonic", "replay_in", "replay", "order", "sorted+centroid", "chain+contrib", "write"]
        prev = st[0]
        rec = {"count": int(st[10]), "m": int(st[11]), "total": int(st[7] - st[0])}
        for k, nm in zip(order[1:], names[1:]):
            if st[k] == 0 or st[k] < prev:
                continue
            rec[nm] = int(st[k] - prev)
            prev = st[k]
        rows.append(rec)
    rows.sort(key=lambda r: -r["total"])
    out["cube_slice_kernel_slowest"] = rows[:4]
    out["cube_slice_kernel_workgroups"] = len(rows)
    cloud.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
