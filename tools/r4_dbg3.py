"""Round-4 diagnosis: device vs oracle, contribution by contribution, on the floor slice of the noise-free yard 128 x 2048 scan."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "d-liom_amd"))
sys.path.insert(0, ROOT)


def main():
    import dliom as dl
    from dliom import synth
    from oracle import oracle as orc
    ctx = dl.Context()
    with synth.scene("ground"):
        raw, _ = synth.scan(synth.trajectory_pose(0.5), 128, 2048)
        raw64, _ = synth.scan(synth.trajectory_pose(0.5), 64, 1024)
    pts = raw[orc.voxel_filter(0.15, raw)]
    keys = np.round(pts[:, 2].astype(np.float64) / 0.2)
    u, c = np.unique(keys, return_counts=True)
    floor = pts[keys == u[np.argmax(c)]]
    cube, _ = synth.scan(synth.trajectory_pose(0.7), 64, 1024)
    for name, p in (("floor_only", floor), ("whole", pts), ("yard64_filtered", raw64[orc.voxel_filter(0.15, raw64)]), ("yard64_raw", raw64),
                    ("cube", cube[orc.voxel_filter(0.15, cube)])):
        cloud = dl.PointCloud(ctx, p)
        gb, gv = dl.diag_histogram_contributions(ctx, cloud, 120)
        wb, wv = orc.histogram_contributions(p, 120)
        rec = {"case": name, "device": int(len(gb)), "oracle": int(len(wb))}
        k = min(len(gb), len(wb))
        bad = np.nonzero((gb[:k] != wb[:k]) | (gv[:k].view(np.uint32) != wv[:k].view(np.uint32)))[0]
        rec["mismatches_in_common_prefix"] = int(len(bad))
        if len(bad):
            i = int(bad[0])
            rec["first_bad"] = i
            rec["device_around"] = [[int(b), float(v)] for b, v in zip(gb[max(0, i - 3):i + 6], gv[max(0, i - 3):i + 6])]
            rec["oracle_around"] = [[int(b), float(v)] for b, v in zip(wb[max(0, i - 3):i + 6], wv[max(0, i - 3):i + 6])]
        print(json.dumps(rec), flush=True)
        cloud.close()


if __name__ == "__main__":
    main()
