#!/usr/bin/env python
"""tools/sweep_soak.py [--runs N] (GPU box): soak of ods_sweep_lds_kernel (r06) for rare races -- LDS patch double buffer, block barriers, box reduction, register prefetch.
Per case the double volume is computed once by the GATHER kernel (two single-source sweeps through the C ABI) and then N times by msi_ods_sweep_volume (the LDS-staged
kernel at these shapes), every result compared bit for bit.  Cases: configs[2] (16 x 320 x 640, D = 64, bf16, identical poses: corner reuse + prefetch on every frame),
configs[3]'s per-GPU shard (4 x 640 x 1280, D = 32, fp32), and a mixed batch (poses / baselines changing between frames, a rotated source: recompute, fallback blocks)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from matryodshka_amd import MSI, _native as N
from matryodshka_amd.synthetic import make_inputs

ap = argparse.ArgumentParser()
ap.add_argument("--runs", type=int, default=300)
a = ap.parse_args()
m = MSI()


def rot(ax, ay, az):
    cx, sx, cy, sy, cz, sz = np.cos(ax), np.sin(ax), np.cos(ay), np.sin(ay), np.cos(az), np.sin(az)
    rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]]); ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]]); rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    r = np.eye(4); r[:3, :3] = rz @ ry @ rx
    return r.astype(np.float32)


def case(name, b, h, w, d, bf16, mixed):
    inp = make_inputs(1234 + b + d, b, h, w)
    ref = m.preprocess_image(torch.from_numpy(inp["ref_image"]))
    src = m.preprocess_image(torch.from_numpy(inp["src_image"]))
    p0 = np.tile(np.eye(4, dtype=np.float32)[None], (b, 1, 1)); p1 = p0.copy()
    intr = inp["intrinsics"].copy()
    if mixed:
        p1[1, 0, 3] = 0.01
        p1[2] = rot(0.3, -0.5, 0.2)
        p0[3] = rot(-0.1, 0.05, 0.0)
        if b > 6:
            intr[5:7, 0, 0] = 0.4
    t0, t1, ti = torch.from_numpy(p0).cuda(), torch.from_numpy(p1).cuda(), torch.from_numpy(intr).cuda()
    depths = torch.tensor(m.inv_depths(1.0, 100.0, d), dtype=torch.float32).cuda()
    trig = m._trig(h, w)
    dt = torch.bfloat16 if bf16 else torch.float32
    want = torch.zeros((b, h, w, 6 * d), dtype=dt, device="cuda")
    single = N.lib.msi_ods_sphere_sweep_bf16 if bf16 else N.lib.msi_ods_sphere_sweep_f32
    for k, (img, pose, order) in enumerate(((ref, t0, 1), (src, t1, -1))):
        N.check(single(img.data_ptr(), pose.data_ptr(), ti.data_ptr(), depths.data_ptr(), trig.data_ptr(), b, h, w, d, order, want.data_ptr(), 6 * d, k * 3 * d, None), "single")
    bufs = [torch.empty_like(want) for _ in range(3)]
    bad = 0
    for r in range(0, a.runs, 3):
        for o in bufs:      # three launches queued back to back, then compared
            N.check(N.lib.msi_ods_sweep_volume(ref.data_ptr(), src.data_ptr(), t0.data_ptr(), t1.data_ptr(), ti.data_ptr(), depths.data_ptr(), trig.data_ptr(),
                                               b, h, w, d, o.data_ptr(), int(bf16), None), "volume")
        bad += sum(0 if torch.equal(o, want) else 1 for o in bufs)
    print("%-34s %d x %d x %d, D = %d, %s: %d of %d volumes differ from the gather kernel's" % (name, b, h, w, d, "bf16" if bf16 else "fp32", bad, (a.runs + 2) // 3 * 3), flush=True)
    return bad


bad = case("configs[2] (identical frames)", 16, 320, 640, 64, True, False)
bad += case("configs[3] shard (identical frames)", 4, 640, 1280, 32, False, False)
bad += case("mixed poses / baselines", 8, 320, 640, 32, False, True)
bad += case("mixed poses / baselines, bf16 D = 64", 8, 320, 640, 64, True, True)
sys.exit(1 if bad else 0)
