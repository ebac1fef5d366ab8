#!/usr/bin/env python
"""Randomised parity sweep of the geometry kernels (K1 sweep, K3 assembly, K4 render rgb + depth, K5) against the
oracle: random batch / image size / sphere count / baseline / target position; identity poses (the test-time
configuration, bit-reproducible branches) and small random rigid poses (isolated branch flips allowed: 99.9th
percentile gate, as tests/test_gpu_geometry.py does).

    python tools/fuzz_geometry.py [--n 30] [--seed 0]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from matryodshka_amd import MSI
from oracle.msi import MSI as OracleMSI
from matryodshka_amd.synthetic import make_inputs, random_rgba

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=30)
ap.add_argument("--seed", type=int, default=0)
a = ap.parse_args()
rng = np.random.RandomState(a.seed)
m, o = MSI(), OracleMSI()
fails = 0
worst = 0.0
for it in range(a.n):
    b = int(rng.choice([1, 1, 2]))
    h, w = 2 * int(rng.randint(4, 40)), 2 * int(rng.randint(4, 60))
    d = 4 * int(rng.randint(1, 9))
    inp = make_inputs(int(rng.randint(1 << 30)), b, h, w)
    inp["intrinsics"][:, 0, 0] = rng.uniform(0.01, 0.08)
    posed = rng.rand() < 0.4
    if posed:
        th = rng.uniform(-0.05, 0.05)
        p = np.eye(4, dtype=np.float32)
        p[0, 0], p[0, 2], p[2, 0], p[2, 2] = np.cos(th), np.sin(th), -np.sin(th), np.cos(th)
        p[:3, 3] = rng.uniform(-0.02, 0.02, 3)
        inp["src_pose"] = np.tile(p[None], (b, 1, 1))
    planes = m.inv_depths(1.0, float(rng.uniform(20, 100)), d)
    ref, src = m.preprocess_image(torch.from_numpy(inp["ref_image"])), m.preprocess_image(torch.from_numpy(inp["src_image"]))
    psv = m.format_network_input(ref, src, inp["ref_pose"], inp["src_pose"], planes, inp["intrinsics"]).cpu().numpy()
    psv_o = o.format_network_input(o.preprocess_image(inp["ref_image"]), o.preprocess_image(inp["src_image"]),
                                   inp["ref_pose"], inp["src_pose"], planes, inp["intrinsics"])
    e_psv = np.abs(psv - psv_o)
    g_psv = float(np.percentile(e_psv, 99.9)) if posed else float(e_psv.max())
    pred = rng.uniform(-1, 1, size=(b, h, w, 2 * d)).astype(np.float32)
    out = m.assemble_layers(torch.from_numpy(psv_o).cuda(), torch.from_numpy(pred).cuda(), d)
    out_o = o.assemble(psv_o, pred, d)
    e_asm = float(np.abs(out["rgba_layers"].cpu().numpy() - out_o["rgba_layers"]).max())
    rgba = random_rgba(int(rng.randint(1 << 30)), b, h, w, d)
    rgb, dep = m.msi_render_equirect_view_and_depth(torch.from_numpy(rgba).cuda(), inp["tgt_pose_rt"], inp["tgt_pos"], planes, inp["intrinsics"])
    rgb_o = o.msi_render_equirect_view(rgba, inp["tgt_pose_rt"], inp["tgt_pos"], planes, inp["intrinsics"])
    dep_o = o.msi_render_equirect_depth(rgba, inp["tgt_pose_rt"], inp["tgt_pos"], planes, inp["intrinsics"])
    e_rgb, e_dep = float(np.abs(rgb.cpu().numpy() - rgb_o).max()), float(np.abs(dep.cpu().numpy() - dep_o).max())
    u8 = m.deprocess_image(rgb).cpu().numpy().astype(int) - o.deprocess_image(rgb_o).astype(int)
    ok = g_psv <= 1e-3 and e_asm == 0.0 and e_rgb <= 1e-3 and e_dep <= 1e-3 and np.abs(u8).max() <= 1
    worst = max(worst, g_psv, e_rgb, e_dep)
    fails += 0 if ok else 1
    print("%3d b=%d %3dx%-3d D=%2d %s psv %.1e  assemble %.1e  rgb %.1e  depth %.1e  u8 %d %s" % (
        it, b, h, w, d, "posed   " if posed else "identity", g_psv, e_asm, e_rgb, e_dep, np.abs(u8).max(), "" if ok else " <-- FAIL"), flush=True)
print("worst fp32 error %.2e (gate 1e-3); failures: %d" % (worst, fails))
sys.exit(1 if fails else 0)
