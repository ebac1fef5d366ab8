#!/usr/bin/env python
"""CNN-only timing at the BASELINE config (640x320, Cin=192 -> 64, ngf=64): HIP-event time of
msi_net_forward_f32 per frame and the achieved fp32 TFLOP/s.  Used with
`rocprofv3 --kernel-trace` + tools/rocprof_summary.py for the per-layer breakdown."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--warmup", type=int, default=3)
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--height", type=int, default=320)
ap.add_argument("--width", type=int, default=640)
ap.add_argument("--planes", type=int, default=32)
ap.add_argument("--no-coord-net", action="store_true")
ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"])
ap.add_argument("--opt", action="append", default=[], help="plan option KEY=VALUE (msi_net_plan_set_option), repeatable")
a = ap.parse_args()

from matryodshka_amd import MSI, nets
coord = not a.no_coord_net
cin, nout = 6 * a.planes, 2 * a.planes
m = MSI(weights=nets.init_weights(cin, nout, 64, coord), coord_net=coord, dtype=a.dtype)
for kv in a.opt:
    k, v = kv.split("=")
    m.net_options[int(k)] = int(v, 0)
x = torch.rand((a.batch, a.height, a.width, cin), device="cuda") * 2 - 1
if a.dtype == "bf16":
    x = x.bfloat16()
for _ in range(a.warmup):
    m.run_net(x, nout, 64)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.steps):
    m.run_net(x, nout, 64)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / a.steps
fl = bench.cnn_flops(a.height, a.width, cin, nout, 64, coord) * a.batch
peak = 2500.0 if a.dtype == "bf16" else bench.PEAK_FP32_MFMA_TFLOPS
print("cnn forward: %.3f ms/call (batch %d, %s)  %.1f TFLOP/s  (%.1f%% of %.1f)" % (
    ms, a.batch, a.dtype, fl / ms / 1e9, 100 * fl / ms / 1e9 / peak, peak))
