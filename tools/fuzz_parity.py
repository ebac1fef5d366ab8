#!/usr/bin/env python
"""Randomised parity sweep of the network (K2) against the oracle: random (batch, H, W, Cin, Cout, ngf, CoordNet,
dtype) within the supported set, small enough for the CPU oracle.  Exercises M-tile tails, channel tails
(Cin, ngf not multiples of 32), N tails, tiny widths (column border classes overlapping) and both padding modes.

    python tools/fuzz_parity.py [--n 40] [--seed 0] [--halo]

--halo: shapes that reach the halo-patch kernels (ngf 32 / 64, maps in multiples of 4 x 16 / 8 x 16 / 16 x 16, channel
counts in multiples of 32 / 64) with a random MSI_NET_OPT_HALO in {0, 1, 3, 5, 7}, a random FIXUP_KERNEL and a random NUM_CUS
(8 / 16 CUs in the plan: the stride-2 halo kernels and the K splits are then reached by small grids).
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from matryodshka_amd import MSI
from oracle import nets as onets

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=40)
ap.add_argument("--seed", type=int, default=0)
ap.add_argument("--halo", action="store_true")
a = ap.parse_args()
from matryodshka_amd import _native as N
rng = np.random.RandomState(a.seed)
worst = {"f32": 0.0, "bf16": 0.0}
fails = 0
for it in range(a.n):
    dtype = "bf16" if rng.rand() < 0.3 else "f32"
    q = 8 if dtype == "bf16" else 4
    b = int(rng.choice([1, 1, 2, 3]))
    h = 8 * int(rng.randint(1, 9))
    w = 8 * int(rng.randint(1, 13))
    cin = q * int(rng.randint(1, 13))
    nout = 4 * int(rng.randint(1, 9))
    ngf = q * int(rng.randint(1, 6))
    coord = bool(rng.rand() < 0.6)
    opts = {}
    if a.halo:
        h = 16 * int(rng.randint(1, 7))
        w = 16 * int(rng.randint(1, 9))
        if rng.rand() < 0.5:
            h, w = 32 * int(rng.randint(1, 4)), 128 * int(rng.randint(1, 3))   # every level tiles into patches
        ngf = int(rng.choice([32, 64]))
        cq = 64 if dtype == "bf16" else 32
        cin = cq * int(rng.randint(1, 4)) if rng.rand() < 0.7 else q * int(rng.randint(1, 13))
        opts = {N.NET_OPT_HALO: int(rng.choice([0, 1, 3, 5, 5, 7])), N.NET_OPT_FIXUP_KERNEL: int(rng.rand() < 0.3),
                N.NET_OPT_BIGTILE: int(rng.choice([1, 1, 2])), N.NET_OPT_NUM_CUS: int(rng.choice([256, 8, 16]))}
    if not coord:          # wrap_pad(x, 2, 2) at 1/8 resolution needs at least two columns / rows (the reference fails below that too)
        h, w = max(h, 16), max(w, 16)
    weights = onets.init_weights(cin, nout, ngf=ngf, coord_net=coord, seed=int(rng.randint(1 << 30)), randomize_affine=True)
    x = rng.uniform(-1, 1, size=(b, h, w, cin)).astype(np.float32)
    if dtype == "bf16":
        x = onets.bf16_round(x)
    m = MSI(weights=weights, coord_net=coord, dtype=dtype)
    m.net_options.update(opts)
    xt = torch.from_numpy(x).cuda()
    pred = m.run_net(xt.bfloat16() if dtype == "bf16" else xt, nout, ngf).cpu().numpy()
    ref = onets.forward(weights, x, coord_net=coord, bf16=dtype == "bf16")
    err = float(np.abs(pred - ref).max())
    tol = 6e-2 if dtype == "bf16" else 1e-3
    ok = np.isfinite(pred).all() and err <= tol
    worst[dtype] = max(worst[dtype], err)
    fails += 0 if ok else 1
    print("%3d %-4s b=%d %3dx%-3d cin=%3d nout=%2d ngf=%2d coord=%d %s max-abs %.2e %s" % (
        it, dtype, b, h, w, cin, nout, ngf, coord, " ".join("%d=%d" % kv for kv in sorted(opts.items())), err,
        "" if ok else "  <-- FAIL"), flush=True)
print("worst max-abs: f32 %.2e (gate 1e-3), bf16 %.2e (gate 6e-2); failures: %d" % (worst["f32"], worst["bf16"], fails))
sys.exit(1 if fails else 0)
