#!/usr/bin/env python
"""Bitwise reproducibility of the network forward at sizes where the tail split and the in-launch fix-up (last
arriver sums the K-range slabs) are active: N runs must equal the first, and the in-launch result must match the
separate-launch fix-up (plan option FIXUP_KERNEL) bit for bit; also tests/test_gpu_cnn.py.   python tools/determinism.py [runs]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from matryodshka_amd import MSI, nets

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for dtype, batch in (("f32", 1), ("f32", 2), ("bf16", 1)):
    m = MSI(weights=nets.init_weights(192, 64, 64, True), coord_net=True, dtype=dtype)
    x = torch.rand((batch, 320, 640, 192), device="cuda") * 2 - 1
    if dtype == "bf16":
        x = x.bfloat16()
    ref = m.run_net(x, 64, 64).clone()
    bad = sum(0 if torch.equal(m.run_net(x, 64, 64), ref) else 1 for _ in range(runs))
    from matryodshka_amd import _native as N
    m2 = MSI(weights=nets.init_weights(192, 64, 64, True), coord_net=True, dtype=dtype)
    m2.net_options[N.NET_OPT_FIXUP_KERNEL] = 1
    alt = m2.run_net(x, 64, 64).clone()
    print("%s batch %d: %d/%d runs bitwise-identical, finite: %s, max |in-launch - fix-up kernel| = %.2e" % (
        dtype, batch, runs - bad, runs, bool(torch.isfinite(ref).all()), float((alt - ref).abs().max())), flush=True)
