#!/usr/bin/env python
"""Stress check (GPU box): eight DIFFERENT network inputs queued back to back without a host sync, hundreds of rounds, both networks -- every
prediction must equal its solo run bit for bit (a stale K-range slab / ticket / LayerNorm shard cannot hide behind a repeated input).
r04: 0 mismatching forwards of 5 600.  The short form runs as tests/test_gpu_split3.py::test_different_frames_back_to_back_equal_their_solo_runs."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from matryodshka_amd import MSI, _native as N
from oracle import nets as onets
for coord, b, h, w, cin, nout, ngf in ((True, 1, 160, 320, 96, 32, 64), (False, 4, 128, 256, 48, 16, 64), (True, 1, 320, 640, 192, 64, 64), (False, 1, 320, 640, 192, 64, 64)):
    weights = onets.init_weights(cin, nout, ngf=ngf, coord_net=coord, seed=41, randomize_affine=True)
    m = MSI(weights=weights, coord_net=coord)
    g = torch.Generator(device="cuda").manual_seed(7)
    xs = [torch.rand((b, h, w, cin), device="cuda", generator=g) * (0.5 + 0.25 * i) - 0.3 * i for i in range(8)]
    solo = []
    for x in xs:
        torch.cuda.synchronize(); solo.append(m.run_net(x, nout, ngf).clone()); torch.cuda.synchronize()
    bad = 0; worst = 0.0
    reps = 250 if h < 320 else 100
    for rep in range(reps):
        queued = [m.run_net(x, nout, ngf).clone() for x in xs]
        torch.cuda.synchronize()
        for q, s in zip(queued, solo):
            if not torch.equal(q, s):
                bad += 1; worst = max(worst, float((q - s).abs().max()))
    print("coord" if coord else "wrap ", (b, h, w), "mismatching forwards %d of %d, worst %.2e" % (bad, reps * 8, worst), flush=True)
