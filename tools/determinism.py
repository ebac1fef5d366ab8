import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from matryodshka_amd import MSI, nets
for dtype in ("f32", "bf16"):
    m = MSI(weights=nets.init_weights(192, 64, 64, True), coord_net=True, dtype=dtype)
    x = torch.rand((1, 320, 640, 192), device="cuda") * 2 - 1
    if dtype == "bf16":
        x = x.bfloat16()
    ref = m.run_net(x, 64, 64).clone()
    bad = 0
    for i in range(20):
        y = m.run_net(x, 64, 64)
        if not torch.equal(y, ref):
            bad += 1
    print(dtype, "bitwise-identical over 20 runs:", bad == 0, "finite:", bool(torch.isfinite(ref).all()))
