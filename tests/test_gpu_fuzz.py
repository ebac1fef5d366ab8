"""Randomised GPU parity of the network against the oracle (a fixed-seed slice of tools/fuzz_parity.py):
random batch / image size / channel counts / CoordNet / dtype within the supported set -- M-tile tails,
channel tails (Cin, ngf not multiples of 32), N tails, widths at which the column border classes overlap."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_random_network_shapes_match_oracle():
    import torch
    from matryodshka_amd import MSI
    from oracle import nets as onets
    rng = np.random.RandomState(2024)
    for it in range(14):
        dtype = "bf16" if it % 4 == 3 else "f32"
        q = 8 if dtype == "bf16" else 4
        b = int(rng.choice([1, 2, 3]))
        h, w = 8 * int(rng.randint(1, 7)), 8 * int(rng.randint(1, 10))
        cin, nout, ngf = q * int(rng.randint(1, 11)), 4 * int(rng.randint(1, 9)), q * int(rng.randint(1, 5))
        coord = bool(rng.rand() < 0.6)
        if not coord:            # wrap_pad(x, 2, 2) at 1/8 resolution needs two columns / rows
            h, w = max(h, 16), max(w, 16)
        weights = onets.init_weights(cin, nout, ngf=ngf, coord_net=coord, seed=int(rng.randint(1 << 30)), randomize_affine=True)
        x = rng.uniform(-1, 1, size=(b, h, w, cin)).astype(np.float32)
        if dtype == "bf16":
            x = onets.bf16_round(x)
        m = MSI(weights=weights, coord_net=coord, dtype=dtype)
        xt = torch.from_numpy(x).cuda()
        pred = m.run_net(xt.bfloat16() if dtype == "bf16" else xt, nout, ngf).cpu().numpy()
        ref = onets.forward(weights, x, coord_net=coord, bf16=dtype == "bf16")
        err = np.abs(pred - ref).max()
        assert np.isfinite(pred).all() and err <= (6e-2 if dtype == "bf16" else 1e-3), (it, dtype, b, h, w, cin, nout, ngf, coord, err)
