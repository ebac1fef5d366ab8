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


def test_random_sweep_shapes_on_the_lds_staged_kernel_match_oracle_and_the_gather_kernel():
    """Round 6: a fixed-seed slice over shapes that take ods_sweep_lds_kernel (batch >= 2, (W * D / 2) % 256 == 0): random sizes, sphere counts, baselines (up to 10 x the
    usual one: wide polar boxes, fallback blocks), far planes, identity and small rigid source poses -- the volume against the oracle (identity poses: max; posed: 99.9th
    percentile, isolated branch flips allowed as in tests/test_gpu_geometry.py) and, bit for bit, against each frame swept alone (batch 1: the gather kernel)."""
    import torch
    from matryodshka_amd import MSI
    from oracle.msi import MSI as OracleMSI
    from matryodshka_amd.synthetic import make_inputs
    rng = np.random.RandomState(606)
    m, o = MSI(), OracleMSI()
    for it in range(10):
        b = int(rng.choice([2, 3, 5]))
        d = int(rng.choice([16, 32, 64]))
        h, w = 2 * int(rng.randint(4, 24)), 32 * int(rng.randint(1, 5))
        assert (w * (d // 2)) % 256 == 0
        inp = make_inputs(int(rng.randint(1 << 30)), b, h, w)
        inp["intrinsics"][:, 0, 0] = rng.uniform(0.01, 0.3)
        posed = rng.rand() < 0.5
        if posed:
            th = rng.uniform(-0.05, 0.05)
            p = np.eye(4, dtype=np.float32)
            p[0, 0], p[0, 2], p[2, 0], p[2, 2] = np.cos(th), np.sin(th), -np.sin(th), np.cos(th)
            p[:3, 3] = rng.uniform(-0.02, 0.02, 3)
            inp["src_pose"] = np.tile(p[None], (b, 1, 1))
        planes = m.inv_depths(1.0, float(rng.uniform(20, 100)), d)
        ref, src = m.preprocess_image(torch.from_numpy(inp["ref_image"])), m.preprocess_image(torch.from_numpy(inp["src_image"]))
        psv_t = m.format_network_input(ref, src, inp["ref_pose"], inp["src_pose"], planes, inp["intrinsics"])
        for k in range(b):
            one = m.format_network_input(ref[k:k + 1], src[k:k + 1], inp["ref_pose"][k:k + 1], inp["src_pose"][k:k + 1], planes, inp["intrinsics"][k:k + 1])
            assert torch.equal(one[0], psv_t[k]), (it, k, b, h, w, d)
        psv_o = o.format_network_input(o.preprocess_image(inp["ref_image"]), o.preprocess_image(inp["src_image"]),
                                       inp["ref_pose"], inp["src_pose"], planes, inp["intrinsics"])
        e = np.abs(psv_t.cpu().numpy() - psv_o)
        g = float(np.percentile(e, 99.9)) if posed else float(e.max())
        assert g <= 1e-3, (it, b, h, w, d, posed, g)
