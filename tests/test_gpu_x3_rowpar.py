"""Plan option X3_ROWPAR (round 5): the rate-2 layers of the split kernels on ROW-PARITY tiles (conv_halo_x3_kernel<3, ...>: a tile's four rows are every other image
row, so the dilation along H is the tile's own row stride -- 6 x 20-pixel patch, two-stage weight ring, three workgroups per CU).  Same arithmetic and summation order
per output element as the plain rate-2 tile: a layer of whole tiles fed identical inputs is BIT-identical to it (K-range tiles differ by summation order, later layers
through the per-wave partial sums of the LayerNorm statistics); every gate of the fp32 path applies."""
import numpy as np
import pytest

from tests.test_gpu_cnn import _run, env  # noqa: F401

pytestmark = pytest.mark.gpu
ALL = 0x3ffff


@pytest.mark.parametrize("coord", [True, False])
@pytest.mark.parametrize("split_f16", [0, ALL])
@pytest.mark.parametrize("b,h,w,cin,nout,ngf", [(1, 64, 128, 96, 32, 32), (2, 64, 128, 32, 8, 32), (1, 128, 128, 192, 64, 64), (1, 320, 640, 96, 32, 32)])
def test_row_parity_tiles_match_the_oracle_and_the_plain_rate2_tile(env, coord, split_f16, b, h, w, cin, nout, ngf):
    torch, MSI, nets, N, onets = env
    if split_f16 and (h, w) != (64, 128):
        pytest.skip("the fp16 form: one size")
    opts = {N.NET_OPT_F32_SPLIT_F16: split_f16}
    pred, ref, raws, acts = _run(env, b, h, w, cin, nout, ngf, coord, seed=5, options={**opts, N.NET_OPT_X3_ROWPAR: ALL})
    base, _, raws_b, _ = _run(env, b, h, w, cin, nout, ngf, coord, seed=5, options={**opts, N.NET_OPT_X3_ROWPAR: 0})
    m = MSI(weights=onets.init_weights(cin, nout, ngf=ngf, coord_net=coord, seed=5, randomize_affine=True), coord_net=coord)
    m.net_options[N.NET_OPT_F32_SPLIT_F16] = split_f16
    plan = m._plan(b, h, w, cin, nout, ngf)                      # (the default: row-parity tiles on)
    kern = [plan.layer_kernel(i) for i in range(17)]
    par = [i for i in range(17) if "conv_halo_x3_kernel<3," in kern[i][0]]
    assert par == [7, 8, 9], kern                                 # conv4_1 .. conv4_3 (their input: H / 8 x W / 8, a multiple of 8 rows here)
    for name, raw in raws.items():
        o = acts[name]
        err = np.abs(raw - o).max() / (np.abs(o).max() + 1e-12)
        assert err < (6e-4 if split_f16 else 2e-4), "%s: relative max err %g" % (name, err)
        d = np.abs(raw - raws_b[name]).max() / (np.abs(o).max() + 1e-12)
        assert d < 2e-5, "%s: row-parity vs plain rate-2 tile %g" % (name, d)
    whole = all(kern[i][2] == 0 for i in par)                     # no tile of the three layers is cut into K-ranges
    if whole:
        # conv4_1: identical inputs, the same products in the same order -> bit-identical output.  (conv4_2 / conv4_3 see conv4_1's LayerNorm statistics, whose per-wave
        # float partial sums group different pixels per wave: equal to ~1e-7, asserted above)
        assert np.array_equal(raws["conv4_1"], raws_b["conv4_1"])
    e1, e0 = np.abs(pred - ref).max(), np.abs(base - ref).max()
    print("row-parity vs oracle %.2e | plain vs oracle %.2e | row-parity vs plain %.2e | whole tiles: %s" % (e1, e0, np.abs(pred - base).max(), whole))
    assert e1 <= 1e-3 and e1 <= 2 * e0 + 2e-6


def test_row_parity_is_deterministic_and_fixup_launch_agrees(env):
    torch, MSI, nets, N, onets = env
    b, h, w, cin, nout, ngf = 1, 320, 640, 96, 32, 64             # conv4_x at 40 x 80: 400 tiles cut into K-ranges (configs[1]'s grid)
    weights = onets.init_weights(cin, nout, ngf=ngf, coord_net=True, seed=9, randomize_affine=True)
    x = torch.rand((b, h, w, cin), device="cuda") * 2 - 1
    m = MSI(weights=weights, coord_net=True)
    first = m.run_net(x, nout, ngf).clone()
    for _ in range(20):
        assert torch.equal(m.run_net(x, nout, ngf), first)
    assert m.network_status() == 0
    plan = m._plan(b, h, w, cin, nout, ngf)
    assert any(plan.layer_kernel(i)[2] > 0 and "conv_halo_x3_kernel<3," in plan.layer_kernel(i)[0] for i in range(17)), plan.kernels()
    f = MSI(weights=weights, coord_net=True)
    f.net_options[N.NET_OPT_FIXUP_KERNEL] = 1
    assert torch.equal(f.run_net(x, nout, ngf), first)
    # a height whose eighth is NOT a multiple of 8 rows keeps the plain tile (the option is a request, the plan decides)
    g = MSI(weights=weights, coord_net=True)
    pl = g._plan(1, 96, 128, cin, nout, ngf)                      # conv4_x input: 12 x 16
    assert not any("conv_halo_x3_kernel<3," in pl.layer_kernel(i)[0] for i in range(17)), pl.kernels()
