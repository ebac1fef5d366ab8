"""Plan option X3_TILE8 (round 5): the stride-1, rate-1 layers of the six-product split on 8 x 16-pixel tiles (conv_halo8_x3_kernel: two accumulators per wave
sharing the weight fragments).  Same arithmetic and per-accumulator summation order as the 4-row tile: whole tiles are bit-identical to the default plan, tiles cut
into K-ranges differ by summation order only; every gate of the fp32 path applies (layers within 2e-4 of their scale, tanh output within 1e-3 of the oracle),
bitwise deterministic, fix-up launch == in-launch hand-off."""
import numpy as np
import pytest

from tests.test_gpu_cnn import _run, env  # noqa: F401

pytestmark = pytest.mark.gpu
ALL = 0x3ffff | (1 << 30)   # every eligible layer, whatever its grid size


@pytest.mark.parametrize("coord", [True, False])
@pytest.mark.parametrize("b,h,w,cin,nout,ngf", [(1, 32, 64, 96, 32, 32), (2, 16, 48, 32, 8, 32), (1, 64, 128, 192, 64, 64), (1, 160, 320, 96, 32, 64)])
def test_tile8_layers_match_the_oracle_and_the_4row_tile(env, coord, b, h, w, cin, nout, ngf):
    torch, MSI, nets, N, onets = env
    pred, ref, raws, acts = _run(env, b, h, w, cin, nout, ngf, coord, seed=3, options={N.NET_OPT_X3_TILE8: ALL})
    base, _, raws_b, _ = _run(env, b, h, w, cin, nout, ngf, coord, seed=3, options={N.NET_OPT_X3_TILE8: 0})
    m = MSI(weights=onets.init_weights(cin, nout, ngf=ngf, coord_net=coord, seed=3, randomize_affine=True), coord_net=coord)
    m.net_options[N.NET_OPT_X3_TILE8] = ALL
    plan = m._plan(b, h, w, cin, nout, ngf)
    kern = [plan.layer_kernel(i) for i in range(17)]
    wide = [i for i in range(17) if "conv_halo8_x3_kernel" in kern[i][0]]
    assert len(wide) >= (2 if h % 64 == 0 or h == 160 else 1), kern            # (the 8-row tile really ran: conv1_1 / conv8_2 at least where H % 8 == 0)
    for name, raw in raws.items():
        o = acts[name]
        err = np.abs(raw - o).max() / (np.abs(o).max() + 1e-12)
        assert err < 2e-4, "%s: relative max err %g" % (name, err)
        d = np.abs(raw - raws_b[name]).max() / (np.abs(o).max() + 1e-12)
        assert d < 2e-5, "%s: 8-row vs 4-row tile %g" % (name, d)
    e8, e4 = np.abs(pred - ref).max(), np.abs(base - ref).max()
    print("tile8 vs oracle %.2e | 4-row tile vs oracle %.2e | tile8 vs 4-row %.2e | layers on the 8-row tile: %s" % (e8, e4, np.abs(pred - base).max(), wide))
    assert e8 <= 1e-3 and e8 <= 2 * e4 + 2e-6


def test_tile8_is_deterministic_and_fixup_launch_agrees(env):
    torch, MSI, nets, N, onets = env
    b, h, w, cin, nout, ngf = 1, 160, 320, 96, 32, 64          # 80 x 160 and 40 x 80 layers: 8-row tiles cut into K-ranges
    weights = onets.init_weights(cin, nout, ngf=ngf, coord_net=True, seed=9, randomize_affine=True)
    x = torch.rand((b, h, w, cin), device="cuda") * 2 - 1
    m = MSI(weights=weights, coord_net=True)
    m.net_options[N.NET_OPT_X3_TILE8] = ALL
    first = m.run_net(x, nout, ngf).clone()
    for _ in range(20):
        assert torch.equal(m.run_net(x, nout, ngf), first)
    assert m.network_status() == 0
    plan = m._plan(b, h, w, cin, nout, ngf)
    assert any(plan.layer_kernel(i)[2] > 0 and "conv_halo8_x3_kernel" in plan.layer_kernel(i)[0] for i in range(17)), plan.kernels()
    f = MSI(weights=weights, coord_net=True)
    f.net_options[N.NET_OPT_X3_TILE8] = ALL
    f.net_options[N.NET_OPT_FIXUP_KERNEL] = 1
    assert torch.equal(f.run_net(x, nout, ngf), first)
