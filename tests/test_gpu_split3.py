"""Plan options F32_SPLIT3 (default) and F32_SPLIT_F16 (opt-in: the same kernels on a 2-way fp16 split with three products; every test below runs both).
F32_SPLIT3 (round 4): the stride-1 halo-patch layers of an fp32 plan compute their convolution as a 3-way bf16 split
of both operands with SIX products on the bf16 MFMA (conv_halo_x3_kernel): fp32-grade arithmetic -- the dropped terms are below
2^-26 of a product -- so every gate of the native fp32 path applies unchanged: layers within 2e-4 of their scale, the tanh output
within 1e-3, the full-size fixtures within 1e-3, bitwise determinism, the fix-up launch equal to the in-launch hand-off."""
import numpy as np
import pytest

from tests.test_gpu_cnn import _run, env  # noqa: F401

pytestmark = pytest.mark.gpu
ALL = 0x3ffff
# the two split arithmetics: six bf16 products (F32_SPLIT_F16 = 0) and three fp16 products of 22-bit operands (F32_SPLIT_F16 = ALL)
ARITH = [pytest.param(0, id="bf16x6"), pytest.param(ALL, id="f16x3")]


@pytest.mark.parametrize("f16", ARITH)
@pytest.mark.parametrize("coord", [True, False])
@pytest.mark.parametrize("b,h,w,cin,nout,ngf", [(1, 32, 64, 96, 32, 32), (2, 16, 48, 32, 8, 32), (1, 64, 128, 192, 64, 64)])
def test_split3_layers_match_the_oracle(env, coord, b, h, w, cin, nout, ngf, f16):
    torch, MSI, nets, N, onets = env
    pred, ref, raws, acts = _run(env, b, h, w, cin, nout, ngf, coord, seed=3, options={N.NET_OPT_F32_SPLIT3: ALL, N.NET_OPT_F32_SPLIT_F16: f16})
    native, _, raws_n, _ = _run(env, b, h, w, cin, nout, ngf, coord, seed=3, options={N.NET_OPT_F32_SPLIT3: 0})
    m = MSI(weights=onets.init_weights(cin, nout, ngf=ngf, coord_net=coord, seed=3, randomize_affine=True), coord_net=coord)
    m.net_options[N.NET_OPT_F32_SPLIT3] = ALL
    m.net_options[N.NET_OPT_F32_SPLIT_F16] = f16
    kern = [m._plan(b, h, w, cin, nout, ngf).layer_kernel(i)[0] for i in range(17)]
    assert sum("_x3_kernel" in k for k in kern) >= (8 if w % 64 == 0 else 2), kern      # (the split path really ran)
    if f16:
        assert sum(k.endswith(", 2>") or k.endswith("<2>") for k in kern) >= (8 if w % 64 == 0 else 2), kern  # (... in its fp16 form)
    worst = 0.0
    for name, raw in raws.items():
        o = acts[name]
        err = np.abs(raw - o).max() / (np.abs(o).max() + 1e-12)
        worst = max(worst, err)
        assert err < 2e-4, "%s: relative max err %g" % (name, err)
    e_split, e_native = np.abs(pred - ref).max(), np.abs(native - ref).max()
    print("split3 vs oracle %.2e | native fp32 vs oracle %.2e | split3 vs native %.2e | worst layer %.2e" % (
        e_split, e_native, np.abs(pred - native).max(), worst))
    assert e_split <= 1e-3 and e_split <= 2 * e_native + 2e-6


@pytest.mark.parametrize("f16", ARITH)
def test_split3_is_deterministic_and_fixup_launch_agrees(env, f16):
    torch, MSI, nets, N, onets = env
    b, h, w, cin, nout, ngf = 1, 160, 320, 96, 32, 64          # 40 x 80 deepest layers: tiles cut into K-ranges
    weights = onets.init_weights(cin, nout, ngf=ngf, coord_net=True, seed=9, randomize_affine=True)
    x = torch.rand((b, h, w, cin), device="cuda") * 2 - 1
    m = MSI(weights=weights, coord_net=True)
    m.net_options[N.NET_OPT_F32_SPLIT3] = ALL
    m.net_options[N.NET_OPT_F32_SPLIT_F16] = f16
    first = m.run_net(x, nout, ngf).clone()
    for _ in range(10):
        assert torch.equal(m.run_net(x, nout, ngf), first)
    assert m.network_status() == 0
    plan = m._plan(b, h, w, cin, nout, ngf)
    assert any(plan.layer_kernel(i)[2] > 0 and "_x3_kernel" in plan.layer_kernel(i)[0] for i in range(17))
    f = MSI(weights=weights, coord_net=True)
    f.net_options[N.NET_OPT_F32_SPLIT3] = ALL
    f.net_options[N.NET_OPT_F32_SPLIT_F16] = f16
    f.net_options[N.NET_OPT_FIXUP_KERNEL] = 1
    assert torch.equal(f.run_net(x, nout, ngf), first)


@pytest.mark.parametrize("fixture", ["full_640x320x32_samples.npz", "full_640x320x32_wrap_samples.npz"])
def test_split3_full_size_fixtures(fixture):
    """The configs[1] frame (CoordNet) and the reference's default network (wrap padding) at 640x320x32, ngf 64, through
    the split path: the same dense oracle samples and the same 1e-3 gate as the native path; max-abs reported beside it."""
    import torch
    from matryodshka_amd import MSI, _native as N
    from matryodshka_amd.synthetic import make_inputs
    from oracle import nets as onets
    from tests.test_golden import _fixture, _check_dense
    z, cfg = _fixture(fixture)
    d, ngf, coord, seed = int(cfg["d"]), int(cfg["ngf"]), bool(cfg["coord"]), int(cfg["seed"])
    inp = make_inputs(seed, int(cfg["b"]), int(cfg["h"]), int(cfg["w"]))
    weights = onets.init_weights(6 * d, 2 * d, ngf=ngf, coord_net=coord, seed=seed, randomize_affine=True)
    errs = {}
    for tag, opt, f16 in (("native", 0, 0), ("split3", ALL, 0), ("split_f16", ALL, ALL)):
        m = MSI(weights=weights, coord_net=coord)
        m.net_options[N.NET_OPT_F32_SPLIT3] = opt
        m.net_options[N.NET_OPT_F32_SPLIT_F16] = f16
        planes = m.inv_depths(1.0, 100.0, d)
        pred, net_input = m.infer_msi(torch.from_numpy(inp["src_image"]), torch.from_numpy(inp["ref_image"]), None, None,
                                      inp["ref_pose"], inp["src_pose"], inp["intrinsics"], "blend_psv", d, planes, ngf=ngf)
        rgb, dep = m.msi_render_equirect_view_and_depth(pred["rgba_layers"], inp["tgt_pose_rt"], inp["tgt_pos"], planes, inp["intrinsics"])
        assert m.network_status() == 0
        got = dict(rgba_layers=pred["rgba_layers"].cpu().numpy(), rgb=rgb.cpu().numpy(), depth=dep.cpu().numpy())
        errs[tag] = _check_dense(z, got, ("rgba_layers", "rgb", "depth"), 1e-3)
    print(fixture, {t: {k: "%.2e" % v[0] for k, v in e.items()} for t, e in errs.items()})
    for k in ("rgba_layers", "rgb", "depth"):
        for tag in ("split3", "split_f16"):
            assert errs[tag][k][0] <= 2 * errs["native"][k][0] + 1e-5, (tag, k, errs[tag][k], errs["native"][k])


def test_split_f16_flags_operands_beyond_the_fp16_range(env):
    """The fp16 form needs |operand| <= 65504: an input beyond that poisons the layer (h = inf) -- and says so in the status word
    (MSI_NET_STATUS_F16_SPLIT_RANGE); the six-product bf16 form (fp32's exponent range) digests the same input silently and correctly."""
    torch, MSI, nets, N, onets = env
    b, h, w, cin, nout, ngf = 1, 32, 64, 96, 32, 32
    weights = onets.init_weights(cin, nout, ngf=ngf, coord_net=True, seed=5, randomize_affine=True)
    x = torch.rand((b, h, w, cin), device="cuda") * 2 - 1
    x[0, 7, 9, 3] = 7.0e4
    m = MSI(weights=weights, coord_net=True)
    m.net_options[N.NET_OPT_F32_SPLIT3] = ALL
    m.net_options[N.NET_OPT_F32_SPLIT_F16] = ALL
    m.run_net(x, nout, ngf)
    with pytest.raises(N.MsiError) as ei:
        m.network_status()
    assert "fp16 range" in str(ei.value)
    m6 = MSI(weights=weights, coord_net=True)
    m6.net_options[N.NET_OPT_F32_SPLIT3] = ALL
    m6.net_options[N.NET_OPT_F32_SPLIT_F16] = 0
    y6 = m6.run_net(x, nout, ngf)
    assert bool(torch.isfinite(y6).all())


@pytest.mark.parametrize("f16", ARITH)
@pytest.mark.parametrize("coord,b,h,w,cin,nout,ngf", [(True, 1, 160, 320, 96, 32, 64), (False, 4, 128, 256, 48, 16, 64), (True, 2, 64, 128, 96, 32, 64)])
def test_different_frames_back_to_back_equal_their_solo_runs(env, coord, b, h, w, cin, nout, ngf, f16):
    """Repeating ONE input cannot show a stale hand-off (a K-range slab, a ticket, a LayerNorm shard left over from the previous forward
    holds the very values the next one would write): eight DIFFERENT inputs are queued back to back without a host sync in between and
    every prediction must equal, bit for bit, the one the same input gives when it runs alone on an idle device."""
    torch, MSI, nets, N, onets = env
    weights = onets.init_weights(cin, nout, ngf=ngf, coord_net=coord, seed=41, randomize_affine=True)
    m = MSI(weights=weights, coord_net=coord)
    m.net_options[N.NET_OPT_F32_SPLIT_F16] = f16
    g = torch.Generator(device="cuda").manual_seed(7)
    xs = [torch.rand((b, h, w, cin), device="cuda", generator=g) * (0.5 + 0.25 * i) - 0.3 * i for i in range(8)]
    solo = []
    for x in xs:
        torch.cuda.synchronize()
        solo.append(m.run_net(x, nout, ngf).clone())
        torch.cuda.synchronize()
    assert m.network_status() == 0
    for rep in range(6):
        queued = [m.run_net(x, nout, ngf).clone() for x in xs]          # (clone: run_net returns the model's output buffer)
        torch.cuda.synchronize()
        for i, (q, s) in enumerate(zip(queued, solo)):
            assert torch.equal(q, s), (rep, i, float((q - s).abs().max()))
