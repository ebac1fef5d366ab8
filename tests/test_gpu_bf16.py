"""GPU parity of the bf16 network path (BASELINE configs[2]: bf16 operands, fp32 accumulate) against
the oracle's bf16 emulation (oracle/nets.py forward(bf16=True): operands rounded to bf16 at the same
points, everything else fp32).

Tolerances (stated here because north_star's 1e-3 is the fp32 gate):
  * first-layer raw output (fp32 accumulators of identical bf16 operands): 1e-4 of the layer's scale --
    only the fp32 summation order differs;
  * deeper layers / the tanh prediction: both sides round activations to bf16 (2^-8 relative), and a
    summation-order difference can flip a rounding, so agreement is a few bf16 ulps of the layer scale:
    max-abs <= 4e-2, mean-abs <= 3e-3 on the prediction in [-1, 1];
  * against the fp32 oracle the bf16 path is reported, and bounded loosely (mean-abs <= 2e-2)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    from matryodshka_amd import MSI, nets
    from oracle import nets as onets
    from oracle.msi import MSI as OracleMSI
    return torch, MSI, nets, onets, OracleMSI


@pytest.mark.parametrize("coord", [True, False])
@pytest.mark.parametrize("b,h,w,cin,nout,ngf", [(1, 32, 64, 96, 32, 16), (2, 16, 40, 24, 8, 16), (1, 16, 32, 48, 16, 64)])
def test_bf16_net_matches_bf16_oracle(env, coord, b, h, w, cin, nout, ngf):
    torch, MSI, nets, onets, _ = env
    weights = onets.init_weights(cin, nout, ngf=ngf, coord_net=coord, seed=3, randomize_affine=True)
    rng = np.random.RandomState(4)
    x = onets.bf16_round(rng.uniform(-1, 1, size=(b, h, w, cin)).astype(np.float32))
    m = MSI(weights=weights, coord_net=coord, dtype='bf16')
    pred = m.run_net(torch.from_numpy(x).cuda().bfloat16(), nout, ngf).cpu().numpy()
    ref, acts = onets.forward(weights, x, coord_net=coord, return_activations=True, bf16=True)
    ref32 = onets.forward(weights, x, coord_net=coord)
    from tests.util import read_raw_output
    desc, packed, ws = m._net(b, h, w, cin, nout, ngf)
    infos = nets.layer_infos(desc)
    raw = read_raw_output(ws, packed, infos[0], b, "bf16")
    o = acts["conv1_1/raw"]
    # the raw output is stored as fp16 (11 significand bits): 2^-11 relative to each value, i.e. <= 5e-4 of the layer scale
    assert np.abs(raw - o).max() <= 5e-4 * np.abs(o).max()
    err = np.abs(pred - ref)
    assert err.max() <= 4e-2 and err.mean() <= 3e-3, (err.max(), err.mean())
    assert np.abs(pred - ref32).mean() <= 2e-2


def test_bf16_big_tile_matches_small_tile(env):
    """The 128x128 tile (chosen for large grids) forced on a small problem: same accumulators as the 64x64
    tile up to fp32 summation order, hence the same prediction up to isolated bf16 rounding flips."""
    torch, MSI, nets, onets, _ = env
    b, h, w, cin, nout, ngf = 2, 32, 64, 48, 16, 64
    weights = onets.init_weights(cin, nout, ngf=ngf, coord_net=True, seed=9, randomize_affine=True)
    x = torch.from_numpy(np.random.RandomState(1).uniform(-1, 1, (b, h, w, cin)).astype(np.float32)).cuda().bfloat16()
    from matryodshka_amd import _native as N
    m = MSI(weights=weights, coord_net=True, dtype='bf16')
    m.net_options[N.NET_OPT_BIGTILE] = 0
    small = m.run_net(x, nout, ngf).cpu().numpy()
    m.net_options[N.NET_OPT_BIGTILE] = 2
    big = m.run_net(x, nout, ngf).cpu().numpy()
    ref = onets.forward(weights, x.float().cpu().numpy(), coord_net=True, bf16=True)
    d = np.abs(big - small)
    assert d.max() <= 4e-2 and d.mean() <= 3e-3, (d.max(), d.mean())
    e = np.abs(big - ref)
    assert e.max() <= 4e-2 and e.mean() <= 3e-3, (e.max(), e.mean())


def test_bf16_sweep_is_rounded_fp32_sweep(env):
    torch, MSI, nets, onets, OracleMSI = env
    from tests.util import make_inputs
    b, h, w, d = 1, 32, 64, 8
    inp = make_inputs(11, b, h, w)
    m = MSI(dtype='bf16')
    o = OracleMSI(dtype='bf16')
    planes = m.inv_depths(1.0, 100.0, d)
    ref = m.preprocess_image(torch.from_numpy(inp["ref_image"]))
    src = m.preprocess_image(torch.from_numpy(inp["src_image"]))
    psv = m.format_network_input(ref, src, inp["ref_pose"], inp["src_pose"], planes, inp["intrinsics"])
    assert psv.dtype == torch.bfloat16
    psv_o = o.format_network_input(o.preprocess_image(inp["ref_image"]), o.preprocess_image(inp["src_image"]),
                                   inp["ref_pose"], inp["src_pose"], planes, inp["intrinsics"])
    diff = np.abs(psv.float().cpu().numpy() - psv_o)
    # identical up to fp32 differences that cross a bf16 rounding boundary (one bf16 ulp <= 2^-8 below 1)
    assert diff.max() <= 2.0 ** -7 and (diff > 0).mean() < 2e-3, (diff.max(), (diff > 0).mean())


def test_bf16_pipeline_config3_shapes(env):
    """D = 64 (Cin = 384, 128 head channels), batch 2, reduced image: infer + render through the bf16 path."""
    torch, MSI, nets, onets, OracleMSI = env
    from tests.util import make_inputs
    b, h, w, d, ngf = 2, 32, 64, 64, 16
    inp = make_inputs(21, b, h, w)
    weights = onets.init_weights(6 * d, 2 * d, ngf=ngf, coord_net=True, seed=2, randomize_affine=True)
    m = MSI(weights=weights, coord_net=True, dtype='bf16')
    o = OracleMSI(weights=weights, coord_net=True, dtype='bf16')
    planes = m.inv_depths(1.0, 100.0, d)
    out, net_input = m.infer_msi(torch.from_numpy(inp["src_image"]), torch.from_numpy(inp["ref_image"]), None, None,
                                 inp["ref_pose"], inp["src_pose"], inp["intrinsics"], 'blend_psv', d, planes, ngf=ngf)
    out_o, _ = o.infer_msi(inp["src_image"], inp["ref_image"], None, None, inp["ref_pose"], inp["src_pose"],
                           inp["intrinsics"], 'blend_psv', d, planes, ngf=ngf)
    rgba = out["rgba_layers"].cpu().numpy()
    err = np.abs(rgba - out_o["rgba_layers"])
    assert err.max() <= 4e-2 and err.mean() <= 3e-3, (err.max(), err.mean())
    rgb = m.msi_render_equirect_view(out["rgba_layers"], inp["tgt_pose_rt"], inp["tgt_pos"], planes, inp["intrinsics"])
    rgb_o = o.msi_render_equirect_view(out_o["rgba_layers"], inp["tgt_pose_rt"], inp["tgt_pos"], planes, inp["intrinsics"])
    e2 = np.abs(rgb.cpu().numpy() - rgb_o)
    assert e2.max() <= 6e-2 and e2.mean() <= 3e-3, (e2.max(), e2.mean())
    assert np.isfinite(rgb.cpu().numpy()).all()


def test_bf16_full_size_network_matches_bf16_oracle(env):
    """BASELINE configs[1] shapes (640x320, D = 32, ngf = 64, batch 1) through the bf16 network: the grid sizes at
    which the tail split, the two-level split of the 40x80 layers and the fix-up kernel are active."""
    torch, MSI, nets, onets, _ = env
    cin, nout, ngf = 192, 64, 64
    weights = onets.init_weights(cin, nout, ngf=ngf, coord_net=True, seed=8964, randomize_affine=True)
    rng = np.random.RandomState(5)
    x = onets.bf16_round(rng.uniform(-1, 1, size=(1, 320, 640, cin)).astype(np.float32))
    m = MSI(weights=weights, coord_net=True, dtype='bf16')
    pred = m.run_net(torch.from_numpy(x).cuda().bfloat16(), nout, ngf).cpu().numpy()
    ref = onets.forward(weights, x, coord_net=True, bf16=True)
    err = np.abs(pred - ref)
    assert err.max() <= 6e-2 and err.mean() <= 3e-3, (err.max(), err.mean())


def test_bf16_rejects_unsupported_channels(env):
    torch, MSI, nets, onets, _ = env
    from matryodshka_amd import _native as N
    desc = nets.make_desc(1, 16, 32, 12, 4, 12, True, dtype="bf16")      # 12 channels: not a multiple of 8
    assert N.lib.msi_net_workspace_bytes(desc) == 0 and b"multiples of 8" in N.lib.msi_last_error_string()


# per-layer gates (relative to the layer's max |raw|): measured on MI355X (r02) x ~3; the bf16 network agrees with its
# oracle to fp32 summation order except where a difference flips a bf16 rounding of an activation (one bf16 ulp of one
# operand), so the error grows slowly with depth instead of being "a few percent everywhere"
_BF16_LAYER_GATES = {  # name: (max-abs / scale, mean-abs / scale); measured: conv1_1 3.7e-7 / 2.5e-8 ... conv6_3 5.5e-3 / 8.0e-4
    # (r03: the raw output is stored as fp16, 2^-11 of each value: the floor of the first layers' gates)
    "conv1_1": (5e-4, 1e-4), "conv1_2": (6e-4, 1e-4), "conv2_1": (1e-3, 1e-4), "conv2_2": (3e-3, 1e-4),
    "conv3_1": (5e-3, 2e-4), "conv3_2": (7e-3, 3e-4), "conv3_3": (7e-3, 6e-4), "conv4_1": (9e-3, 1.0e-3),
    "conv4_2": (1.1e-2, 1.4e-3), "conv4_3": (1.1e-2, 1.5e-3), "conv6_1": (1.1e-2, 1.5e-3), "conv6_2": (1.4e-2, 1.7e-3),
    "conv6_3": (1.7e-2, 2.0e-3), "conv7_1": (1.5e-2, 1.7e-3), "conv7_2": (1.6e-2, 2.0e-3), "conv8_1": (1.3e-2, 1.5e-3),
    "conv8_2": (1.2e-2, 1.5e-3),
}


def test_bf16_every_layer_tracks_the_bf16_oracle(env):
    """the bf16 tolerances were only checked on conv1_1 and on the final prediction.  Every layer's raw
    fp32 output (the accumulators the LayerNorm statistics are taken from) against the bf16 oracle, relative to the
    layer's scale, with per-layer gates."""
    torch, MSI, nets, onets, _ = env
    b, h, w, cin, nout, ngf = 1, 64, 128, 96, 32, 32
    weights = onets.init_weights(cin, nout, ngf=ngf, coord_net=True, seed=17, randomize_affine=True)
    x = onets.bf16_round(np.random.RandomState(8).uniform(-1, 1, size=(b, h, w, cin)).astype(np.float32))
    m = MSI(weights=weights, coord_net=True, dtype="bf16")
    pred = m.run_net(torch.from_numpy(x).cuda().bfloat16(), nout, ngf).cpu().numpy()
    ref, acts = onets.forward(weights, x, coord_net=True, return_activations=True, bf16=True)
    from tests.util import read_raw_output
    desc, packed, ws = m._net(b, h, w, cin, nout, ngf)
    report = {}
    for info in nets.layer_infos(desc):
        if info.kind == nets.KIND_HEAD:
            continue
        name = info.name.decode()
        raw = read_raw_output(ws, packed, info, b, "bf16")
        o = acts[name + "/raw"]
        scale = np.abs(o).max()
        err = np.abs(raw - o) / scale
        report[name] = (float(err.max()), float(err.mean()))
    print("bf16 per-layer relative errors (max, mean):", {k: ("%.1e" % v[0], "%.1e" % v[1]) for k, v in report.items()})
    for name, (mx, mn) in report.items():
        gmx, gmn = _BF16_LAYER_GATES[name]
        assert mx <= gmx and mn <= gmn, (name, mx, mn, gmx, gmn)
    e = np.abs(pred - ref)
    assert e.max() <= 4e-2 and e.mean() <= 2e-3, (e.max(), e.mean())


@pytest.mark.parametrize("coord,b,h,w,cin,nout,ngf", [(True, 1, 160, 320, 192, 64, 64), (False, 2, 64, 128, 64, 16, 64),
                                                     (True, 2, 32, 64, 128, 32, 32)])
def test_bf16_halo_patch_kernel_matches_tap_kernel_and_oracle(env, coord, b, h, w, cin, nout, ngf):
    """conv_halo_bf16_kernel (plan option HALO, default on: the stride-1 3x3 layers with 64-channel chunks stage one
    LDS-stationary halo patch per chunk -- from the bf16 operand copy, or from the producer's raw fp32 output with its
    LayerNorm + ReLU + bf16 rounding applied on the way) against the tap-DMA bf16 kernel: identical bf16 operands up to
    isolated rounding flips of an activation, fp32 accumulation in a different order; bitwise deterministic."""
    torch, MSI, nets, onets, _ = env
    from matryodshka_amd import _native as N
    weights = onets.init_weights(cin, nout, ngf=ngf, coord_net=coord, seed=41, randomize_affine=True)
    x = onets.bf16_round(np.random.RandomState(6).uniform(-1, 1, size=(b, h, w, cin)).astype(np.float32))
    xg = torch.from_numpy(x).cuda().bfloat16()
    halo = MSI(weights=weights, coord_net=coord, dtype="bf16")
    tap = MSI(weights=weights, coord_net=coord, dtype="bf16")
    tap.net_options[N.NET_OPT_HALO] = 0
    p1, p0 = halo.run_net(xg, nout, ngf), tap.run_net(xg, nout, ngf)
    plan = halo._plan(b, h, w, cin, nout, ngf)
    raw_layers = [i for i in range(17) if N.lib.msi_net_plan_layer_is_normalized(plan.handle, i) == 0]
    if ngf == 64 and h % 64 == 0:
        assert len(raw_layers) >= 4, raw_layers       # producers whose every consumer is a 128x128 halo layer
    d = (p1 - p0).abs()
    assert float(d.max()) <= 4e-2 and float(d.mean()) <= 3e-3, (float(d.max()), float(d.mean()))   # (as big tile vs small tile)
    for _ in range(3):
        assert torch.equal(halo.run_net(xg, nout, ngf), p1)
    if h * w <= 64 * 128:
        ref = onets.forward(weights, x, coord_net=coord, bf16=True)
        e = np.abs(p1.cpu().numpy() - ref)
        assert e.max() <= 4e-2 and e.mean() <= 3e-3, (e.max(), e.mean())


@pytest.mark.parametrize("coord,b,h,w,cin,nout,ngf", [(True, 1, 160, 320, 192, 64, 64), (False, 2, 64, 128, 64, 16, 64)])
def test_bf16_halo_kernel_eight_and_four_waves_agree(env, coord, b, h, w, cin, nout, ngf):
    """Plan option BF16_WAVES: the 128x128 tile of conv_halo_bf16_kernel as 4 x 2 waves of 32 pixels x 64 channels (default) or
    2 x 2 waves of 64 x 64 (r02): the same bf16 operands and fp32 accumulation per output element; only the partition of the
    LayerNorm sums over waves differs (every wave's share is rounded to one fixed-point unit): isolated bf16 rounding flips of an
    activation, propagated through 17 layers -- the gates of the halo-vs-tap comparison; either shape is bitwise deterministic.
    Anything but 4 / 8 is refused."""
    torch, MSI, nets, onets, _ = env
    from matryodshka_amd import _native as N
    weights = onets.init_weights(cin, nout, ngf=ngf, coord_net=coord, seed=43, randomize_affine=True)
    x = onets.bf16_round(np.random.RandomState(9).uniform(-1, 1, size=(b, h, w, cin)).astype(np.float32))
    xg = torch.from_numpy(x).cuda().bfloat16()
    w8 = MSI(weights=weights, coord_net=coord, dtype="bf16")
    w4 = MSI(weights=weights, coord_net=coord, dtype="bf16")
    w4.net_options[N.NET_OPT_BF16_WAVES] = 4
    p8, p4 = w8.run_net(xg, nout, ngf), w4.run_net(xg, nout, ngf)
    d = (p8 - p4).abs()
    assert float(d.max()) <= 4e-2 and float(d.mean()) <= 3e-3, (float(d.max()), float(d.mean()))
    for _ in range(3):
        assert torch.equal(w8.run_net(xg, nout, ngf), p8)
        assert torch.equal(w4.run_net(xg, nout, ngf), p4)
    bad = MSI(weights=weights, coord_net=coord, dtype="bf16")
    bad.net_options[N.NET_OPT_BF16_WAVES] = 6
    with pytest.raises(Exception):
        bad.run_net(xg, nout, ngf)


@pytest.mark.parametrize("b,h,w,d,ngf", [(2, 32, 64, 8, 16), (1, 64, 128, 64, 64), (1, 32, 64, 32, 32), (1, 16, 32, 48, 16)])
def test_bf16_fused_tail_matches_two_step_path(env, b, h, w, d, ngf):
    """msi_net_plan_forward_rgba on a bf16 plan (1x1 head on the fp32 MFMA over the bf16-rounded operands + conv8_2's
    LayerNorm + the RGBA assembly from the bf16 sweep volume in one kernel) against run_net + assemble_layers: the same
    operands, only the head's fp32 summation order differs."""
    torch, MSI, nets, onets, _ = env
    from matryodshka_amd import _native as N
    cin = 6 * d
    weights = onets.init_weights(cin, 2 * d, ngf=ngf, coord_net=True, seed=77, randomize_affine=True)
    x = torch.from_numpy(onets.bf16_round(np.random.RandomState(2).uniform(-1, 1, size=(b, h, w, cin)).astype(np.float32))).cuda().bfloat16()
    fused = MSI(weights=weights, coord_net=True, dtype="bf16")
    two = MSI(weights=weights, coord_net=True, dtype="bf16")
    two.net_options[N.NET_OPT_HEAD_FUSE_LN] = 0
    extra = "blend_weights alpha"
    pf = fused.infer_layers(x, d, ngf, extra_outputs=extra)
    pt = two.infer_layers(x, d, ngf, extra_outputs=extra)
    plan = fused._plan(b, h, w, cin, 2 * d, ngf)
    for k in ("rgba_layers", "blend_weights", "alphas"):
        e = float((pf[k] - pt[k]).abs().max())
        assert e <= 2e-6, (k, e)
    assert torch.equal(fused.infer_layers(x, d, ngf)["rgba_layers"], pf["rgba_layers"])
