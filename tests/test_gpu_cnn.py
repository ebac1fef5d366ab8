"""GPU parity of K2 (implicit-GEMM fp32-MFMA CNN) against the torch-CPU oracle,
layer by layer (the LayerNorm+ReLU'd activations the kernels leave in the workspace)
and end to end.  Tolerance 1e-3 max-abs on the tanh output (north_star); layer
activations are compared relative to their own scale (fp32 summation-order
differences only)."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
_HEAD_FUSED = True   # default plan: the fp32 head applies conv8_2's LayerNorm itself (conv8_2's buffer stays raw)


@pytest.fixture(scope="module")
def env():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    from matryodshka_amd import MSI, nets, _native
    from oracle import nets as onets
    return torch, MSI, nets, _native, onets


def _run(env, b, h, w, cin, nout, ngf, coord, seed=0, options=None):
    torch, MSI, nets, N, onets = env
    weights = onets.init_weights(cin, nout, ngf=ngf, coord_net=coord, seed=seed, randomize_affine=True)
    rng = np.random.RandomState(seed + 1)
    x = rng.uniform(-1, 1, size=(b, h, w, cin)).astype(np.float32)
    m = MSI(weights=weights, coord_net=coord)
    m.net_options.update(options or {})
    pred = m.run_net(torch.from_numpy(x).cuda(), nout, ngf)
    torch.cuda.synchronize()
    ref, acts = onets.forward(weights, x, coord_net=coord, return_activations=True)
    desc, _, ws = m._net(b, h, w, cin, nout, ngf)
    plan = m._plan(b, h, w, cin, nout, ngf)
    raws = {}
    for li, info in enumerate(nets.layer_infos(desc)):
        if info.kind == nets.KIND_HEAD:
            continue
        n = b * info.out_h * info.out_w * info.cout
        raw = ws[info.raw_offset:info.raw_offset + 4 * n].view(torch.float32).reshape(b, info.out_h, info.out_w, info.cout)
        name = info.name.decode()
        raws[name] = raw.cpu().numpy()
        # a layer whose every consumer applies its LayerNorm while loading (the head, halo-patch layers) stays RAW in memory
        if N.lib.msi_net_plan_layer_is_normalized(plan.handle, li) == 0:
            acts[name] = acts[name + "/raw"]
    return pred.cpu().numpy(), ref, raws, acts


@pytest.mark.parametrize("coord", [True, False])
@pytest.mark.parametrize("b,h,w,cin,nout,ngf", [(1, 32, 64, 96, 32, 16), (2, 16, 40, 24, 8, 16), (1, 24, 48, 12, 4, 12)])
def test_net_matches_oracle(env, coord, b, h, w, cin, nout, ngf):
    pred, ref, raws, acts = _run(env, b, h, w, cin, nout, ngf, coord)
    for name, raw in raws.items():
        o = acts[name]          # (acts[name] is the raw output where the plan leaves the buffer raw, see _run)
        assert raw.shape == o.shape, name
        scale = np.abs(o).max() + 1e-12
        err = np.abs(raw - o).max() / scale
        assert err < 2e-4, "%s: relative max err %g" % (name, err)
    assert pred.shape == ref.shape
    assert np.abs(pred - ref).max() <= 1e-3, np.abs(pred - ref).max()


def test_net_reference_width_channels(env):
    """ngf=64 (the reference's width) on a small image: exercises the 128-wide N tiles,
    the two-source skip concat at 1024/512/256 channels and Cout=512 layers."""
    pred, ref, raws, acts = _run(env, 1, 16, 32, 48, 16, 64, True, seed=5)
    for name, raw in raws.items():
        o = acts[name]
        err = np.abs(raw - o).max() / (np.abs(o).max() + 1e-12)
        assert err < 2e-4, "%s: relative max err %g" % (name, err)
    assert np.abs(pred - ref).max() <= 1e-3


def test_workspace_too_small_is_an_error(env):
    torch, MSI, nets, N, onets = env
    desc = nets.make_desc(1, 16, 32, 24, 8, 16, True)
    packed = torch.zeros(N.lib.msi_net_packed_floats(desc), device="cuda")
    x = torch.zeros((1, 16, 32, 24), device="cuda")
    y = torch.zeros((1, 16, 32, 8), device="cuda")
    ws = torch.zeros(1024, dtype=torch.uint8, device="cuda")
    rc = N.lib.msi_net_forward_f32(desc, packed.data_ptr(), x.data_ptr(), y.data_ptr(), ws.data_ptr(), ws.numel(), None)
    assert rc == -4 and b"workspace" in N.lib.msi_last_error_string()


def test_unfused_head_and_odd_output_counts(env):
    """Plan option HEAD_FUSE_LN = 0 (separate LayerNorm pass before the head) and head widths that are not a
    multiple of 4 (which_color_pred blend_bg: 2 D + 3, blend_bg_psv: 3 D + 3 -> scalar store / statistics tails)."""
    torch, MSI, nets, N, onets = env
    pred, ref, raws, acts = _run(env, 1, 16, 32, 24, 11, 16, True, seed=3, options={N.NET_OPT_HEAD_FUSE_LN: 0})
    o = acts["conv8_2"]
    assert np.abs(raws["conv8_2"] - o).max() / np.abs(o).max() < 2e-4
    assert np.abs(pred - ref).max() <= 1e-3
    pred, ref, _, _ = _run(env, 2, 16, 32, 24, 15, 16, False, seed=4)
    assert np.abs(pred - ref).max() <= 1e-3


def test_layernorm_affine_matches_fp64_statistics(env):
    """The published per-channel affine (scale | shift) of every layer against fp64 two-pass statistics of the
    oracle's raw activations -- including msi_train_net's conv-transposes, whose statistics run over the uncropped
    (2H+10) x (2W+10) output (nets.py:423-435)."""
    torch, MSI, nets, N, onets = env
    for coord in (True, False):
        b, h, w, cin, nout, ngf = 2, 16, 40, 24, 8, 16
        weights = onets.init_weights(cin, nout, ngf=ngf, coord_net=coord, seed=6, randomize_affine=True)
        x = np.random.RandomState(2).uniform(-1, 1, size=(b, h, w, cin)).astype(np.float32)
        m = MSI(weights=weights, coord_net=coord)
        m.net_options[N.NET_OPT_HEAD_FUSE_LN] = 0
        m.net_options[N.NET_OPT_HALO] = 0           # (every layer normalised by its own ln_apply launch, which publishes the affine)
        m.run_net(torch.from_numpy(x).cuda(), nout, ngf)
        torch.cuda.synchronize()
        _, acts = onets.forward(weights, x, coord_net=coord, return_activations=True)
        desc, _, ws = m._net(b, h, w, cin, nout, ngf)
        for info in nets.layer_infos(desc):
            if info.kind == nets.KIND_HEAD:
                continue
            name = info.name.decode()
            aff = ws[info.affine_offset:info.affine_offset + 4 * b * 2 * info.cout].view(torch.float32).reshape(b, 2, info.cout).cpu().numpy()
            exp = acts[name + "/affine"]
            tol = 2e-4 * np.abs(exp).max() + 1e-7
            assert np.abs(aff - exp).max() <= tol, (name, np.abs(aff - exp).max(), tol)


def test_forward_is_bitwise_deterministic(env):
    """Fixed summation orders in the fix-up (ascending k whoever arrives last) and exact integer accumulation of the
    LayerNorm sums: the same input gives the same bits, at the grid sizes where the tail split is active."""
    torch, MSI, nets, N, onets = env
    m = MSI(weights=nets.init_weights(96, 32, 64, True), coord_net=True)
    x = torch.rand((1, 160, 320, 96), device="cuda") * 2 - 1
    ref = m.run_net(x, 32, 64).clone()
    for _ in range(3):
        assert torch.equal(m.run_net(x, 32, 64), ref)


@pytest.mark.parametrize("dtype,batch", [("f32", 1), ("f32", 2), ("bf16", 1)])
def test_full_size_split_k_handoff_is_deterministic(env, dtype, batch):
    """BASELINE configs[1] shapes (640x320, 192 -> 64 channels, ngf 64): >= 20 repeats are bitwise identical, and the
    in-launch split-K hand-off (sc1 slabs + relaxed ticket, last arriver sums) equals the separate fix-up launch
    (plan option FIXUP_KERNEL) bit for bit -- same slabs, same ascending-k order, same epilogue."""
    torch, MSI, nets, N, onets = env
    weights = nets.init_weights(192, 64, 64, True)
    m = MSI(weights=weights, coord_net=True, dtype=dtype)
    x = torch.rand((batch, 320, 640, 192), device="cuda") * 2 - 1
    if dtype == "bf16":
        x = x.bfloat16()
    ref = m.run_net(x, 64, 64).clone()
    assert bool(torch.isfinite(ref).all())
    for _ in range(20):
        assert torch.equal(m.run_net(x, 64, 64), ref)
    alt = MSI(weights=weights, coord_net=True, dtype=dtype)
    alt.net_options[N.NET_OPT_FIXUP_KERNEL] = 1
    assert torch.equal(alt.run_net(x, 64, 64), ref)
    nosplit = MSI(weights=weights, coord_net=True, dtype=dtype)
    nosplit.net_options[N.NET_OPT_TAILSPLIT] = 0
    tol = 2e-5 if dtype == "f32" else 4e-2      # unsplit tiles: another fp32 summation order only
    assert float((nosplit.run_net(x, 64, 64) - ref).abs().max()) <= tol


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("coord,b,h,w,cin,nout,ngf", [(True, 1, 160, 320, 96, 32, 64), (False, 2, 32, 64, 24, 8, 16),
                                                     (True, 3, 16, 40, 24, 8, 16)])
def test_apply_ahead_equals_separate_layernorm_launches(env, dtype, coord, b, h, w, cin, nout, ngf):
    """Plan option APPLY_AHEAD (default off: measured slower): a layer's LayerNorm + ReLU applied by the first workgroups of its consumer's
    launch, behind per-row counters, against one ln_apply launch per layer -- the same arithmetic on the same sums, so the
    prediction and every activation are bit-identical; the wait never times out (error word stays 0)."""
    torch, MSI, nets, N, onets = env
    import ctypes
    probe = ctypes.c_void_p()
    assert N.lib.msi_net_plan_create(nets.make_desc(1, 16, 32, 24, 8, 16, True), ctypes.byref(probe)) == 0
    built = N.lib.msi_net_plan_set_option(probe, N.NET_OPT_APPLY_AHEAD, 1) == 0
    N.lib.msi_net_plan_destroy(probe)
    if not built:
        pytest.skip("apply-ahead is a measured-slower experiment: build with MSI_CNN_DEFINES=-DMSI_EXPERIMENTS to test it")
    weights = onets.init_weights(cin, nout, ngf=ngf, coord_net=coord, seed=21, randomize_affine=True)
    x = torch.rand((b, h, w, cin), device="cuda") * 2 - 1
    if dtype == "bf16":
        x = x.bfloat16()
    outs = []
    for ahead in (1, 0):
        m = MSI(weights=weights, coord_net=coord, dtype=dtype)
        m.net_options[N.NET_OPT_APPLY_AHEAD] = ahead
        m.net_options[N.NET_OPT_HALO] = 0           # (same buffers raw / normalised on both sides)
        pred = m.run_net(x, nout, ngf)
        for _ in range(5):
            assert torch.equal(m.run_net(x, nout, ngf), pred)
        torch.cuda.synchronize()
        desc, _, ws = m._net(b, h, w, cin, nout, ngf)
        acts = []
        for info in nets.layer_infos(desc):
            if info.kind != nets.KIND_HEAD:
                n = b * info.out_h * info.out_w * info.cout
                acts.append(ws[info.raw_offset:info.raw_offset + 4 * n].clone())
        outs.append((pred.clone(), acts))
    assert torch.equal(outs[0][0], outs[1][0])
    if dtype == "f32":      # (the bf16 path keeps the raw fp32 outputs and writes bf16 copies: compare the raw ones)
        pass
    for a, c in zip(outs[0][1], outs[1][1]):
        assert torch.equal(a, c)


@pytest.mark.parametrize("coord,b,h,w,d,ngf", [(True, 1, 32, 64, 32, 64), (False, 2, 16, 40, 8, 16), (True, 1, 16, 32, 64, 16),
                                               (True, 3, 16, 24, 4, 12)])
def test_fused_head_assembly_is_bit_identical(env, coord, b, h, w, d, ngf):
    """msi_net_plan_forward_rgba (1x1 head + conv8_2's LayerNorm + RGBA assembly in one kernel; `pred` never in HBM)
    against msi_net_plan_forward + msi_assemble_rgba_f32: same MFMA sequence, same fp32 expressions -> same bits for the
    layer stack, the optional blend weights / alphas, and (through the oracle) within 1e-3 of the reference path."""
    torch, MSI, nets, N, onets = env
    from oracle.msi import MSI as OracleMSI
    weights = onets.init_weights(6 * d, 2 * d, ngf=ngf, coord_net=coord, seed=33, randomize_affine=True)
    m = MSI(weights=weights, coord_net=coord)
    x = torch.rand((b, h, w, 6 * d), device="cuda") * 2 - 1
    fused = m.infer_layers(x, d, ngf, extra_outputs="blend_weights alphas")
    pred = m.run_net(x, 2 * d, ngf)
    two = m.assemble_layers(x, pred, d, extra_outputs="blend_weights alphas")
    for k in ("rgba_layers", "blend_weights", "alphas"):
        assert torch.equal(fused[k], two[k]), k
    ref = OracleMSI(weights=weights, coord_net=coord).assemble(x.cpu().numpy(), onets.forward(weights, x.cpu().numpy(), coord_net=coord), d)
    assert np.abs(fused["rgba_layers"].cpu().numpy() - ref["rgba_layers"]).max() <= 1e-3
    # the optional tanh output of the fused kernel == the stand-alone head
    desc, packed, ws = m._net(b, h, w, 6 * d, 2 * d, ngf)
    plan = m._plan(b, h, w, 6 * d, 2 * d, ngf)
    rgba = torch.empty((b, d, h, w, 4), device="cuda")
    p2 = torch.empty((b, h, w, 2 * d), device="cuda")
    N.check(N.lib.msi_net_plan_forward_rgba(plan.handle, packed.data_ptr(), x.data_ptr(), rgba.data_ptr(), 0, 0, p2.data_ptr(),
                                            ws.data_ptr(), ws.numel(), None, None), "forward_rgba")
    assert torch.equal(p2, pred)


@pytest.mark.parametrize("halo_opt", [1, 3, 5, 7])
@pytest.mark.parametrize("coord,b,h,w,cin,nout,ngf", [(True, 1, 160, 320, 96, 32, 64), (False, 2, 32, 64, 32, 8, 32),
                                                     (True, 2, 16, 48, 64, 16, 32), (True, 1, 320, 640, 192, 64, 64),
                                                     (False, 4, 128, 256, 48, 16, 64), (False, 1, 24, 48, 32, 8, 32)])
def test_halo_patch_kernel_matches_tap_kernel_and_oracle(env, coord, b, h, w, cin, nout, ngf, halo_opt):
    """conv_halo_kernel (plan option HALO, default on: stride-1 3x3 fp32 layers stage one LDS-stationary halo patch per
    input chunk and apply the producer's LayerNorm on the way) against the tap-DMA kernel: same products, chunk-major
    instead of tap-major summation order -> equal to fp32 round-off; bitwise deterministic; K-ranges at chunk boundaries
    (the first and last shapes split tiles); SAME-zero and wrap padding; rate-2 layers.
    Bit 2 (5 = the default, 7): the stride-2 layers on conv_halo_s2_kernel (parity-plane patches; TF SAME on an even input = one
    padded row / column at the far side, or wrap_pad(1, 1) + VALID) where the grid is large enough -- their producers conv1_1 /
    conv2_1 (/ conv3_2) are then never normalised in memory either."""
    torch, MSI, nets, N, onets = env
    weights = onets.init_weights(cin, nout, ngf=ngf, coord_net=coord, seed=29, randomize_affine=True)
    x = torch.rand((b, h, w, cin), device="cuda") * 2 - 1
    halo = MSI(weights=weights, coord_net=coord)
    halo.net_options[N.NET_OPT_HALO] = halo_opt      # 5: the default; bit 1: + convt_halo_kernel on the conv-transposes (measured slower: opt-in)
    tap = MSI(weights=weights, coord_net=coord)
    tap.net_options[N.NET_OPT_HALO] = 0
    p1, p0 = halo.run_net(x, nout, ngf), tap.run_net(x, nout, ngf)
    plan = halo._plan(b, h, w, cin, nout, ngf)
    raw_layers = [i for i in range(17) if N.lib.msi_net_plan_layer_is_normalized(plan.handle, i) == 0]
    if h % 32 == 0 and w % 128 == 0:                 # every level tiles into 4 x 16 patches:
        assert len(raw_layers) >= (8 if not (halo_opt & 2) or not coord else 14), raw_layers   # (bit 1: everything but the stride-2 layers' sources)
    s2_layers = [i for i, (oh, ow, co) in enumerate([(h // 2, w // 2, 2 * ngf), (h // 4, w // 4, 4 * ngf), (h // 8, w // 8, 8 * ngf)])
                 if oh % 4 == 0 and ow % 16 == 0 and co % 64 == 0 and (oh // 4) * (ow // 16) * (co // 64) * b >= 3 * 256]
    if halo_opt & 4:
        for i in s2_layers:                          # conv1_1 / conv2_1 / conv3_2 feed conv1_2 / conv2_2 / conv3_3 only
            assert [0, 2, 5][i] in raw_layers, (i, raw_layers)
    assert float((p1 - p0).abs().max()) <= 2e-5
    for _ in range(5):
        assert torch.equal(halo.run_net(x, nout, ngf), p1)
    fix = MSI(weights=weights, coord_net=coord)
    fix.net_options[N.NET_OPT_FIXUP_KERNEL] = 1
    fix.net_options[N.NET_OPT_HALO] = halo_opt
    assert torch.equal(fix.run_net(x, nout, ngf), p1)
    if h * w <= 160 * 320:
        ref = onets.forward(weights, x.cpu().numpy(), coord_net=coord)
        assert np.abs(p1.cpu().numpy() - ref).max() <= 1e-3


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_full_size_network_properties_without_the_oracle(env, dtype):
    """Size-independent properties of msi_train_net (wrap padding along W, no CoordNet) at the BASELINE size 640x320, ngf 64,
    where the CPU oracle is too slow to be the checker:
      * W-roll equivariance: every layer wraps along W and the strides multiply to 8, so rolling the input by a multiple of
        8 columns rolls the prediction by the same amount -- this walks every tile seam, halo column, wrap column and
        K-range boundary of the tap, halo and conv-transpose kernels through different positions of the tile grid;
      * gain invariance: LayerNorm (eps 1e-12) removes a power-of-two gain of the input exactly up to rounding of the
        statistics."""
    torch, MSI, nets, N, onets = env
    cin, nout, ngf = 96, 32, 64
    weights = onets.init_weights(cin, nout, ngf=ngf, coord_net=False, seed=123, randomize_affine=True)
    m = MSI(weights=weights, coord_net=False, dtype=dtype)
    x = torch.rand((1, 320, 640, cin), device="cuda") * 2 - 1
    if dtype == "bf16":
        x = x.bfloat16()
    base = m.run_net(x, nout, ngf).clone()
    assert bool(torch.isfinite(base).all())
    # fp32: only the summation order changes (which tiles are split, which wave sums which LayerNorm shard): measured 8e-5 on
    # the tanh prediction after 17 layers with random gamma / beta -- a seam or wrap-column bug shows at 1e-2 and above;
    # bf16: rounding flips of activations on top
    tol = 3e-4 if dtype == "f32" else 4e-2
    for shift in (8, 56, 328):
        rolled = m.run_net(torch.roll(x, shifts=shift, dims=2).contiguous(), nout, ngf)
        d = (rolled - torch.roll(base, shifts=shift, dims=2)).abs()
        assert float(d.max()) <= tol, (shift, float(d.max()))
        if dtype == "bf16":
            assert float(d.mean()) <= 3e-3, (shift, float(d.mean()))
    gained = m.run_net((x.float() * 4.0).to(x.dtype), nout, ngf)     # exact in fp32 and in bf16
    d = (gained - base).abs()
    assert float(d.max()) <= tol, float(d.max())


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("gain", [1e-4, 1e4])
def test_layernorm_follows_the_weight_scale(env, gain, dtype):
    """The LayerNorm sums are one-word fixed point inside a per-layer window whose exponent the packer takes from the
    weights (cnn.hip: LN_S1_BITS), so the window follows any scale of the weights: a network whose conv / conv-transpose
    weights are ALL multiplied by 2^-13 (~1e-4) or 2^13 (~1e4) still matches the ORACLE run on the same scaled weights
    (fp64 two-pass statistics, eps 1e-12 -- which is why the small gain is not exactly invariant: var ~1e-8 against
    eps 1e-12 moves the prediction by ~2e-4, in the reference too) and msi_net_plan_status stays clean; the large gain
    also reproduces the unscaled prediction (LayerNorm removes it)."""
    torch, MSI, nets, N, onets = env
    b, h, w, cin, nout, ngf = 1, 32, 64, 48, 16, 16
    g = float(2.0 ** round(np.log2(gain)))
    weights = onets.init_weights(cin, nout, ngf=ngf, coord_net=True, seed=41, randomize_affine=True)
    scaled = {k: (v * np.float32(g) if k.endswith("/weights") and not k.startswith("color_pred") else v) for k, v in weights.items()}
    xh = np.random.RandomState(5).uniform(-1, 1, size=(b, h, w, cin)).astype(np.float32)
    x = torch.from_numpy(xh).cuda()
    if dtype == "bf16":
        x = x.bfloat16()
    m0, m1 = MSI(weights=weights, coord_net=True, dtype=dtype), MSI(weights=scaled, coord_net=True, dtype=dtype)
    p0, p1 = m0.run_net(x, nout, ngf), m1.run_net(x, nout, ngf)
    assert m0.network_status() == 0 and m1.network_status() == 0
    ref = onets.forward(scaled, x.float().cpu().numpy(), coord_net=True, bf16=dtype == "bf16")
    err = np.abs(p1.cpu().numpy() - ref)
    if dtype == "f32":
        assert err.max() <= 1e-4, err.max()
    else:
        assert err.max() <= 4e-2 and err.mean() <= 3e-3, (err.max(), err.mean())
    if g > 1:
        assert float((p1 - p0).abs().max()) <= (2e-5 if dtype == "f32" else 4e-2)


def test_status_reports_statistics_outside_the_fixed_point_window(env):
    """An input 2^14 times larger / smaller than the [-1, 1] sweep volume the packer assumes pushes conv1_1's raw output out of
    its window: the forward must SAY so (MSI_E_RANGE through MSI.network_status), not return silently wrong values; the
    next, in-range forward is clean again."""
    torch, MSI, nets, N, onets = env
    b, h, w, cin, nout, ngf = 1, 32, 64, 48, 16, 16
    # (msi_train_net: with CoordNet the unscaled |sin(lat)| channel keeps conv1_1's output of order 0.1 whatever the image gain)
    weights = onets.init_weights(cin, nout, ngf=ngf, coord_net=False, seed=43, randomize_affine=True)
    m = MSI(weights=weights, coord_net=False)
    x = torch.rand((b, h, w, cin), device="cuda") * 2 - 1
    m.run_net(x, nout, ngf)
    assert m.network_status() == 0
    for gain, word in ((2.0 ** 14, "overflow"), (2.0 ** -14, "below the resolution")):
        m.run_net(x * gain, nout, ngf)
        with pytest.raises(N.MsiError) as ei:
            m.network_status()
        assert "LayerNorm" in str(ei.value) and ("left its fixed-point window" in str(ei.value) if word == "overflow" else word in str(ei.value))
    m.run_net(torch.full_like(x, float("nan")), nout, ngf)
    with pytest.raises(N.MsiError):
        m.network_status()
    m.run_net(x, nout, ngf)
    assert m.network_status() == 0


def test_bf16_plan_reports_raw_outputs_beyond_the_fp16_range(env):
    """a bf16 plan stores raw outputs as fp16 of x * 2^-e; an input 2^12 times the [-1, 1] volume puts conv1_1's
    stored values near / beyond 65504 -- inside the fixed-point window of the sums (~3000 x 2^e x sqrt-of-count headroom), so
    only the fp16-range check can say it.  The forward must report MSI_E_RANGE; in-range forwards stay clean."""
    torch, MSI, nets, N, onets = env
    b, h, w, cin, nout, ngf = 1, 32, 64, 48, 16, 16
    weights = onets.init_weights(cin, nout, ngf=ngf, coord_net=False, seed=44, randomize_affine=True)
    m = MSI(weights=weights, coord_net=False, dtype="bf16")
    x = (torch.rand((b, h, w, cin), device="cuda") * 2 - 1)
    m.run_net(x.bfloat16(), nout, ngf)
    assert m.network_status() == 0
    m.run_net((x * 8.0).bfloat16(), nout, ngf)                  # a gain of 8 is far inside every range
    assert m.network_status() == 0
    m.run_net((x * 2.0 ** 12).bfloat16(), nout, ngf)
    with pytest.raises(N.MsiError) as ei:
        m.network_status()
    assert "LayerNorm" in str(ei.value)
    m.run_net(x.bfloat16(), nout, ngf)
    assert m.network_status() == 0
