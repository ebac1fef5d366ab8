"""msi_net_plan_calibrate (round 5): the LayerNorm fixed-point windows are an ESTIMATE the packer makes from the weights; a network whose
raw outputs sit far from that estimate ends every forward in MSI_E_RANGE.  Calibration measures the windows on a frame and rewrites them in the packed blob:
afterwards the same network runs to parity with the oracle, with no plan option touched."""
import numpy as np
import pytest

from tests.test_gpu_cnn import env  # noqa: F401

pytestmark = pytest.mark.gpu


def _hostile_weights(onets, cin, nout, ngf, coord, seed):
    """conv weights scaled per layer by 10^U(-3, 3), gamma by 10^U(-3, 3) (beta with it), and -- what the packer's rms estimate cannot see -- a common
    offset on some layers' weights: post-ReLU inputs are positive, so the products add coherently (K mean(w) mean(x) instead of sqrt(K) rms(w) rms(x))."""
    w = onets.init_weights(cin, nout, ngf=ngf, coord_net=coord, seed=seed, randomize_affine=True)
    rng = np.random.RandomState(seed + 100)
    for k in sorted(w):
        if k.endswith("/weights") and not k.startswith("color_pred"):
            w[k] = (w[k] * 10.0 ** rng.uniform(-3, 3)).astype(np.float32)
            if rng.rand() < 0.5:
                w[k] = (w[k] + 10.0 * np.abs(w[k]).mean()).astype(np.float32)
        elif k.endswith("/gamma") and not k.startswith("conv8_2"):      # (conv8_2's affine feeds tanh through the unscaled head: left O(1) so that 1e-3 on the output means something)
            f = 10.0 ** rng.uniform(-3, 3)
            w[k] = (w[k] * f).astype(np.float32)
            w[k.replace("gamma", "beta")] = (w[k.replace("gamma", "beta")] * f).astype(np.float32)
    return w


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("coord", [True, False])
def test_calibration_heals_windows_the_weights_do_not_predict(env, coord, dtype):
    torch, MSI, nets, N, onets = env
    b, h, w, cin, nout, ngf = 2, 64, 128, 96, 32, 32
    weights = _hostile_weights(onets, cin, nout, ngf, coord, seed=11)
    rng = np.random.RandomState(3)
    x = (rng.uniform(-1, 1, size=(b, h, w, cin)) * 3.0e4).astype(np.float32)      # (and an input 2^15 times the [-1, 1] volume the estimate assumes)
    m = MSI(weights=weights, coord_net=coord, dtype=dtype)
    xd = torch.from_numpy(x).cuda()
    m.run_net(xd, nout, ngf)
    with pytest.raises(N.MsiError):
        m.network_status()                                                        # the estimated windows do not hold this network
    moved = m.calibrate(xd, nout, ngf)
    assert moved >= 1
    pred = m.run_net(xd, nout, ngf)
    assert m.network_status() == 0
    if dtype == "bf16":
        # (bf16 plans store raw outputs as fp16 of x * 2^-e with e from the SAME window, cnn_device.h raw_mul: the calibrated window also keeps them in fp16's range;
        #  tolerances of tests/test_gpu_bf16.py against the bf16 oracle)
        ref = onets.forward(weights, x, coord_net=coord, bf16="scaled")
        err = np.abs(pred.cpu().numpy() - ref)
        print("calibrated %d layers (bf16); pred vs bf16 oracle max %.2e mean %.2e" % (moved, err.max(), err.mean()))
        assert err.max() <= 6e-2 and err.mean() <= 3e-3, (err.max(), err.mean())
    else:
        ref = onets.forward(weights, x, coord_net=coord)
        err = float(np.abs(pred.cpu().numpy() - ref).max())
        print("calibrated %d layers; pred vs oracle %.2e" % (moved, err))
        assert err <= 1e-3
    # the windows live in the packed blob: another batch size of the same model needs no second calibration
    x1 = xd[:1].contiguous()
    p1 = m.run_net(x1, nout, ngf)
    assert m.network_status() == 0
    assert float((p1 - pred[:1]).abs().max()) <= (1e-5 if dtype == "f32" else 2.0 ** -7)
    # a healthy network: calibration moves nothing by more than the centring and changes no result beyond summation rounding of the statistics
    good = onets.init_weights(cin, nout, ngf=ngf, coord_net=coord, seed=5, randomize_affine=True)
    g = MSI(weights=good, coord_net=coord, dtype=dtype)
    xg = torch.from_numpy(rng.uniform(-1, 1, size=(b, h, w, cin)).astype(np.float32)).cuda()
    before = g.run_net(xg, nout, ngf).clone()
    g.calibrate(xg, nout, ngf)
    after = g.run_net(xg, nout, ngf)
    assert g.network_status() == 0 and float((after - before).abs().max()) <= (2e-6 if dtype == "f32" else 2.0 ** -7)


def test_calibration_reports_a_network_without_finite_output(env):
    torch, MSI, nets, N, onets = env
    b, h, w, cin, nout, ngf = 1, 32, 64, 24, 8, 16
    weights = onets.init_weights(cin, nout, ngf=ngf, coord_net=True, seed=2, randomize_affine=True)
    weights["conv2_1/weights"][0, 0, 0, 0] = np.nan
    m = MSI(weights=weights, coord_net=True)
    x = torch.rand((b, h, w, cin), device="cuda") * 2 - 1
    with pytest.raises(N.MsiError) as ei:
        m.calibrate(x, nout, ngf)
    assert "conv2_1" in str(ei.value)


def _ln_windows(m, nets, torch, b, h, w, cin, nout, ngf):
    """The four scale doubles of every LayerNorm layer, read from the packed blob on the device."""
    desc, packed, _ = m._net(b, h, w, cin, nout, ngf)
    host = packed.cpu().numpy()
    out = []
    for info in nets.layer_infos(desc)[:-1]:
        off = int(info.ln_scale_offset)
        out.append(host[off:off + 8].view(np.float64).copy())
    return np.stack(out)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_failed_calibration_leaves_every_window_as_it_was(env, dtype):
    """calibration is ALL OR NOTHING.  A constant frame (rmax == 0 on conv1_1 of msi_train_net: the window used to walk down to 2^-120) and a
    NaN frame both fail -- and afterwards the blob is bit for bit what it was, and a healthy frame still runs with status 0 and the same result."""
    torch, MSI, nets, N, onets = env
    b, h, w, cin, nout, ngf = 1, 32, 64, 24, 8, 16
    weights = onets.init_weights(cin, nout, ngf=ngf, coord_net=False, seed=4, randomize_affine=True)
    m = MSI(weights=weights, coord_net=False, dtype=dtype)
    x = torch.rand((b, h, w, cin), device="cuda") * 2 - 1
    before = m.run_net(x, nout, ngf).clone()
    assert m.network_status() == 0
    w0 = _ln_windows(m, nets, torch, b, h, w, cin, nout, ngf)
    for bad in (torch.zeros_like(x), torch.full_like(x, float("nan"))):
        with pytest.raises(N.MsiError) as ei:
            m.calibrate(bad, nout, ngf)
        assert "unchanged" in str(ei.value)
        assert np.array_equal(_ln_windows(m, nets, torch, b, h, w, cin, nout, ngf), w0)
        after = m.run_net(x, nout, ngf)
        assert m.network_status() == 0 and torch.equal(after, before)


def test_calibration_ignores_a_constant_sample_among_resolvable_ones(env):
    """a batch with one black frame used to ping-pong (under -> 12 bits down -> recentre) until the iteration cap; the window now follows the samples it resolves."""
    torch, MSI, nets, N, onets = env
    b, h, w, cin, nout, ngf = 3, 32, 64, 24, 8, 16
    weights = _hostile_weights(onets, cin, nout, ngf, False, seed=21)
    m = MSI(weights=weights, coord_net=False)
    x = (torch.rand((b, h, w, cin), device="cuda") * 2 - 1) * 2.0e3
    x[1] = 0.0
    moved = m.calibrate(x, nout, ngf)
    assert moved >= 1
    pred = m.run_net(x, nout, ngf)
    try:
        bits = m.network_status()
    except N.MsiError:
        bits = None                      # (the constant sample itself may be flagged LN_UNDERFLOW: that is the forward's own report about THAT sample)
    ref = onets.forward(weights, x.cpu().numpy(), coord_net=False)
    for k in (0, 2):
        assert float(np.abs(pred[k].cpu().numpy() - ref[k]).max()) <= 1e-3, (k, bits)
