"""Known-answer tests that pin the CPU oracle.  The reference has no
tests / golden vectors of its own and cannot be run here (Python 2 + TF 1.14), so
these analytic facts derived from the reference's formulas are what the oracle is
held to.  Each test names the reference lines it encodes."""
import numpy as np
import pytest
import torch
import torch.nn.functional as TF

from oracle import geometry as G
from oracle import nets
from oracle.msi import MSI

F = np.float32


def test_kat1_inv_depths():
    """msi.py:1196-1217: ends included, uniform in inverse depth, strictly descending."""
    d = MSI().inv_depths(1.0, 100.0, 32)
    assert len(d) == 32 and d[0] == 100.0 and d[-1] == 1.0
    assert all(a > b for a, b in zip(d[:-1], d[1:]))
    inv = 1.0 / np.array(d)
    assert np.allclose(np.diff(inv), np.diff(inv)[0], rtol=1e-9)


def test_kat2_grid_roundtrip():
    """spherical.py:42-44 + 54-68: theta_phi_to_pixels(lat_long_grid) is the integer pixel grid."""
    h, w = 320, 640
    S, T = G.lat_long_grid((h, w))
    uv = G.theta_phi_to_pixels(S, T, w, h)
    assert np.abs(uv[..., 0] - np.arange(w)[None, :]).max() < 1e-3
    assert np.abs(uv[..., 1] - np.arange(h)[:, None]).max() < 1e-3
    # tf.linspace semantics: first element exact, constant fp32 step
    s, _ = G.lat_long_axes(h, w)
    assert s[0] == F(-np.pi + np.pi / w) and s.dtype == np.float32


def test_kat3_identity_render_is_mirror():
    """spherical.py:268-326 with tgt_pos=0, pose=I: u = W-1-j, v = i (to 6.1e-5 at 640x320)."""
    h, w = 320, 640
    uv = G.intersect_sphere(np.eye(4), np.zeros(3), np.array([100.0, 1.0], F), w, h)
    jj = (w - 1 - np.arange(w))[None, :]
    ii = np.arange(h)[:, None]
    for d in range(2):
        assert np.abs(uv[d, ..., 0] - jj).max() < 1e-4
        assert np.abs(uv[d, ..., 1] - ii).max() < 1e-4


def test_kat4_sweep_mirror_and_disparity():
    """spherical.py:170-233: u ~ W-1-j +- asin(r/(d cosT)) W/2pi; +order shifts +, -order shifts -."""
    h, w, r = 320, 640, 0.032
    S, T = G.lat_long_grid((h, w))
    pts = G.backproject_spherical(S, T, np.array([100.0, 1.0], F))
    expect = np.arcsin(r / 1.0) * w / (2 * np.pi)   # 3.26 px at the equator for d = 1
    for order in (1, -1):
        uv, _, valid = G.project_ods(pts, order, r, w, h)
        d = uv[1, h // 2, :, 0] - (w - 1 - np.arange(w))
        d = np.where(d > w / 2, d - w, np.where(d < -w / 2, d + w, d))   # wrap-around columns
        assert abs(np.median(d) - order * expect) < 0.02
        far = uv[0, h // 2, :, 0] - (w - 1 - np.arange(w))
        assert abs(np.median(far)) < 0.1                                  # d = 100: 0.03 px
        assert np.abs(uv[1, 20:-20, :, 1] - np.arange(h)[20:-20, None]).max() < 0.6


def test_kat5_invalid_pixel_rule():
    """spherical.py:226-229: where disc < 0 (d cosT < r: polar rows) the sample is pixel (1,1)."""
    h, w, r = 320, 640, 0.032
    S, T = G.lat_long_grid((h, w))
    pts = G.backproject_spherical(S, T, np.array([1.0], F))
    uv, _, valid = G.project_ods(pts, 1, r, w, h)
    assert (~valid[0]).sum() == 6 * w          # 3 rows at each pole: |T| > acos(0.032)
    assert np.all(uv[0][~valid[0]] == 1.0)
    rows = np.where((~valid[0]).any(axis=1))[0]
    assert list(rows) == [0, 1, 2, h - 3, h - 2, h - 1]


def test_kat6_resample_wraps_both_axes():
    """sampling.py:150-165: weights from unwrapped corners, indices mod W and mod H."""
    h, w = 4, 6
    img = np.arange(h * w, dtype=F).reshape(1, h, w, 1)
    px = np.array([[[[w - 0.5, 1.0], [2.0, -0.5], [-0.25, h - 0.75]]]], dtype=F)
    out = G.resample(img, px)[0, 0, :, 0]
    assert out[0] == 0.5 * (img[0, 1, w - 1, 0] + img[0, 1, 0, 0])          # x wraps W-1 -> 0
    assert out[1] == 0.5 * (img[0, h - 1, 2, 0] + img[0, 0, 2, 0])          # y wraps H-1 -> 0
    e = (0.75 * 0.25 * img[0, h - 1, w - 1, 0] + 0.75 * 0.75 * img[0, h - 1, 0, 0]
         + 0.25 * 0.25 * img[0, 0, w - 1, 0] + 0.25 * 0.75 * img[0, 0, 0, 0])
    assert abs(out[2] - e) < 1e-6


def test_kat7_over_composite():
    """projector.py:246-265 / 225-244: first alpha ignored; opaque layer k hides the farther ones."""
    rng = np.random.RandomState(0)
    d, k = 6, 2
    layers = [rng.uniform(-1, 1, size=(1, 3, 4, 4)).astype(F) for _ in range(d)]
    for i, l in enumerate(layers):
        l[..., 3] = 1.0 if i == k else 0.0
    layers[0][..., 3] = 0.3
    out = G.over_composite(layers)
    assert np.array_equal(out, layers[k][..., :3])
    dep = G.over_composite_depth(layers)
    assert np.allclose(dep, k / d)
    # alpha of layer 0 really is ignored
    layers[0][..., 3] = 0.9
    assert np.array_equal(G.over_composite(layers), out)


def test_kat8_conv_layernorm_semantics_vs_torch():
    """App. B [TF-knowledge]: SAME padding 0/1 at stride 2, conv-transpose y = 2i + k - 1,
    LayerNorm over (H,W,C) with per-channel affine -- the oracle net against independent
    plain-torch formulations."""
    rng = np.random.RandomState(1)
    x = torch.from_numpy(rng.normal(size=(1, 5, 8, 12)).astype(F))
    w = torch.from_numpy(rng.normal(size=(7, 5, 3, 3)).astype(F))
    assert nets._same_pad(8, 3, 2) == (0, 1) and nets._same_pad(8, 3, 1) == (1, 1) and nets._same_pad(8, 5, 1) == (2, 2)
    y = TF.conv2d(TF.pad(x, (0, 1, 0, 1)), w, stride=2)
    assert tuple(y.shape) == (1, 7, 4, 6)
    # conv-transpose as the adjoint of the SAME stride-2 conv: <conv(u), v> == <u, convT(v)>
    wt = torch.from_numpy(rng.normal(size=(5, 7, 4, 4)).astype(F))       # [Cin, Cout, kh, kw]
    u = torch.from_numpy(rng.normal(size=(1, 7, 16, 24)).astype(F)).double()
    v = torch.from_numpy(rng.normal(size=(1, 5, 8, 12)).astype(F)).double()
    fwd = TF.conv2d(TF.pad(u, (1, 1, 1, 1)), wt.double(), stride=2)      # k4 s2 SAME: pad 1/1
    adj = TF.conv_transpose2d(v, wt.double(), stride=2, padding=1)
    assert abs(float((fwd * v).sum()) - float((u * adj).sum())) < 1e-6 * float(fwd.abs().sum())
    # LayerNorm: zero mean / unit variance over (C,H,W), then per-channel affine, then ReLU
    g = rng.uniform(0.5, 1.5, size=5).astype(F)
    b = rng.uniform(-0.2, 0.2, size=5).astype(F)
    out = nets.layer_norm_relu(x, g, b)
    ref = TF.layer_norm(x, tuple(x.shape[1:]), eps=1e-12) * torch.from_numpy(g).view(1, -1, 1, 1) + torch.from_numpy(b).view(1, -1, 1, 1)
    assert torch.allclose(out, torch.relu(ref), atol=2e-6)


def test_kat9_deprocess():
    """msi.py:1173-1194 + convert_image_dtype: floor(((x+1)/2)*255.5); depth variant skips (x+1)/2."""
    m = MSI()
    x = np.array([-1.0, -0.5, 0.0, 0.5, 1.0], F)
    assert list(m.deprocess_image(x)) == [0, 63, 127, 191, 255]
    assert list(m.deprocess_depth_image(np.array([0.0, 0.5, 1.0], F))) == [0, 127, 255]
    assert np.array_equal(m.preprocess_image(np.array([0, 255], np.uint8)), np.array([-1.0, 1.0], F))


def test_coordnet_channel():
    """nets.py:260-265: the extra channel is |sin(linspace(-pi/2, pi/2, H))|, constant along W."""
    x = torch.zeros((1, 2, 5, 3))
    y = nets.add_sph_coords(x)
    assert tuple(y.shape) == (1, 3, 5, 3)
    col = y[0, 2, :, 0].numpy()
    assert np.allclose(col, [1.0, np.sin(np.pi / 4), 0.0, np.sin(np.pi / 4), 1.0], atol=1e-6)
    assert torch.equal(y[0, 2, :, 0], y[0, 2, :, 2])


def test_layer_table_matches_reference_counts():
    """nets.py:471-515: 16.98 M parameters at in=192, out=64, ngf=64 with CoordNet."""
    w = nets.init_weights(192, 64, 64, True)
    n = sum(v.size for v in w.values())
    assert n == 16980160
    assert w["conv1_1/weights"].shape == (3, 3, 193, 64)
    assert w["conv6_1/weights"].shape == (4, 4, 256, 1024)
    assert w["color_pred/weights"].shape == (1, 1, 64, 64) and w["color_pred/biases"].shape == (64,)


def test_kat10_wrap_convT_layernorm_runs_before_the_crop():
    """nets.py:423-435: msi_train_net's conv6_1 / conv7_1 / conv8_1 are slim.conv2d_transpose(wrap_pad(skip, 2, 2),
    padding='VALID') under the layer_norm arg_scope, so LayerNorm + ReLU see the whole (2H+10) x (2W+10) output and
    the [5:-5] crop follows.  Independent scatter-form restatement (y[2i+kh, 2j+kw] += x[i, j] . w[kh, kw]) on a tiny
    network; also shows that normalising after the crop is a different function."""
    rng = np.random.RandomState(7)
    cin, nout, ngf, h, w = 8, 4, 4, 8, 16
    weights = nets.init_weights(cin, nout, ngf=ngf, coord_net=False, seed=7, randomize_affine=True)
    x = rng.uniform(-1, 1, size=(1, h, w, cin)).astype(F)
    _, acts = nets.forward(weights, x, coord_net=False, return_activations=True)
    skip = np.concatenate([acts["conv4_3"], acts["conv3_3"]], axis=3)[0].astype(np.float64)      # [H/8, W/8, 16 ngf]
    hh, ww, _ = skip.shape
    padded = np.zeros((hh + 4, ww + 4, skip.shape[2]))
    padded[2:-2] = np.concatenate([skip[:, -2:], skip, skip[:, :2]], axis=1)                     # wrap W, zeros H
    wt = weights["conv6_1/weights"].astype(np.float64)                                           # [4,4,Cout,Cin]
    full = np.zeros((2 * hh + 10, 2 * ww + 10, wt.shape[2]))
    for i in range(hh + 4):
        for j in range(ww + 4):
            for kh in range(4):
                for kw in range(4):
                    full[2 * i + kh, 2 * j + kw] += wt[kh, kw] @ padded[i, j]
    g = weights["conv6_1/LayerNorm/gamma"].astype(np.float64)
    be = weights["conv6_1/LayerNorm/beta"].astype(np.float64)

    def ln_relu(t):
        return np.maximum((t - t.mean()) / np.sqrt(t.var() + 1e-12) * g + be, 0.0)

    expect = ln_relu(full)[5:-5, 5:-5]
    assert np.abs(acts["conv6_1/raw"][0] - full[5:-5, 5:-5]).max() < 1e-5
    assert np.abs(acts["conv6_1"][0] - expect).max() < 2e-5
    assert np.abs(ln_relu(full[5:-5, 5:-5]) - expect).max() > 1e-3    # crop-then-normalise is NOT what the reference does
    assert np.all(full[:4] == 0) and np.all(full[-4:] == 0) and np.any(full[4] != 0) and np.any(full[-5] != 0)


def test_reference_function_bodies_agree_with_the_oracle():
    """oracle/crosscheck_reference.py: the reference's own geometry / nets / msi function bodies, executed over a numpy
    stand-in for tensorflow, are bit-identical to the oracle's restatement (build container only: /root/reference does not
    exist on the GPU box and is never read by a -m gpu test, smoke() or bench.py)."""
    import os
    import subprocess
    import sys
    import pytest
    if not os.path.isdir("/root/reference/geometry"):
        pytest.skip("no reference checkout here")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "oracle", "crosscheck_reference.py")], cwd=root, timeout=600,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert p.returncode == 0 and "MISMATCH" not in p.stdout and p.stdout.count("bit-identical") >= 50, p.stdout[-3000:]
