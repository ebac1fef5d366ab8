"""Oracle (TEST INFRASTRUCTURE ONLY, parity unpinned -- see oracle/__init__.py):
torch-CPU fp32 restatement of the reference's encoder-decoder CNN.

Reference followed: matryodshka/nets.py
  :260-270  add_sph_coords / coord_conv2d       (CoordNet, `--coord_net`)
  :288-295  wrap_pad
  :387-450  msi_train_net        (horizontal wrap + vertical zero pad, VALID)
  :471-515  msi_coord_train_net  (SAME zero pad + |sin(lat)| coordinate channel)

tf.contrib.slim semantics restated here are [TF-knowledge]:
  * slim.conv2d: NHWC, weights [kh,kw,Cin,Cout], cross-correlation, SAME pad =
    total max((ceil(in/s)-1)*s + k_eff - in, 0), floor(total/2) before, the
    rest after; no bias when a normalizer is set; conv -> LayerNorm -> ReLU.
  * slim.conv2d_transpose k4 s2: weights [kh,kw,Cout,Cin]; SAME: out = 2*in,
    y[2i+k-1] += x[i]*w[k]  (== torch conv_transpose2d(stride=2,padding=1));
    VALID: out = 2*in+2, y[2i+k] += x[i]*w[k].
  * slim.layer_norm: per sample over (H,W,C); gamma/beta per channel;
    eps = 1e-12; two-pass variance.
  * head `color_pred`: 1x1 conv with bias, tanh, no normalizer.
Variable names mirror the TF checkpoint scope `net/` (nets.py:483).
"""
import math
import numpy as np
import torch
import torch.nn.functional as TF

F32 = np.float32
LN_EPS = 1e-12

# (name, kind, cout_mult, stride, rate) ; kind: 'c' conv3x3, 't' convT4x4s2
_ENCODER = [
    ("conv1_1", 1, 1, 1), ("conv1_2", 2, 2, 1), ("conv2_1", 2, 1, 1), ("conv2_2", 4, 2, 1),
    ("conv3_1", 4, 1, 1), ("conv3_2", 4, 1, 1), ("conv3_3", 8, 2, 1),
    ("conv4_1", 8, 1, 2), ("conv4_2", 8, 1, 2), ("conv4_3", 8, 1, 2),
]


def layer_table(in_channels, num_outputs, ngf=64, coord_net=True):
    """[(name, kind, cin (without coord channel), cout)] in graph order."""
    ex = 1 if coord_net else 0
    t = []
    cin = in_channels
    for name, mult, _s, _r in _ENCODER:
        t.append((name, "c", cin, ngf * mult, ex))
        cin = ngf * mult
    t.append(("conv6_1", "t", ngf * 16, ngf * 4, 0))
    t.append(("conv6_2", "c", ngf * 4, ngf * 4, ex))
    t.append(("conv6_3", "c", ngf * 4, ngf * 4, ex))
    t.append(("conv7_1", "t", ngf * 8, ngf * 2, 0))
    t.append(("conv7_2", "c", ngf * 2, ngf * 2, ex))
    t.append(("conv8_1", "t", ngf * 4, ngf, 0))
    t.append(("conv8_2", "c", ngf, ngf, ex))
    t.append(("color_pred", "h", ngf, num_outputs, 0))
    return t


def init_weights(in_channels, num_outputs, ngf=64, coord_net=True, seed=8964,
                 randomize_affine=False):
    """Xavier-uniform weights (slim default [TF-knowledge]), LN gamma=1 beta=0,
    head bias 0.  `randomize_affine` perturbs gamma/beta/bias so tests exercise
    them.  Returns dict name -> np.float32 array in TF variable layout."""
    rng = np.random.RandomState(seed)
    w = {}
    for name, kind, cin, cout, ex in layer_table(in_channels, num_outputs, ngf, coord_net):
        if kind == "c":
            shape = (3, 3, cin + ex, cout)
            fan_in, fan_out = 9 * (cin + ex), 9 * cout
        elif kind == "t":
            shape = (4, 4, cout, cin)
            # slim: fan_in/out from shape[-2]/shape[-1] times the receptive field
            fan_in, fan_out = 16 * cout, 16 * cin
        else:
            shape = (1, 1, cin, cout)
            fan_in, fan_out = cin, cout
        lim = math.sqrt(6.0 / (fan_in + fan_out))
        w[name + "/weights"] = rng.uniform(-lim, lim, size=shape).astype(F32)
        if kind == "h":
            b = rng.uniform(-0.1, 0.1, size=(cout,)) if randomize_affine else np.zeros(cout)
            w[name + "/biases"] = b.astype(F32)
        else:
            if randomize_affine:
                g = rng.uniform(0.5, 1.5, size=(cout,))
                be = rng.uniform(-0.2, 0.2, size=(cout,))
            else:
                g, be = np.ones(cout), np.zeros(cout)
            w[name + "/LayerNorm/gamma"] = g.astype(F32)
            w[name + "/LayerNorm/beta"] = be.astype(F32)
    return w


def _same_pad(size, k_eff, stride):
    out = -(-size // stride)
    total = max((out - 1) * stride + k_eff - size, 0)
    return total // 2, total - total // 2


def add_sph_coords(x):
    """nets.add_sph_coords (nets.py:260-265). x: torch [B,C,H,W]. The
    `input/float_info.max` term is exactly +0 in fp32."""
    b, _, h, w = x.shape
    coord = np.abs(np.sin(np.linspace(-np.pi / 2.0, np.pi / 2.0, h))).astype(F32)
    c = torch.from_numpy(coord).view(1, 1, h, 1).expand(b, 1, h, w)
    return torch.cat([x, c], dim=1)


def layer_norm_relu(x, gamma, beta, affine_out=None, raw16=False):
    """slim.layer_norm over (H,W,C) per sample + ReLU (nets.py:401,485 arg_scope).
    Statistics in fp64 (the ideal two-pass value), normalisation in fp32.
    raw16 (the build-defined bf16 variant only): the statistics come from the fp32 accumulators, the affine is applied to
    the raw output as the bf16 plan STORES it -- fp16 (11 significand bits; the kernels scale by a power of two first,
    which does not change the rounding of in-range values)."""
    xd = x.double()
    mean = xd.mean(dim=(1, 2, 3), keepdim=True)
    var = ((xd - mean) ** 2).mean(dim=(1, 2, 3), keepdim=True)
    inv = torch.rsqrt(var + LN_EPS)
    g = torch.from_numpy(gamma).view(1, -1, 1, 1).double()
    be = torch.from_numpy(beta).view(1, -1, 1, 1).double()
    scale = (inv * g).float()
    shift = (be - mean * inv * g).float()
    if affine_out is not None:      # tests: the per-sample, per-channel affine [B, 2, C]
        affine_out.append(torch.stack([scale.flatten(1), shift.flatten(1)], dim=1).numpy())
    if raw16 == "scaled":
        # the kernels' power-of-two pre-scale made explicit (needed only where raw values leave fp16's RANGE, e.g. tests/test_gpu_calibrate.py's mis-scaled
        # networks; in-range values round identically with and without it): bring each sample's rms to ~1 before the fp16 rounding
        rms = torch.sqrt((xd ** 2).mean(dim=(1, 2, 3), keepdim=True)).clamp_min(1e-300)
        p2 = torch.exp2(-torch.round(torch.log2(rms))).float()
        x = (x * p2).half().float() / p2
    elif raw16:
        x = x.half().float()
    return torch.relu(x * scale + shift)


def _conv_w(w):   # [kh,kw,Cin,Cout] -> [Cout,Cin,kh,kw]
    return torch.from_numpy(np.ascontiguousarray(np.transpose(w, (3, 2, 0, 1))))


def _convT_w(w):  # [kh,kw,Cout,Cin] -> [Cin,Cout,kh,kw]
    return torch.from_numpy(np.ascontiguousarray(np.transpose(w, (3, 2, 0, 1))))


def wrap_pad(x, lp, rp):
    """nets.wrap_pad (nets.py:288-295) on NCHW: wrap along W, zeros along H."""
    x = torch.cat([x[..., -lp:], x, x[..., :rp]], dim=-1)
    return TF.pad(x, (0, 0, lp, rp))


def bf16_round(t):
    """fp32 -> bf16 -> fp32 (round to nearest even), torch tensor or numpy array."""
    if isinstance(t, np.ndarray):
        return torch.from_numpy(np.ascontiguousarray(t, dtype=F32)).bfloat16().float().numpy()
    return t.bfloat16().float()


def split3(t):
    """x = h + m + l with bf16 parts (8 + 8 + 8 significand bits): the operand format of the 6-product study below."""
    h = t.bfloat16().float()
    r = t - h
    m = r.bfloat16().float()
    return h, m, (r - m).bfloat16().float()


def conv_split3(fn, x, w):
    """STUDY ONLY (tools/split3_study.py; not a product path): fp32 convolution emulated by six bf16 x bf16
    products with fp32 accumulation -- h.h + h.m + m.h + h.l + l.h + m.m (the dropped m.l, l.m, l.l terms are <= 2^-24 of the
    leading one).  Each term is an fp32 torch convolution of bf16-representable operands (exact products, fp32 sums), the
    terms are added smallest first."""
    xh, xm, xl = split3(x)
    wh, wm, wl = split3(w)
    return ((fn(xm, wm) + fn(xl, wh)) + fn(xh, wl)) + ((fn(xm, wh) + fn(xh, wm)) + fn(xh, wh))


def split_f16(t):
    """x = h + m' 2^-11 with fp16 parts (11 + 11 significand bits; m' is the remainder scaled into the fp16 normal range):
    the operand format of the three-product form (plan option F32_SPLIT_F16)."""
    h = t.half().float()
    return h, ((t - h) * 2048.0).half().float()


def conv_split_f16(fn, x, w):
    """STUDY ONLY (tools/split3_study.py; not a product path): fp32 convolution emulated by THREE fp16 x fp16 products with fp32
    accumulation -- h.h + (h.m' + m'.h) 2^-11; operands carry 22 significand bits, the dropped m'.m' term is <= 2^-22 of the
    leading one.  Each term is an fp32 torch convolution of fp16-representable operands (exact products, fp32 sums)."""
    xh, xm = split_f16(x)
    wh, wm = split_f16(w)
    return fn(xh, wh) + (fn(xm, wh) + fn(xh, wm)) * (1.0 / 2048.0)


def forward(weights, net_input, coord_net=True, return_activations=False, bf16=False, split3_products=False):
    """split3_products: False | True (six bf16 products, conv_split3) | "f16" (three fp16 products, conv_split_f16).
    bf16: False | True | "scaled" (True with the fp16 raw storage's power-of-two pre-scale explicit: layer_norm_relu)."""
    return _forward(weights, net_input, coord_net, return_activations, bf16, split3_products)


def _forward(weights, net_input, coord_net=True, return_activations=False, bf16=False, split3_products=False):
    """msi_coord_train_net (nets.py:471-515) / msi_train_net (:387-450).
    net_input: np [B,H,W,Cin] fp32.  Returns np [B,H,W,num_outputs] fp32.

    bf16=True models BASELINE configs[2] (the reference has no bf16 code: this defines the variant the
    build implements): every convolution OPERAND is rounded to bf16 -- the network input, the weights,
    the coordinate channel, each LayerNorm+ReLU output -- while products are accumulated in fp32, the
    LayerNorm statistics are taken from the unrounded accumulators, the raw convolution output is kept as
    fp16 (r03: half the activation bytes; measured here: mean |bf16 variant - fp32 network| + 0.3 %, max within
    its seed-to-seed noise; a bf16 raw output would be + 19 %) and the affine, the head bias and tanh are fp32."""
    rnd = bf16_round if bf16 else (lambda t: t)
    split_fn = conv_split_f16 if split3_products == "f16" else conv_split3
    x = rnd(torch.from_numpy(np.ascontiguousarray(np.transpose(net_input, (0, 3, 1, 2)))).float())
    acts = {}
    affines = {}

    def ln(name, y):
        out = []
        y = layer_norm_relu(y, weights[name + "/LayerNorm/gamma"], weights[name + "/LayerNorm/beta"], affine_out=out, raw16=bf16)
        affines[name] = out[0]
        return y

    def conv(name, x, stride=1, rate=1):
        w = rnd(_conv_w(weights[name + "/weights"]))
        if coord_net:
            x = rnd(add_sph_coords(x))
            _, _, h, wd = x.shape
            pt, pb = _same_pad(h, 2 * rate + 1, stride)
            pl, pr = _same_pad(wd, 2 * rate + 1, stride)
            x = TF.pad(x, (pl, pr, pt, pb))
        else:
            x = wrap_pad(x, rate, rate)
        if split3_products:
            y = split_fn(lambda a, b_: TF.conv2d(a, b_, stride=stride, dilation=rate), x, w)
        else:
            y = TF.conv2d(x, w, stride=stride, dilation=rate)
        acts[name + "/raw"] = y
        y = rnd(ln(name, y))
        acts[name] = y
        return y

    def convT(name, x):
        w = rnd(_convT_w(weights[name + "/weights"]))
        if coord_net:
            if split3_products:
                y = split_fn(lambda a, b_: TF.conv_transpose2d(a, b_, stride=2, padding=1), x, w)
            else:
                y = TF.conv_transpose2d(x, w, stride=2, padding=1)
            acts[name + "/raw"] = y
            y = rnd(ln(name, y))
        else:
            # nets.py:423-435: slim.conv2d_transpose(wrap_pad(skip, 2, 2), padding='VALID') runs under the
            # layer_norm arg_scope, so LayerNorm + ReLU cover the FULL (2H+10) x (2W+10) output -- zero rows and
            # wrap columns of the border included -- and the [5:-5] crop happens afterwards, in the consumer's
            # input expression (cnv6_1[:,5:-5,5:-5,:], nets.py:426)
            y = TF.conv_transpose2d(wrap_pad(x, 2, 2), w, stride=2, padding=0)
            acts[name + "/raw"] = y[:, :, 5:-5, 5:-5]
            y = rnd(ln(name, y))
            y = y[:, :, 5:-5, 5:-5]
        acts[name] = y
        return y

    with torch.no_grad():
        c11 = conv("conv1_1", x)
        c12 = conv("conv1_2", c11, stride=2)
        c21 = conv("conv2_1", c12)
        c22 = conv("conv2_2", c21, stride=2)
        c31 = conv("conv3_1", c22)
        c32 = conv("conv3_2", c31)
        c33 = conv("conv3_3", c32, stride=2)
        c41 = conv("conv4_1", c33, rate=2)
        c42 = conv("conv4_2", c41, rate=2)
        c43 = conv("conv4_3", c42, rate=2)
        c61 = convT("conv6_1", torch.cat([c43, c33], dim=1))
        c62 = conv("conv6_2", c61)
        c63 = conv("conv6_3", c62)
        c71 = convT("conv7_1", torch.cat([c63, c22], dim=1))
        c72 = conv("conv7_2", c71)
        c81 = convT("conv8_1", torch.cat([c72, c12], dim=1))
        c82 = conv("conv8_2", c81)
        w = rnd(_conv_w(weights["color_pred/weights"]))
        b = torch.from_numpy(weights["color_pred/biases"])
        if split3_products:
            pred = torch.tanh(split_fn(lambda a, b_: TF.conv2d(a, b_), c82, w) + b.view(1, -1, 1, 1))
        else:
            pred = torch.tanh(TF.conv2d(c82, w, bias=b))
    out = np.ascontiguousarray(pred.permute(0, 2, 3, 1).numpy())
    if return_activations:
        res = {k: np.ascontiguousarray(v.permute(0, 2, 3, 1).numpy()) for k, v in acts.items()}
        res.update({k + "/affine": v for k, v in affines.items()})     # [B, 2, C]: scale | shift
        return out, res
    return out
