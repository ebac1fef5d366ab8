"""Host-side description of the MSI prediction network: variable names/shapes,
initialisation, and the parameter blob handed to the native library.

Mirrors the variable structure of the reference's `msi_coord_train_net`
(nets.py:471-515) and `msi_train_net` (nets.py:387-450): TF scope `net/`,
`<layer>/weights`, `<layer>/LayerNorm/{gamma,beta}`, `color_pred/{weights,biases}`
(the variables test.py:191-202 restores).  The arithmetic lives in
csrc/cnn.hip; nothing here computes a convolution.
"""
import math

import numpy as np

from . import _native as N

LAYER_NAMES = ["conv1_1", "conv1_2", "conv2_1", "conv2_2", "conv3_1", "conv3_2", "conv3_3",
               "conv4_1", "conv4_2", "conv4_3", "conv6_1", "conv6_2", "conv6_3", "conv7_1",
               "conv7_2", "conv8_1", "conv8_2", "color_pred"]
KIND_CONV, KIND_CONVT, KIND_HEAD = 0, 1, 2


def make_desc(batch, height, width, in_channels, num_outputs, ngf=64, coord_net=True, dtype="f32"):
    """dtype: "f32" (BASELINE configs 1, 2, 4, 5) or "bf16" (configs[2]: bf16 operands, fp32 accumulate)."""
    if dtype not in ("f32", "bf16"):
        raise ValueError("dtype must be 'f32' or 'bf16'")
    return N.NetDesc(int(batch), int(height), int(width), int(in_channels), int(num_outputs),
                     int(ngf), 1 if coord_net else 0, N.MSI_DTYPE_BF16 if dtype == "bf16" else N.MSI_DTYPE_F32)


def layer_infos(desc):
    out = []
    for i in range(N.MSI_NET_NUM_LAYERS):
        info = N.LayerInfo()
        N.check(N.lib.msi_net_layer_info(desc, i, info), "msi_net_layer_info")
        out.append(info)
    return out


def variable_shapes(in_channels, num_outputs, ngf=64, coord_net=True):
    """Ordered list of (tf_variable_name, shape) -- the parameter blob order of
    include/msi_hip.h (weights, gamma, beta | weights, biases)."""
    desc = make_desc(1, 8, 8, in_channels, num_outputs, ngf, coord_net)
    shapes = []
    for info in layer_infos(desc):
        name = info.name.decode()
        if info.kind == KIND_CONV:
            shapes.append((name + "/weights", (3, 3, info.cin + info.has_coord, info.cout)))
        elif info.kind == KIND_CONVT:
            shapes.append((name + "/weights", (4, 4, info.cout, info.cin)))
        else:
            shapes.append((name + "/weights", (1, 1, info.cin, info.cout)))
        if info.kind == KIND_HEAD:
            shapes.append((name + "/biases", (info.cout,)))
        else:
            shapes.append((name + "/LayerNorm/gamma", (info.cout,)))
            shapes.append((name + "/LayerNorm/beta", (info.cout,)))
    return shapes


def init_weights(in_channels, num_outputs, ngf=64, coord_net=True, seed=8964):
    """Random initialisation with slim's defaults: Xavier-uniform conv weights,
    LayerNorm gamma=1 / beta=0, zero head bias.  (There is no network access for
    the pretrained checkpoint; bench.py runs on these.)"""
    rng = np.random.RandomState(seed)
    w = {}
    for name, shape in variable_shapes(in_channels, num_outputs, ngf, coord_net):
        if name.endswith("/weights"):
            rf = shape[0] * shape[1]
            lim = math.sqrt(6.0 / (rf * shape[2] + rf * shape[3]))
            w[name] = rng.uniform(-lim, lim, size=shape).astype(np.float32)
        elif name.endswith("/gamma"):
            w[name] = np.ones(shape, np.float32)
        else:
            w[name] = np.zeros(shape, np.float32)
    return w


def _lookup(weights, name):
    for key in (name, "net/" + name, name + ":0", "net/" + name + ":0"):
        if key in weights:
            return weights[key]
    raise KeyError("missing network variable %r" % name)


def flatten_params(weights, in_channels, num_outputs, ngf=64, coord_net=True):
    """dict of TF-named arrays -> the flat fp32 parameter blob."""
    parts = []
    for name, shape in variable_shapes(in_channels, num_outputs, ngf, coord_net):
        a = _lookup(weights, name)
        if hasattr(a, "detach"):
            a = a.detach().cpu().numpy()
        a = np.asarray(a, dtype=np.float32)
        if tuple(a.shape) != tuple(shape):
            raise ValueError("variable %s has shape %s, expected %s" % (name, a.shape, shape))
        parts.append(a.reshape(-1))
    return np.ascontiguousarray(np.concatenate(parts))


def unflatten_params(blob, in_channels, num_outputs, ngf=64, coord_net=True):
    out, off = {}, 0
    for name, shape in variable_shapes(in_channels, num_outputs, ngf, coord_net):
        n = int(np.prod(shape))
        out[name] = np.asarray(blob[off:off + n], dtype=np.float32).reshape(shape).copy()
        off += n
    return out


def pack_params(desc, blob):
    """Parameter blob -> the MFMA-tile-ordered blob csrc/cnn.hip streams (host)."""
    blob = np.ascontiguousarray(blob, dtype=np.float32)
    need = N.lib.msi_net_param_floats(desc)
    if need == 0:
        raise N.MsiError("unsupported network descriptor: " + N.last_error())
    if blob.size != need:
        raise ValueError("parameter blob has %d floats, network needs %d" % (blob.size, need))
    packed = np.empty(N.lib.msi_net_packed_floats(desc), dtype=np.float32)
    N.check(N.lib.msi_net_pack_weights_host(desc, blob.ctypes.data, packed.ctypes.data),
            "msi_net_pack_weights_host")
    return packed
