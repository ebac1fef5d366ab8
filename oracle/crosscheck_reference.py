#!/usr/bin/env python
"""Transcription cross-check of oracle/geometry.py against the REFERENCE'S OWN function bodies (test infrastructure;
build container only -- it reads /root/reference where it lies, copies nothing, and never travels to the GPU box).

    python oracle/crosscheck_reference.py [--reference /root/reference]

What it does: imports the reference's geometry/spherical.py, sampling.py, projector.py and homography.py unmodified, with
`tensorflow` and `tensorflow_graphics` replaced by the ~60-op numpy stand-in below, runs their entry points
(ods_sphere_sweep, projective_forward_sphere / _ods / _sphere_to_perspective, perspective_plane_sweep,
projective_forward_homography, over_composite, over_composite_depth) on small seeded inputs with NON-identity poses, and
asserts BIT-EQUALITY with the oracle's restatement of the same functions; and matryodshka/nets.py's msi_train_net /
msi_coord_train_net over a stand-in slim whose layer arithmetic is the oracle's own primitives (the WIRING is what is
checked there: layer order, skip concatenations, strides / rates, wrap_pad, LayerNorm before the [5:-5] crop); and
matryodshka/msi.py's MSI class end to end on one frame: inv_depths, preprocess_image, format_network_input, infer_msi with
the four which_color_pred schemes, msi_render_equirect_view / _depth, deprocess_image / deprocess_depth_image.

What it proves and what it does not: the stand-in evaluates every elementwise op in fp32 exactly as
the oracle assumes TensorFlow does (one numpy fp32 op per TF op, Python scalars converted to the tensor's dtype, matmul as a
k-ordered fp32 dot product, sin / cos as the correctly rounded fp32 value, tf.linspace as start + step * i) -- so it pins
NO TensorFlow semantics; parity stays "unpinned" (oracle/__init__.py).  What it removes is the one risk a reader of the
reference can remove: a mis-read association, operand order, branch, axis or index in the oracle's restatement -- the
expression trees executed here are the reference's own.
"""
import argparse
import contextlib
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
F = np.float32


# ---------------------------------------------------------------------------------------------------- numpy stand-in for tf
class _Shape(object):
    def __init__(self, s):
        self._s = tuple(int(v) for v in s)

    def as_list(self):
        return list(self._s)

    def __len__(self):
        return len(self._s)


class T(np.ndarray):
    """An fp32 / int32 array that answers the two tensor methods the reference calls."""

    def get_shape(self):
        return _Shape(self.shape)


def _t(x, dtype=None):
    a = np.asarray(x, dtype=dtype)
    if a.dtype == np.float64:
        a = a.astype(F)                       # tf.convert_to_tensor of Python floats: float32
    if a.dtype == np.int64:
        a = a.astype(np.int32)
    return a.view(T)


def _dt(d):
    return {"int32": np.int32, "float32": F}.get(d, d) if isinstance(d, str) else d


def _linspace(start, stop, num):
    start, stop = F(start), F(stop)
    if num == 1:
        return _t(np.array([start], dtype=F))
    step = F((stop - start) / F(num - 1))
    return _t((start + step * np.arange(num, dtype=F)).astype(F))


def _matmul(a, b, transpose_b=False, name=None):
    a, b = np.asarray(a), np.asarray(b)
    if transpose_b:
        b = np.swapaxes(b, -1, -2)
    acc = a[..., :, 0:1] * b[..., 0:1, :]
    for k in range(1, a.shape[-1]):
        acc = acc + a[..., :, k:k + 1] * b[..., k:k + 1, :]     # one fp32 rounding per product and per sum, k ascending
    return _t(acc)


def _add_n(xs):
    acc = xs[0]
    for x in xs[1:]:
        acc = acc + x
    return _t(acc)


def _round64(fn):
    return lambda x: _t(fn(np.asarray(x, dtype=np.float64)).astype(F))   # "correctly rounded fp32 sin / cos" (trig tables)


_GRAPH_TENSORS = {}


def make_tf():
    tf = types.ModuleType("tensorflow")
    tf.float32, tf.int32 = F, np.int32
    tf.reshape = lambda x, s, name=None: _t(np.reshape(x, [int(v) for v in s]))
    tf.expand_dims = lambda x, axis=None, **kw: _t(np.expand_dims(x, axis))
    tf.stack = lambda xs, axis=0, name=None: _t(np.stack([np.asarray(x) for x in xs], axis=axis))
    tf.concat = lambda xs, axis, name=None: _t(np.concatenate([np.asarray(x) for x in xs], axis=axis))
    tf.tile = lambda x, m: _t(np.tile(x, [int(v) for v in m]))
    tf.cast = lambda x, d: _t(np.asarray(x).astype(_dt(d)))
    tf.transpose = lambda x, perm=None: _t(np.transpose(x, perm))
    tf.matrix_transpose = lambda x: _t(np.swapaxes(x, -1, -2))
    tf.square = lambda x: _t(np.asarray(x) * np.asarray(x))
    tf.ones_like = lambda x: _t(np.ones_like(x))
    tf.zeros_like = lambda x: _t(np.zeros_like(x))
    tf.zeros = lambda s, dtype=F: _t(np.zeros([int(v) for v in s], dtype=_dt(dtype)))
    tf.ones = lambda s, dtype=F: _t(np.ones([int(v) for v in s], dtype=_dt(dtype)))
    tf.eye = lambda n: _t(np.eye(n, dtype=F))
    tf.where = lambda c, a, b: _t(np.where(c, a, b))
    tf.matmul = _matmul
    tf.sin, tf.cos = _round64(np.sin), _round64(np.cos)
    tf.sqrt = lambda x: _t(np.sqrt(x))
    tf.abs = lambda x: _t(np.abs(x))
    tf.sign = lambda x: _t(np.sign(x))
    tf.floor = lambda x: _t(np.floor(x))
    tf.atan2 = lambda y, x: _t(np.arctan2(y, x))
    tf.is_nan = lambda x: np.isnan(x)
    tf.greater = lambda a, b: np.greater(a, b)
    tf.greater_equal = lambda a, b: np.greater_equal(a, b)
    tf.less_equal = lambda a, b: np.less_equal(a, b)
    tf.equal = lambda a, b: _t(np.equal(a, b))
    tf.divide = lambda a, b, name=None: _t(np.asarray(a) / np.asarray(b))
    tf.mod = lambda a, n: _t(np.mod(a, n))                       # floor-mod on int32 [TF-knowledge]
    tf.linspace = _linspace
    tf.meshgrid = lambda a, b: [_t(v) for v in np.meshgrid(a, b)]
    tf.range = lambda n: _t(np.arange(n, dtype=np.int32))
    tf.gather_nd = lambda p, idx: _t(np.asarray(p)[tuple(np.asarray(idx)[:, k] for k in range(np.asarray(idx).shape[1]))])
    tf.slice = lambda x, begin, size: _t(np.asarray(x)[tuple(slice(int(b), int(b) + int(s)) for b, s in zip(begin, size))])
    tf.unstack = lambda x, axis=0: [_t(v) for v in np.moveaxis(np.asarray(x), axis, 0)]
    tf.add_n = _add_n
    tf.convert_to_tensor = lambda x, dtype=None: _t(x, _dt(dtype))
    tf.constant = lambda v, shape=None, dtype=None: _t(np.reshape(np.asarray(v, dtype=F), shape) if shape else np.asarray(v, dtype=F))
    tf.is_tensor = lambda x: isinstance(x, np.ndarray)
    tf.name_scope = lambda name: contextlib.nullcontext()
    tf.get_default_graph = lambda: types.SimpleNamespace(get_tensor_by_name=lambda n: _GRAPH_TENSORS[n])
    tf.app = types.SimpleNamespace(flags=types.SimpleNamespace(FLAGS=types.SimpleNamespace()))
    from oracle import geometry as G     # tf.contrib.resampler is a C++ op: the oracle's own statement of it stands in
    tf.contrib = types.SimpleNamespace(resampler=types.SimpleNamespace(
        resampler=lambda imgs, coords: _t(G.resampler_zero_pad(np.asarray(imgs), np.asarray(coords)))))
    tf.random = types.SimpleNamespace(uniform=None)
    # --- what matryodshka/nets.py touches besides the above
    tf.nn = types.SimpleNamespace(relu="relu", tanh="tanh")
    tf.app.flags.FLAGS.net_only = False
    tf.app.flags.FLAGS.which_color_pred = "blend_psv"
    tf.variable_scope = lambda name, reuse=None: contextlib.nullcontext()
    tf.shape = lambda x: [int(v) for v in np.asarray(x).shape]
    tf.identity = lambda x, name=None: x
    tf.Tensor = np.ndarray

    def pad(x, paddings, mode="CONSTANT"):
        assert mode == "CONSTANT"
        return _t(np.pad(np.asarray(x), [(int(a), int(b)) for a, b in paddings]))
    tf.pad = pad

    # --- what matryodshka/msi.py touches besides the above
    def convert_image_dtype(image, dtype, saturate=False):
        """tf.image.convert_image_dtype [TF-knowledge]: float -> float passes through; uint8 -> float multiplies by
        1 / 255 (a Python double converted to fp32); float -> uint8 = cast(image * 255.5) without saturation (the inputs of
        this check stay in range, where the cast is defined)."""
        a = np.asarray(image)
        if a.dtype == np.uint8 and dtype == F:
            return _t(a.astype(F) * (1.0 / 255.0))
        if a.dtype == F and dtype == np.uint8:
            return (a * 255.5).astype(np.uint8).view(T)
        assert a.dtype == dtype, (a.dtype, dtype)
        return _t(a)
    tf.uint8 = np.uint8
    tf.image = types.SimpleNamespace(convert_image_dtype=convert_image_dtype)
    fl = tf.app.flags.FLAGS
    fl.supervision, fl.input_type, fl.operation, fl.transform_inverse_reg, fl.jitter = "", "ODS", "train", False, False
    return tf


class SlimStandIn(object):
    """tensorflow.contrib.slim as far as msi_train_net / msi_coord_train_net use it: arg_scope (the normalizer default),
    conv2d, conv2d_transpose, layer_norm.  The ARITHMETIC of each layer is the oracle's own primitive (torch-CPU conv with
    the oracle's SAME-padding split, the oracle's LayerNorm + ReLU): what is checked is the WIRING the reference's
    function body performs -- layer order, inputs and skip concatenation order, strides / rates, wrap_pad amounts, the
    [5:-5] crop after the LayerNorm, the coordinate channel, padding modes."""
    layer_norm = "layer_norm"

    def __init__(self):
        self.weights = None
        self.default_norm = None
        self.calls = []

    @contextlib.contextmanager
    def arg_scope(self, ops, normalizer_fn=None):
        old, self.default_norm = self.default_norm, normalizer_fn
        try:
            yield
        finally:
            self.default_norm = old

    _UNSET = object()

    def _finish(self, y, scope, activation_fn, normalizer_fn):
        import torch
        from oracle import nets as onets
        w = self.weights
        if normalizer_fn is None:
            y = y + torch.from_numpy(w[scope + "/biases"]).view(1, -1, 1, 1)
        else:
            assert normalizer_fn == "layer_norm" and activation_fn == "relu"
            return onets.layer_norm_relu(y, w[scope + "/LayerNorm/gamma"], w[scope + "/LayerNorm/beta"])
        return torch.tanh(y) if activation_fn == "tanh" else (torch.relu(y) if activation_fn == "relu" else y)

    def conv2d(self, inputs, num_outputs, kernel_size, stride=1, padding="SAME", rate=1, activation_fn="relu",
               normalizer_fn=_UNSET, scope=None):
        import torch
        import torch.nn.functional as TF
        from oracle import nets as onets
        if normalizer_fn is SlimStandIn._UNSET:
            normalizer_fn = self.default_norm
        self.calls.append((scope, "conv", tuple(np.asarray(inputs).shape), stride, rate, padding))
        wt = self.weights[scope + "/weights"]
        assert tuple(wt.shape) == (kernel_size[0], kernel_size[1], np.asarray(inputs).shape[3], num_outputs), (scope, wt.shape)
        x = torch.from_numpy(np.ascontiguousarray(np.transpose(np.asarray(inputs), (0, 3, 1, 2))))
        with torch.no_grad():
            if padding == "SAME" and kernel_size[0] > 1:
                pt, pb = onets._same_pad(x.shape[2], 2 * rate + 1, stride)
                pl, pr = onets._same_pad(x.shape[3], 2 * rate + 1, stride)
                x = TF.pad(x, (pl, pr, pt, pb))
            if normalizer_fn is None:      # (slim adds `biases` only without a normalizer; the oracle hands them to the conv primitive)
                y = TF.conv2d(x, onets._conv_w(wt), bias=torch.from_numpy(self.weights[scope + "/biases"]), stride=stride, dilation=rate)
                y = torch.tanh(y) if activation_fn == "tanh" else (torch.relu(y) if activation_fn == "relu" else y)
            else:
                y = self._finish(TF.conv2d(x, onets._conv_w(wt), stride=stride, dilation=rate), scope, activation_fn, normalizer_fn)
        return _t(np.ascontiguousarray(y.permute(0, 2, 3, 1).numpy()))

    def conv2d_transpose(self, inputs, num_outputs, kernel_size, stride=1, padding="SAME", activation_fn="relu",
                         normalizer_fn=_UNSET, scope=None):
        import torch
        import torch.nn.functional as TF
        from oracle import nets as onets
        if normalizer_fn is SlimStandIn._UNSET:
            normalizer_fn = self.default_norm
        self.calls.append((scope, "convT", tuple(np.asarray(inputs).shape), stride, 1, padding))
        wt = self.weights[scope + "/weights"]
        assert list(kernel_size) == [4, 4] and stride == 2
        x = torch.from_numpy(np.ascontiguousarray(np.transpose(np.asarray(inputs), (0, 3, 1, 2))))
        with torch.no_grad():
            y = TF.conv_transpose2d(x, onets._convT_w(wt), stride=2, padding=1 if padding == "SAME" else 0)
            y = self._finish(y, scope, activation_fn, normalizer_fn)
        return _t(np.ascontiguousarray(y.permute(0, 2, 3, 1).numpy()))


def make_tfgt():
    """tensorflow_graphics.geometry.transformation.rotation_matrix_3d.from_euler (third-party, not vendored: its
    published formula, R = Rz Ry Rx for angles (x, y, z))."""
    def from_euler(angles):
        a = np.asarray(angles, dtype=F).reshape(-1, 3)
        sx, sy, sz = (np.sin(a[:, k].astype(np.float64)).astype(F) for k in range(3))
        cx, cy, cz = (np.cos(a[:, k].astype(np.float64)).astype(F) for k in range(3))
        m = np.stack([cy * cz, (sx * sy * cz) - (cx * sz), (cx * sy * cz) + (sx * sz),
                      cy * sz, (sx * sy * sz) + (cx * cz), (cx * sy * sz) - (sx * cz),
                      -sy, sx * cy, cx * cy], axis=-1).reshape(-1, 3, 3)
        return _t(m)
    tr = types.ModuleType("tensorflow_graphics.geometry.transformation")
    tr.rotation_matrix_3d = types.SimpleNamespace(from_euler=from_euler)
    return tr


def load_reference(root):
    tf = make_tf()
    sys.modules["tensorflow"] = tf
    for name in ("tensorflow_graphics", "tensorflow_graphics.geometry"):
        sys.modules[name] = types.ModuleType(name)
    sys.modules["tensorflow_graphics.geometry.transformation"] = make_tfgt()
    slim = SlimStandIn()
    contrib = types.ModuleType("tensorflow.contrib")
    contrib.slim = slim
    contrib.resampler = tf.contrib.resampler
    tf.contrib = contrib
    sys.modules["tensorflow.contrib"] = contrib
    # projector.py uses Python 2 implicit relative imports (`import homography`); homography.py says `import geometry.sampling`
    sys.path.insert(0, os.path.join(root, "geometry"))
    sys.path.insert(0, root)
    import projector                                            # noqa: E402  (the reference's file, where it lies)
    import sampling                                             # noqa: E402
    import spherical                                            # noqa: E402
    sys.path.insert(0, os.path.join(root, "matryodshka"))
    import nets as refnets                                      # noqa: E402  (matryodshka/nets.py)
    # matryodshka/msi.py: its two imports the path does not need are empty modules (losses: elpips; PNG writing: utils)
    for name in ("elpips", "elpips.elpips"):
        sys.modules[name] = types.ModuleType(name)
    sys.modules["elpips"].elpips = sys.modules["elpips.elpips"]
    pkg = types.ModuleType("matryodshka")
    pkg.__path__ = [os.path.join(root, "matryodshka")]
    sys.modules["matryodshka"] = pkg
    utils = types.ModuleType("matryodshka.utils")
    utils.write_image = None
    sys.modules["matryodshka.utils"] = utils
    import importlib
    refmsi = importlib.import_module("matryodshka.msi")
    return projector, sampling, spherical, refnets, slim, refmsi, tf


# ---------------------------------------------------------------------------------------------------- inputs and checks
def rigid(rng, ang=0.2, tr=0.05):
    a, b, c = rng.uniform(-ang, ang, size=3)
    rx = np.array([[1, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]])
    ry = np.array([[np.cos(b), 0, np.sin(b)], [0, 1, 0], [-np.sin(b), 0, np.cos(b)]])
    rz = np.array([[np.cos(c), -np.sin(c), 0], [np.sin(c), np.cos(c), 0], [0, 0, 1]])
    m = np.eye(4)
    m[:3, :3] = rz @ ry @ rx
    m[:3, 3] = rng.uniform(-tr, tr, size=3)
    return m.astype(F)


def same(name, got, want):
    got, want = np.asarray(got), np.asarray(want)
    assert got.shape == want.shape, (name, got.shape, want.shape)
    ok = np.array_equal(got.astype(F).view(np.uint32), want.astype(F).view(np.uint32)) or np.array_equal(got, want, equal_nan=True)
    n = int((~((got == want) | (np.isnan(got) & np.isnan(want)))).sum())
    print("%-46s %-22s %s" % (name, "x".join(str(v) for v in got.shape), "bit-identical" if ok else "MISMATCH in %d values" % n))
    return ok


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    args = ap.parse_args()
    if not os.path.isdir(os.path.join(args.reference, "geometry")):
        print("no reference checkout at %s (this check runs in the build container only)" % args.reference)
        return 2
    from oracle import geometry as G
    projector, sampling, spherical, refnets, slim, refmsi, tf = load_reference(args.reference)
    rng = np.random.RandomState(20260928)
    b, h, w, d = 2, 12, 24, 5
    img = rng.uniform(-1, 1, size=(b, h, w, 3)).astype(F)
    depths = [float(x) for x in (1.0 / np.linspace(1.0 / 100.0, 1.0, d))]           # far -> near
    poses = np.stack([rigid(rng) for _ in range(b)])
    poses[1] = np.eye(4, dtype=F)                                                   # one identity pose: the test path
    intr = np.tile(np.array([[0.032, 0, 0], [0, 1, 0], [0, 0, 1]], F)[None], (b, 1, 1))
    intr[1, 0, 0] = 0.05
    ok = True
    np.seterr(all="ignore")

    # --- K1: ods_sphere_sweep (projector.py:209-211 -> sweep_one :129-170; spherical.py:116-129, 170-233; sampling.py:135-197)
    # (the reference reads intrinsics[0][0][0] of the tiled PER-SAMPLE slice: the sample's own baseline)
    for order in (1, -1):
        ref = projector.ods_sphere_sweep(_t(img), order, depths, _t(poses), _t(intr))
        ok &= same("ods_sphere_sweep order %+d" % order, ref, G.ods_sphere_sweep(img, order, depths, poses, intr))

    # --- K4: projective_forward_sphere + over_composite(_depth) (projector.py:34-62, 225-265; spherical.py:268-326)
    layers = rng.uniform(-1, 1, size=(d, b, h, w, 4)).astype(F)
    layers[..., 3] = rng.uniform(0, 1, size=(d, b, h, w)).astype(F)
    tgt_pos = rng.uniform(-0.1, 0.1, size=(b, 3)).astype(F)
    tgt_pose = np.stack([rigid(rng, 0.1, 0.02) for _ in range(b)])
    dep_lb = np.tile(np.asarray(depths, F).reshape(-1, 1), (1, b))
    ref = projector.projective_forward_sphere(_t(layers), None, _t(tgt_pose), _t(tgt_pos), _t(dep_lb))
    mine = G.projective_forward_sphere(layers, tgt_pose, tgt_pos, dep_lb)
    ok &= same("projective_forward_sphere", ref, mine)
    ok &= same("over_composite", projector.over_composite([_t(ref[i]) for i in range(d)]), G.over_composite([mine[i] for i in range(d)]))
    ok &= same("over_composite_depth", projector.over_composite_depth([_t(ref[i]) for i in range(d)]),
               G.over_composite_depth([mine[i] for i in range(d)]))

    # --- msi_render_ods_view (projector.py:100-127; spherical.py:328-365): the reference reads intrinsics[0][0][0] for
    # EVERY sample (the batch's first baseline): compare sample by sample with batch 1, as the oracle documents it
    for order in (1, -1):
        for k in range(b):
            ref = projector.projective_forward_ods(_t(layers[:, k:k + 1]), order, _t(intr[k:k + 1]), _t(tgt_pose[k:k + 1]),
                                                   _t(tgt_pos[k:k + 1]), _t(dep_lb[:, k:k + 1]))
            mine = G.projective_forward_ods(layers[:, k:k + 1], order, intr[k:k + 1], tgt_pose[k:k + 1], dep_lb[:, k:k + 1])
            ok &= same("projective_forward_ods order %+d sample %d" % (order, k), ref, mine)

    # --- msi_render_perspective_view (projector.py:64-98; spherical.py:367-401)
    for vw in (0, 3):
        ref = projector.projective_forward_sphere_to_perspective(_t(layers), None, _t(tgt_pose), _t(tgt_pos), _t(dep_lb),
                                                                 viewing_window=vw, tgt_height=9, tgt_width=16)
        mine = G.projective_forward_sphere_to_perspective(layers, tgt_pos, dep_lb, vw, 9, 16)
        ok &= same("projective_forward_sphere_to_perspective vw=%d" % vw, ref, mine)

    # --- PP path: perspective_plane_sweep (projector.py:221-223) and projective_forward_homography (:343-373, homography.py)
    n = 16
    K = np.tile(np.array([[n / 2, 0, n / 2], [0, n / 2, n / 2], [0, 0, 1]], F)[None], (b, 1, 1))
    face = rng.uniform(-1, 1, size=(b, n, n, 3)).astype(F)
    pp_pose = np.stack([rigid(rng, 0.05, 0.05) for _ in range(b)])
    ref = projector.perspective_plane_sweep(_t(face), 1, depths, _t(pp_pose), _t(K))
    ok &= same("perspective_plane_sweep", ref, G.perspective_plane_sweep(face, depths, pp_pose, K))
    Kinv = np.linalg.inv(K.astype(np.float64)).astype(F)
    _GRAPH_TENSORS["intrinsics_inv:0"] = _t(Kinv)
    players = rng.uniform(-1, 1, size=(d, b, n, n, 4)).astype(F)
    ref = projector.projective_forward_homography(_t(players), _t(K), _t(pp_pose), _t(dep_lb))
    ok &= same("projective_forward_homography", ref, G.projective_forward_homography(players, K, Kinv, pp_pose, dep_lb))

    # --- the two networks (matryodshka/nets.py:387-450 msi_train_net, :471-515 msi_coord_train_net): the reference's function
    # bodies drive the stand-in slim ops (the oracle's own primitives), the oracle's forward() is the restatement of the wiring
    from oracle import nets as onets
    for coord, fn in ((False, refnets.msi_train_net), (True, refnets.msi_coord_train_net)):
        cin, nout, ngf = 12, 4, 8
        x = rng.uniform(-1, 1, size=(2, 16, 32, cin)).astype(F)
        slim.weights = onets.init_weights(cin, nout, ngf=ngf, coord_net=coord, seed=77, randomize_affine=True)
        slim.calls = []
        ref = fn(_t(x), nout, ngf=ngf)
        ok &= same("%s (%d layers)" % (fn.__name__, len(slim.calls)), ref, onets.forward(slim.weights, x, coord_net=coord))
        assert [c[0] for c in slim.calls] == [t[0] for t in onets.layer_table(cin, nout, ngf, coord)]

    # --- the MSI class itself (matryodshka/msi.py): inv_depths, preprocess, format_network_input (curr_pose = pose @ ref_pose_inv,
    # order +1 / -1), infer_msi's four colour schemes, the equirect RGB / depth render, deprocess -- one frame (the reference
    # needs B = 1, msi.py:1109-1110), non-identity ref / src poses, uint8 images
    from oracle.msi import MSI as OracleMSI
    d, ngf, h, w = 4, 8, 16, 32
    ref8 = rng.randint(0, 256, size=(1, h, w, 3)).astype(np.uint8)
    src8 = rng.randint(0, 256, size=(1, h, w, 3)).astype(np.uint8)
    ref_pose, src_pose = rigid(rng, 0.05, 0.02)[None], rigid(rng, 0.05, 0.02)[None]
    tpose, tpos = rigid(rng, 0.1, 0.02)[None], rng.uniform(-0.1, 0.1, size=(1, 3)).astype(F)
    intr1 = intr[:1]
    ref_pose_inv = np.linalg.inv(ref_pose.astype(np.float64)).astype(F)
    _GRAPH_TENSORS["ref_pose_inv:0"] = _t(ref_pose_inv)
    m = refmsi.MSI()
    assert m.inv_depths(1.0, 100.0, 32) == OracleMSI().inv_depths(1.0, 100.0, 32)
    planes = m.inv_depths(1.0, 100.0, d)
    print("%-46s %-22s %s" % ("MSI.inv_depths(1, 100, 32)", "32", "identical (Python floats)"))
    for coord in (True, False):
        for scheme, nout in (("blend_psv", 2 * d), ("blend_bg", 2 * d + 3), ("blend_bg_psv", 3 * d + 3), ("alpha_only", d)):
            fl = tf.app.flags.FLAGS
            fl.coord_net, fl.which_color_pred, fl.ngf = coord, scheme, ngf
            slim.weights = onets.init_weights(6 * d, nout, ngf=ngf, coord_net=coord, seed=5, randomize_affine=True)
            # (the reference itself raises UnboundLocalError for blend_bg + 'blend_weights': msi.py:283-284 asks for
            # bg_blend_weights whenever 'bg' is in the scheme's name; the oracle / product return what exists)
            extra = "alphas psv" if scheme == "blend_bg" else "blend_weights alphas psv"
            pred, net_in = m.infer_msi(_t(src8), _t(ref8), None, None, _t(ref_pose), _t(src_pose), _t(intr1), scheme, d, planes,
                                       extra_outputs=extra, ngf=ngf)
            o = OracleMSI(weights=slim.weights, coord_net=coord)
            pred_o, net_o = o.infer_msi(src8, ref8, None, None, ref_pose, src_pose, intr1, scheme, d, planes, extra_outputs=extra,
                                        ngf=ngf, ref_pose_inv=ref_pose_inv)
            tag = "infer_msi %s %s" % ("coord" if coord else "wrap", scheme)
            ok &= same(tag + " net_input", net_in, net_o)
            assert sorted(pred) == sorted(pred_o), (sorted(pred), sorted(pred_o))
            for k in sorted(pred):
                ok &= same(tag + " " + k, pred[k], pred_o[k])
            if scheme == "blend_psv":
                rgb = m.msi_render_equirect_view(pred["rgba_layers"], _t(tpose), _t(tpos), planes, _t(intr1))
                dep = m.msi_render_equirect_depth(pred["rgba_layers"], _t(tpose), _t(tpos), planes, _t(intr1))
                rgb_o = o.msi_render_equirect_view(pred_o["rgba_layers"], tpose, tpos, planes, intr1)
                dep_o = o.msi_render_equirect_depth(pred_o["rgba_layers"], tpose, tpos, planes, intr1)
                ok &= same(tag + " render rgb", rgb, rgb_o)
                ok &= same(tag + " render depth", dep, dep_o)
                ok &= bool(np.array_equal(np.asarray(m.deprocess_image(rgb)), o.deprocess_image(rgb_o)))
                ok &= bool(np.array_equal(np.asarray(m.deprocess_depth_image(dep)), o.deprocess_depth_image(dep_o)))
                print("%-46s %-22s %s" % (tag + " deprocess (uint8)", "1x%dx%dx3 x2" % (h, w), "identical" if ok else "MISMATCH"))

    print("RESULT:", "the oracle's geometry, network-wiring and MSI-class restatement is bit-identical to the reference's own expression trees under the stand-in"
          if ok else "MISMATCH -- see above")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
