"""Seeded synthetic inputs of the hot path: band-limited noise ODS pairs, identity poses,
baseline 0.032, target position inside the unit sphere; PP cube-face pairs.  Used by bench.py, the tools and the
tests (tests/util.py re-exports them) -- no network, no datasets in the image."""
import numpy as np


def smooth_noise(rng, b, h, w, c=3, factor=8):
    """uniform noise at (h/factor, w/factor) bilinearly upsampled to (h, w), in [0,1]."""
    lh, lw = max(h // factor, 2), max(w // factor, 2)
    low = rng.uniform(0.0, 1.0, size=(b, lh, lw, c))
    ys = np.linspace(0, lh - 1, h)
    xs = np.linspace(0, lw - 1, w)
    y0 = np.floor(ys).astype(int); y1 = np.minimum(y0 + 1, lh - 1); fy = (ys - y0)[None, :, None, None]
    x0 = np.floor(xs).astype(int); x1 = np.minimum(x0 + 1, lw - 1); fx = (xs - x0)[None, None, :, None]
    top = low[:, y0][:, :, x0] * (1 - fx) + low[:, y0][:, :, x1] * fx
    bot = low[:, y1][:, :, x0] * (1 - fx) + low[:, y1][:, :, x1] * fx
    return (top * (1 - fy) + bot * fy).astype(np.float32)


def make_inputs(seed, b, h, w, as_uint8=True):
    rng = np.random.RandomState(seed)
    ref = smooth_noise(rng, b, h, w)
    src = smooth_noise(rng, b, h, w)
    if as_uint8:
        ref = np.clip(np.round(ref * 255), 0, 255).astype(np.uint8)
        src = np.clip(np.round(src * 255), 0, 255).astype(np.uint8)
    pose = np.tile(np.eye(4, dtype=np.float32)[None], (b, 1, 1))
    intr = np.tile(np.array([[0.032, 0, 0], [0, 1, 0], [0, 0, 1]], dtype=np.float32)[None], (b, 1, 1))
    tgt_pos = rng.uniform(-0.1, 0.1, size=(b, 3)).astype(np.float32)
    return dict(ref_image=ref, src_image=src, ref_pose=pose, src_pose=pose.copy(), intrinsics=intr,
                tgt_pos=tgt_pos, tgt_pose_rt=np.tile(np.eye(4, dtype=np.float32)[None], (b, 1, 1)))


def random_rgba(seed, b, h, w, d):
    rng = np.random.RandomState(seed)
    rgba = np.empty((b, h, w, d, 4), dtype=np.float32)
    for i in range(d):
        x = smooth_noise(rng, b, h, w, 4, factor=4)
        rgba[..., i, :3] = x[..., :3] * 2 - 1
        rgba[..., i, 3] = x[..., 3]
    return rgba


def pp_inputs(seed, b, n):
    """data_loader.py:205-226 (input_type PP): fx = cx = W/2, fy = cy = H/2; source shifted along -x by the input
    offset, target by the target offset.  Returns (ref, src, K, identity poses, src_pose, tgt_pose)."""
    rng = np.random.RandomState(seed)
    ref, src = smooth_noise(rng, b, n, n), smooth_noise(rng, b, n, n)
    K = np.tile(np.array([[n / 2, 0, n / 2], [0, n / 2, n / 2], [0, 0, 1]], np.float32)[None], (b, 1, 1))
    eye = np.tile(np.eye(4, dtype=np.float32)[None], (b, 1, 1))
    src_pose = eye.copy(); src_pose[:, 0, 3] = -0.064
    tgt_pose = eye.copy(); tgt_pose[:, 0, 3] = -0.03; tgt_pose[:, 1, 3] = 0.01
    return ref, src, K, eye, src_pose, tgt_pose
