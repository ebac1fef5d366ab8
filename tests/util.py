"""Shared synthetic inputs for the parity tests: band-limited noise
ODS pairs, identity poses, baseline 0.032, target position inside the unit sphere."""
import numpy as np


from matryodshka_amd.synthetic import smooth_noise, make_inputs, random_rgba, pp_inputs  # noqa: F401,E402


# ---- stratified dense sample set of the full-size fixtures (tests/golden/make_golden.py writes the values, the -m gpu
# tests regenerate the SAME indices from (shape, extra pixels, seed): only values and the oracle-derived pixels are stored)
def stratified_rows(h):
    """Eight full rows: both polar pairs, the equator pair, and one odd / one even row in between."""
    return sorted({0, 1, h // 2 - 1, h // 2, h - 2, h - 1, (h // 3) | 1, (2 * h // 3) & ~1})


def stratified_pixels(h, w, extra=None):
    """Sorted flat pixel indices y * w + x: eight full rows (polar rows included), the columns on both sides of every
    64-pixel tile seam (x % 64 in {0, 63}), and `extra` (the pixels the oracle marks disc < 0 on the far / near plane)."""
    mask = np.zeros((h, w), dtype=bool)
    mask[stratified_rows(h), :] = True
    mask[:, 0::64] = True
    mask[:, 63::64] = True
    pix = np.flatnonzero(mask.reshape(-1))
    if extra is not None and len(extra):
        pix = np.union1d(pix, np.asarray(extra, dtype=np.int64))
    return pix.astype(np.int64)


def stratified_index(shape, extra=None, seed=0, cap=262144):
    """Flat indices into a C-ordered array of `shape` = (B, H, W, *rest): every batch element, the stratified_pixels,
    and per pixel either all `rest` elements or -- when that exceeds `cap` values in total -- k seeded random ones
    (k = cap // pixels, at least 1).  At least `cap` values whenever the tensor has that many at the pixel set."""
    b, h, w = int(shape[0]), int(shape[1]), int(shape[2])
    rest = int(np.prod(shape[3:])) if len(shape) > 3 else 1
    pix = stratified_pixels(h, w, extra)
    base = (np.arange(b, dtype=np.int64)[:, None] * (h * w) + pix[None, :]).reshape(-1) * rest     # [B * P]
    if base.size * rest <= cap:
        ch = np.arange(rest, dtype=np.int64)[None, :]
        return (base[:, None] + ch).reshape(-1)
    k = max(1, -(-cap // base.size))
    rng = np.random.RandomState(seed)
    ch = rng.randint(0, rest, size=(base.size, k)).astype(np.int64)
    return (base[:, None] + ch).reshape(-1)


def count_equal_11(psv, pre_images, d, round_fn=None):
    """[2, B, D]: texels of the sweep volume whose three channels equal the source image's (1, 1) pixel -- what a
    pixel with a negative discriminant gathers (spherical.py:226-229: u = v = 1).  round_fn: the rounding the volume
    went through (bf16 volumes)."""
    b = psv.shape[0]
    out = np.zeros((2, b, d), dtype=np.int64)
    for s in range(2):
        for k in range(b):
            ref = pre_images[s][k, 1, 1, :].astype(np.float32)
            if round_fn is not None:
                ref = np.asarray(round_fn(ref), dtype=np.float32)
            for j in range(d):
                ch = s * 3 * d + 3 * j
                out[s, k, j] = int(np.all(psv[k, :, :, ch:ch + 3].astype(np.float32) == ref[None, None, :], axis=-1).sum())
    return out


def read_raw_output(ws, packed, info, batch, dtype):
    """A layer's raw (pre-LayerNorm) convolution output [B,H,W,C] as float32 numpy from a plan's workspace (torch uint8
    tensor): fp32 plans store fp32; bf16 plans store fp16 of x * 2^-e, with 2^e = 2^24 / S1 from the layer's LayerNorm
    window in the packed blob (include/msi_hip.h: msi_layer_info.ln_scale_offset)."""
    import torch
    n = batch * info.out_h * info.out_w * info.cout
    shape = (batch, info.out_h, info.out_w, info.cout)
    if dtype == "f32":
        return ws[info.raw_offset:info.raw_offset + 4 * n].view(torch.float32).reshape(shape).cpu().numpy()
    scl = packed[info.ln_scale_offset:info.ln_scale_offset + 8].cpu().numpy().view(np.float64)
    up = np.float32(16777216.0 / scl[0])
    return ws[info.raw_offset:info.raw_offset + 2 * n].view(torch.float16).reshape(shape).float().cpu().numpy() * up
