#!/usr/bin/env python
"""Generates the golden fixtures under tests/golden/ from the CPU oracle.

The reference (Python 2 + TensorFlow 1.14) cannot be imported or run in the build container,
and it ships no vectors of its own, so these are REGRESSION vectors of the oracle (pinned by the
analytic KATs in tests/test_oracle_kat.py), not outputs of the reference binary.

    python tests/golden/make_golden.py            # small set (seconds)
    python tests/golden/make_golden.py --no-small --configs config1 wrap config2 config3 config4
                                                  # full-size fixtures: dense stratified samples (>= 262 144 values per
                                                  # stage) of BASELINE configs[1..4] and of the wrap-pad network (minutes each)
"""
import argparse
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import nets as onets  # noqa: E402
from oracle.msi import MSI as OracleMSI  # noqa: E402
from tests.util import make_inputs  # noqa: E402


def run(seed, b, h, w, d, ngf, coord):
    inp = make_inputs(seed, b, h, w)
    weights = onets.init_weights(6 * d, 2 * d, ngf=ngf, coord_net=coord, seed=seed, randomize_affine=True)
    o = OracleMSI(weights=weights, coord_net=coord)
    planes = o.inv_depths(1.0, 100.0, d)
    pred, net_input = o.infer_msi(inp["src_image"], inp["ref_image"], None, None, inp["ref_pose"], inp["src_pose"],
                                  inp["intrinsics"], "blend_psv", d, planes, extra_outputs="blend_weights alphas", ngf=ngf)
    rgb = o.msi_render_equirect_view(pred["rgba_layers"], inp["tgt_pose_rt"], inp["tgt_pos"], planes, inp["intrinsics"])
    dep = o.msi_render_equirect_depth(pred["rgba_layers"], inp["tgt_pose_rt"], inp["tgt_pos"], planes, inp["intrinsics"])
    return inp, dict(psv=net_input, rgba_layers=pred["rgba_layers"], blend_weights=pred["blend_weights"],
                     alphas=pred["alphas"], rgb=rgb, depth=dep, rgb_u8=o.deprocess_image(rgb),
                     depth_u8=o.deprocess_depth_image(dep))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--full", action="store_true", help="alias of --configs config1")
    ap.add_argument("--configs", nargs="*", default=[], choices=["config1", "wrap", "config2", "config3", "config4", "hres4k"],
                    help="full-size dense stratified fixtures: config1 = BASELINE configs[1] (640x320x32, CoordNet), wrap = the "
                         "same frame through msi_train_net (test.py:52's default network), config2 / 3 / 4 = BASELINE "
                         "configs[2..4] (minutes of CPU each)")
    ap.add_argument("--no-small", action="store_true", help="do not rewrite the small fixtures")
    a = ap.parse_args()
    if not a.no_small:
        for name, cfg in (("small_coord", dict(seed=11, b=1, h=16, w=32, d=4, ngf=8, coord=True)),
                          ("small_wrap", dict(seed=12, b=2, h=16, w=40, d=4, ngf=8, coord=False))):
            inp, out = run(**cfg)
            np.savez_compressed(os.path.join(HERE, name + ".npz"), cfg=np.array(sorted(cfg.items()), dtype=object),
                                **{"in_" + k: v for k, v in inp.items()}, **{"out_" + k: v for k, v in out.items()})
            print("wrote", name, {k: v.shape for k, v in out.items()})
    todo = list(a.configs) + (["config1"] if a.full and "config1" not in a.configs else [])
    for name in todo:
        {"config1": full_config1, "wrap": full_wrap, "config2": full_config2, "config3": full_config3,
         "config4": full_config4, "hres4k": full_hres4k}[name]()


# ---- full-size fixtures: dense stratified samples ------------------------------------------------------------------
# Per stage >= 262 144 values (tests.util.stratified_index): all texels of eight full rows (the polar rows included),
# the columns on both sides of every 64-pixel tile seam, and every pixel the oracle marks disc < 0 (project_ods'
# invalid-pixel rule, spherical.py:226-229) on the far and the near plane; channels per pixel are subsampled (seeded)
# where a stage has more than that.  Only the VALUES and the oracle-derived pixel list are stored; the tests rebuild
# the index from (shape, pixels, seed).  Also stored: the per-(source, sample, plane) count of invalid pixels and of
# sweep-volume texels equal to the (1, 1) sample (what an invalid pixel gathers) -- the GPU volume must reproduce both.
def ods_invalid(o, inp, planes):
    """valid masks of both sweeps, [2, B, D, H, W] bool (format_network_input's poses: msi.py:1124-1129)."""
    from oracle import geometry as G
    from oracle.msi import matmul4
    ref_pose, src_pose = np.asarray(inp["ref_pose"], np.float32), np.asarray(inp["src_pose"], np.float32)
    b = ref_pose.shape[0]
    h, w = inp["ref_image"].shape[1:3]
    ref_pose_inv = np.linalg.inv(ref_pose.astype(np.float64)).astype(np.float32)
    S, T = G.lat_long_grid((h, w))
    depths = np.asarray(planes, np.float32)
    valid = np.empty((2, b, len(planes), h, w), dtype=bool)
    for i, pose in enumerate((ref_pose, src_pose)):
        cur = matmul4(pose, ref_pose_inv)
        for k in range(b):
            pts = G.apply_pose(G.backproject_spherical(S, T, depths), cur[k])
            _, _, v = G.project_ods(pts, 1 if i == 0 else -1, inp["intrinsics"][k, 0, 0], w, h)
            valid[i, k] = v
    return valid


def dense_samples(out, keys, extra, seed):
    from tests.util import stratified_index
    s = {}
    for k in keys:
        a = np.asarray(out[k])
        idx = stratified_index(a.shape, extra, seed)
        s["val_" + k] = a.reshape(-1)[idx]
        s["shape_" + k] = np.array(a.shape, dtype=np.int64)
        s["mean_" + k] = np.float64(a.astype(np.float64).mean())
    return s


def ods_fixture(o, inp, out, planes, d, seed, keys=("psv", "rgba_layers", "rgb", "depth"), round_fn=None):
    valid = ods_invalid(o, inp, planes)
    h, w = valid.shape[-2:]
    extra = np.flatnonzero((~valid[:, :, [0, d - 1]]).any(axis=(0, 1, 2)).reshape(-1))    # disc < 0 on the far / near plane
    s = dense_samples(out, keys, extra, seed)
    s["extra_pixels"] = extra.astype(np.int32)
    s["sample_seed"] = np.int64(seed)
    s["invalid_count"] = (~valid).sum(axis=(3, 4)).astype(np.int64)                        # [2, B, D]
    pre = (o.preprocess_image(inp["ref_image"]), o.preprocess_image(inp["src_image"]))
    from tests.util import count_equal_11
    s["equal11_count"] = count_equal_11(out["psv"], pre, d, round_fn)
    assert (s["equal11_count"] >= s["invalid_count"]).all()
    return s


def full_config1():
    cfg = dict(seed=8964, b=1, h=320, w=640, d=32, ngf=64, coord=True)
    inp, out = run(**cfg)
    o = OracleMSI(coord_net=True)
    planes = o.inv_depths(1.0, 100.0, cfg["d"])
    np.savez_compressed(os.path.join(HERE, "full_640x320x32_samples.npz"),
                        cfg=np.array(sorted(cfg.items()), dtype=object), **ods_fixture(o, inp, out, planes, cfg["d"], seed=1))
    print("wrote config1 samples")


def full_wrap():
    """The reference's DEFAULT network (test.py:52 coord_net=False -> msi_train_net, nets.py:387-450) at the BASELINE
    size: the stages of the frame, plus per layer the LayerNorm affine (scale | shift, fp64 two-pass statistics --
    conv6_1 / conv7_1 / conv8_1 over the uncropped (2H+10) x (2W+10) output, nets.py:423-435) and 4 096 samples of the
    raw convolution output."""
    cfg = dict(seed=8965, b=1, h=320, w=640, d=32, ngf=64, coord=False)
    inp, out = run(**cfg)
    o = OracleMSI(coord_net=False)
    d = cfg["d"]
    planes = o.inv_depths(1.0, 100.0, d)
    s = ods_fixture(o, inp, out, planes, d, seed=6)
    weights = onets.init_weights(6 * d, 2 * d, ngf=cfg["ngf"], coord_net=False, seed=cfg["seed"], randomize_affine=True)
    pred, acts = onets.forward(weights, out["psv"], coord_net=False, return_activations=True)
    rng = np.random.RandomState(7)
    for name in [k[:-len("/affine")] for k in acts if k.endswith("/affine")]:
        s["affine_" + name] = acts[name + "/affine"]
        raw = acts[name + "/raw"].reshape(-1)
        idx = rng.randint(0, raw.size, size=4096)
        s["rawidx_" + name] = idx
        s["rawval_" + name] = raw[idx]
        s["rawscale_" + name] = np.float64(np.abs(raw).max())
    idx = rng.randint(0, pred.size, size=65536)
    s["predidx"] = idx
    s["predval"] = pred.reshape(-1)[idx]
    np.savez_compressed(os.path.join(HERE, "full_640x320x32_wrap_samples.npz"),
                        cfg=np.array(sorted(cfg.items()), dtype=object), **s)
    print("wrote wrap samples")


def full_config2():
    """BASELINE configs[2] shapes: 640x320, 64 spheres + CoordNet, ngf 64, batch 2, the bf16 network as the build
    defines it (oracle/nets.py forward(bf16=True)).  Also the distance of that definition from the FP32 oracle on the
    same inputs (max / mean |bf16 oracle - fp32 oracle| per stage): the number a user of the bf16 path cares about;
    the GPU test reports and gates |bf16 HIP path - fp32 oracle| against it."""
    cfg = dict(seed=8966, b=2, h=320, w=640, d=64, ngf=64, coord=True)
    inp = make_inputs(cfg["seed"], cfg["b"], cfg["h"], cfg["w"])
    d, ngf = cfg["d"], cfg["ngf"]
    weights = onets.init_weights(6 * d, 2 * d, ngf=ngf, coord_net=True, seed=cfg["seed"], randomize_affine=True)
    outs = {}
    for dtype in ("bf16", "f32"):
        o = OracleMSI(weights=weights, coord_net=True, dtype=dtype)
        planes = o.inv_depths(1.0, 100.0, d)
        pred, net_input = o.infer_msi(inp["src_image"], inp["ref_image"], None, None, inp["ref_pose"], inp["src_pose"],
                                      inp["intrinsics"], "blend_psv", d, planes, ngf=ngf)
        rgb = o.msi_render_equirect_view(pred["rgba_layers"], inp["tgt_pose_rt"], inp["tgt_pos"], planes, inp["intrinsics"])
        dep = o.msi_render_equirect_depth(pred["rgba_layers"], inp["tgt_pose_rt"], inp["tgt_pos"], planes, inp["intrinsics"])
        outs[dtype] = dict(psv=net_input, rgba_layers=pred["rgba_layers"], rgb=rgb, depth=dep)
    o = OracleMSI(coord_net=True)
    s = ods_fixture(o, inp, outs["bf16"], planes, d, seed=2, round_fn=onets.bf16_round)
    f32 = dense_samples(outs["f32"], ("rgba_layers", "rgb", "depth"), s["extra_pixels"], 2)
    for k in ("rgba_layers", "rgb", "depth"):
        s["f32val_" + k] = f32["val_" + k]
        diff = np.abs(outs["bf16"][k].astype(np.float64) - outs["f32"][k].astype(np.float64))
        s["bf16_vs_f32_max_" + k] = np.float64(diff.max())
        s["bf16_vs_f32_mean_" + k] = np.float64(diff.mean())
    np.savez_compressed(os.path.join(HERE, "full_config2_bf16_640x320x64_b2_samples.npz"),
                        cfg=np.array(sorted(cfg.items()), dtype=object), **s)
    print("wrote config2 samples", {k: float(s[k]) for k in s if k.startswith("bf16_vs_f32")})


def full_config3():
    """BASELINE configs[3]: 1280x640, 32 spheres, ngf 64, fp32 -- (a) the whole pipeline at the high resolution and
    (b) the reference's high-res mode (test.py:283-394): network at 640x320 (the config1 frame), layers re-assembled
    and rendered at 1280x640 by the per-plane loop of oracle.MSI.render_hres."""
    cfg = dict(seed=8967, b=1, h=640, w=1280, d=32, ngf=64, coord=True, low_seed=8964, low_h=320, low_w=640)
    inp, out = run(cfg["seed"], cfg["b"], cfg["h"], cfg["w"], cfg["d"], cfg["ngf"], True)
    o = OracleMSI(coord_net=True)
    planes = o.inv_depths(1.0, 100.0, cfg["d"])
    samples = ods_fixture(o, inp, out, planes, cfg["d"], seed=3)
    del out
    low_inp, low = run(cfg["low_seed"], 1, cfg["low_h"], cfg["low_w"], cfg["d"], cfg["ngf"], True)
    weights = onets.init_weights(6 * cfg["d"], 2 * cfg["d"], ngf=cfg["ngf"], coord_net=True, seed=cfg["low_seed"], randomize_affine=True)
    o = OracleMSI(weights=weights, coord_net=True)
    hrgb, hdep = o.render_hres(low["blend_weights"], low["alphas"], inp["ref_image"], inp["src_image"], low_inp["ref_pose"],
                               low_inp["src_pose"], low_inp["tgt_pose_rt"], low_inp["tgt_pos"], planes, low_inp["intrinsics"])
    samples.update(dense_samples(dict(hres_rgb=hrgb, hres_depth=hdep), ("hres_rgb", "hres_depth"), samples["extra_pixels"], 4))
    np.savez_compressed(os.path.join(HERE, "full_config3_1280x640x32_samples.npz"),
                        cfg=np.array(sorted(cfg.items()), dtype=object), **samples)
    print("wrote config3 samples")


def full_hres4k():
    """SURVEY 8f-3 / the reference's DEFAULT high-res size (loader.py:34-35: 4096 x 2048; test.py:283-394): network at
    640x320 (the config1 frame), blend weights / alphas upsampled (align_corners) and the layers re-assembled from the
    4096x2048 sweep volume and rendered, plane by plane as the reference's host loop does (oracle.MSI.render_hres;
    ~15 min of numpy).  Dense stratified samples of hres_rgb / hres_depth + the pixels with disc < 0 on the far / near
    plane at the high resolution."""
    import time
    from oracle import geometry as G
    cfg = dict(seed=8969, b=1, h=2048, w=4096, d=32, ngf=64, coord=True, low_seed=8964, low_h=320, low_w=640)
    t0 = time.time()
    low_inp, low = run(cfg["low_seed"], 1, cfg["low_h"], cfg["low_w"], cfg["d"], cfg["ngf"], True)
    print("low-res pass %.0f s" % (time.time() - t0), flush=True)
    hres = make_inputs(cfg["seed"], 1, cfg["h"], cfg["w"])
    o = OracleMSI(coord_net=True)
    planes = o.inv_depths(1.0, 100.0, cfg["d"])
    # invalid pixels of both sweeps on the far / near plane (identity poses: format_network_input's curr_pose = I)
    S, T = G.lat_long_grid((cfg["h"], cfg["w"]))
    bad = np.zeros((cfg["h"], cfg["w"]), dtype=bool)
    for order in (1, -1):
        for dpt in (planes[0], planes[-1]):
            pts = G.apply_pose(G.backproject_spherical(S, T, np.asarray([dpt], np.float32)), np.eye(4, dtype=np.float32))
            _, _, v = G.project_ods(pts, order, low_inp["intrinsics"][0, 0, 0], cfg["w"], cfg["h"])
            bad |= ~v.reshape(cfg["h"], cfg["w"])
    extra = np.flatnonzero(bad.reshape(-1))
    print("invalid far / near pixels at 4096x2048: %d (%.0f s)" % (extra.size, time.time() - t0), flush=True)
    hrgb, hdep = o.render_hres(low["blend_weights"], low["alphas"], hres["ref_image"], hres["src_image"], low_inp["ref_pose"],
                               low_inp["src_pose"], low_inp["tgt_pose_rt"], low_inp["tgt_pos"], planes, low_inp["intrinsics"])
    print("render_hres %.0f s" % (time.time() - t0), flush=True)
    s = dense_samples(dict(hres_rgb=hrgb, hres_depth=hdep), ("hres_rgb", "hres_depth"), extra, 9)
    s["extra_pixels"] = extra.astype(np.int32)
    s["sample_seed"] = np.int64(9)
    np.savez_compressed(os.path.join(HERE, "full_hres_4096x2048x32_samples.npz"),
                        cfg=np.array(sorted(cfg.items()), dtype=object), **s)
    print("wrote hres4k samples", {k: v.shape for k, v in s.items() if hasattr(v, "shape")})


def pp_inputs(seed, b, n):
    """data_loader.py:205-226 (input_type PP): fx = cx = W/2, fy = cy = H/2; source shifted along -x by the input
    offset, target by the target offset (+ a small rotation).  Odd faces additionally get a ROTATED source camera
    (0.05 rad about y, 0.02 about x) so that the slerp of interpolate_pose is not the identity."""
    from tests.util import smooth_noise
    rng = np.random.RandomState(seed)
    ref = smooth_noise(rng, b, n, n); src = smooth_noise(rng, b, n, n)
    K = np.tile(np.array([[n / 2, 0, n / 2], [0, n / 2, n / 2], [0, 0, 1]], np.float32)[None], (b, 1, 1))
    eye = np.tile(np.eye(4, dtype=np.float32)[None], (b, 1, 1))
    src_pose = eye.copy(); src_pose[:, 0, 3] = -0.064
    for k in range(1, b, 2):
        ay, ax = 0.05, 0.02
        ry = np.array([[np.cos(ay), 0, np.sin(ay)], [0, 1, 0], [-np.sin(ay), 0, np.cos(ay)]])
        rx = np.array([[1, 0, 0], [0, np.cos(ax), -np.sin(ax)], [0, np.sin(ax), np.cos(ax)]])
        src_pose[k, :3, :3] = (ry @ rx).astype(np.float32)
    tgt_pose = eye.copy(); tgt_pose[:, 0, 3] = -0.03; tgt_pose[:, 1, 3] = 0.01
    th = 0.02
    tgt_pose[:, 0, 0] = np.cos(th); tgt_pose[:, 0, 2] = np.sin(th); tgt_pose[:, 2, 0] = -np.sin(th); tgt_pose[:, 2, 2] = np.cos(th)
    return ref, src, K, eye, src_pose, tgt_pose


def full_config4():
    """BASELINE configs[4]: input_type=PP, 256x256 cube faces, 32 planes, ngf 64, 2 faces: perspective plane sweep at the
    slerp mid-point pose (train.py:118-121; oracle/poses.py restates utils.py:55-74) -> network -> assembly ->
    mpi_render_view (msi.py:527-548, 644-646).  The interpolated pose itself is stored (`interp_pose`)."""
    from oracle import poses as oposes
    cfg = dict(seed=8968, b=2, n=256, d=32, ngf=64, coord=True)
    ref, src, K, eye, src_pose, tgt_pose = pp_inputs(cfg["seed"], cfg["b"], cfg["n"])
    d, ngf = cfg["d"], cfg["ngf"]
    weights = onets.init_weights(6 * d, 2 * d, ngf=ngf, coord_net=True, seed=cfg["seed"], randomize_affine=True)
    o = OracleMSI(weights=weights, coord_net=True, input_type="PP")
    planes = o.inv_depths(1.0, 100.0, d)
    interp = oposes.interpolate_pose(eye, src_pose)
    interp_inv = np.linalg.inv(interp.astype(np.float64)).astype(np.float32)
    pred, net_input = o.infer_msi(src, ref, None, None, eye, src_pose, K, "blend_psv", d, planes, ngf=ngf, ref_pose_inv=interp_inv)
    rel = np.matmul(tgt_pose, interp_inv).astype(np.float32)
    rgb = o.mpi_render_view(pred["rgba_layers"], rel, planes, K)
    out = dict(psv=net_input, rgba_layers=pred["rgba_layers"], rgb=rgb)
    s = dense_samples(out, ("psv", "rgba_layers", "rgb"), None, 5)
    s["sample_seed"] = np.int64(5)
    s["interp_pose"] = interp
    np.savez_compressed(os.path.join(HERE, "full_config4_pp_256x256x32_b2_samples.npz"),
                        cfg=np.array(sorted(cfg.items()), dtype=object), **s)
    print("wrote config4 samples")


if __name__ == "__main__":
    main()
