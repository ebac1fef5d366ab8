#!/usr/bin/env python
"""Generates the golden fixtures under tests/golden/ from the CPU oracle.

The reference (Python 2 + TensorFlow 1.14) cannot be imported or run in the build container,
and it ships no vectors of its own, so these are REGRESSION vectors of the oracle (pinned by the
analytic KATs in tests/test_oracle_kat.py), not outputs of the reference binary.

    python tests/golden/make_golden.py            # small set (seconds)
    python tests/golden/make_golden.py --full     # + sparse samples of one 640x320x32 frame (minutes)
    python tests/golden/make_golden.py --configs config2 config3 config4   # full-size fixtures of BASELINE configs[2..4]
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
    ap.add_argument("--full", action="store_true", help="+ sparse samples of one BASELINE configs[1] frame (640x320x32)")
    ap.add_argument("--configs", nargs="*", default=[], choices=["config2", "config3", "config4"],
                    help="+ full-size sparse-sample fixtures of BASELINE configs[2] / [3] / [4] (tens of minutes of CPU each)")
    a = ap.parse_args()
    for name, cfg in (("small_coord", dict(seed=11, b=1, h=16, w=32, d=4, ngf=8, coord=True)),
                      ("small_wrap", dict(seed=12, b=2, h=16, w=40, d=4, ngf=8, coord=False))):
        inp, out = run(**cfg)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), cfg=np.array(sorted(cfg.items()), dtype=object),
                            **{"in_" + k: v for k, v in inp.items()}, **{"out_" + k: v for k, v in out.items()})
        print("wrote", name, {k: v.shape for k, v in out.items()})
    if a.full:
        full_config1()
    for name in a.configs:
        {"config2": full_config2, "config3": full_config3, "config4": full_config4}[name]()


def _samples(out, keys, seed=0, n=4096):
    rng = np.random.RandomState(seed)
    samples = {}
    for k in keys:
        flat = np.asarray(out[k]).reshape(-1)
        idx = rng.randint(0, flat.size, size=n)
        samples["idx_" + k] = idx
        samples["val_" + k] = flat[idx]
        samples["mean_" + k] = np.float64(flat.astype(np.float64).mean())
    return samples


def full_config1():
    cfg = dict(seed=8964, b=1, h=320, w=640, d=32, ngf=64, coord=True)
    inp, out = run(**cfg)
    np.savez_compressed(os.path.join(HERE, "full_640x320x32_samples.npz"),
                        cfg=np.array(sorted(cfg.items()), dtype=object), **_samples(out, ("psv", "rgba_layers", "rgb", "depth")))
    print("wrote full-size samples")


def full_config2():
    """BASELINE configs[2] shapes: 640x320, 64 spheres + CoordNet, ngf 64, batch 2, the bf16 network as the build
    defines it (oracle/nets.py forward(bf16=True)); sparse samples of every stage."""
    cfg = dict(seed=8966, b=2, h=320, w=640, d=64, ngf=64, coord=True)
    inp = make_inputs(cfg["seed"], cfg["b"], cfg["h"], cfg["w"])
    d, ngf = cfg["d"], cfg["ngf"]
    weights = onets.init_weights(6 * d, 2 * d, ngf=ngf, coord_net=True, seed=cfg["seed"], randomize_affine=True)
    o = OracleMSI(weights=weights, coord_net=True, dtype="bf16")
    planes = o.inv_depths(1.0, 100.0, d)
    pred, net_input = o.infer_msi(inp["src_image"], inp["ref_image"], None, None, inp["ref_pose"], inp["src_pose"],
                                  inp["intrinsics"], "blend_psv", d, planes, ngf=ngf)
    rgb = o.msi_render_equirect_view(pred["rgba_layers"], inp["tgt_pose_rt"], inp["tgt_pos"], planes, inp["intrinsics"])
    dep = o.msi_render_equirect_depth(pred["rgba_layers"], inp["tgt_pose_rt"], inp["tgt_pos"], planes, inp["intrinsics"])
    out = dict(psv=net_input, rgba_layers=pred["rgba_layers"], rgb=rgb, depth=dep)
    np.savez_compressed(os.path.join(HERE, "full_config2_bf16_640x320x64_b2_samples.npz"),
                        cfg=np.array(sorted(cfg.items()), dtype=object), **_samples(out, ("psv", "rgba_layers", "rgb", "depth"), seed=2))
    print("wrote config2 samples")


def full_config3():
    """BASELINE configs[3]: 1280x640, 32 spheres, ngf 64, fp32 -- (a) the whole pipeline at the high resolution and
    (b) the reference's high-res mode (test.py:283-394): network at 640x320 (the config1 frame), layers re-assembled
    and rendered at 1280x640 by the per-plane loop of oracle.MSI.render_hres."""
    cfg = dict(seed=8967, b=1, h=640, w=1280, d=32, ngf=64, coord=True, low_seed=8964, low_h=320, low_w=640)
    inp, out = run(cfg["seed"], cfg["b"], cfg["h"], cfg["w"], cfg["d"], cfg["ngf"], True)
    samples = _samples(out, ("psv", "rgba_layers", "rgb", "depth"), seed=3)
    low_inp, low = run(cfg["low_seed"], 1, cfg["low_h"], cfg["low_w"], cfg["d"], cfg["ngf"], True)
    weights = onets.init_weights(6 * cfg["d"], 2 * cfg["d"], ngf=cfg["ngf"], coord_net=True, seed=cfg["low_seed"], randomize_affine=True)
    o = OracleMSI(weights=weights, coord_net=True)
    planes = o.inv_depths(1.0, 100.0, cfg["d"])
    hrgb, hdep = o.render_hres(low["blend_weights"], low["alphas"], inp["ref_image"], inp["src_image"], low_inp["ref_pose"],
                               low_inp["src_pose"], low_inp["tgt_pose_rt"], low_inp["tgt_pos"], planes, low_inp["intrinsics"])
    samples.update(_samples(dict(hres_rgb=hrgb, hres_depth=hdep), ("hres_rgb", "hres_depth"), seed=4))
    np.savez_compressed(os.path.join(HERE, "full_config3_1280x640x32_samples.npz"),
                        cfg=np.array(sorted(cfg.items()), dtype=object), **samples)
    print("wrote config3 samples")


def pp_inputs(seed, b, n):
    """data_loader.py:205-226 (input_type PP): fx = cx = W/2, fy = cy = H/2; source shifted along -x by the input
    offset, target by the target offset (+ a small rotation)."""
    from tests.util import smooth_noise
    rng = np.random.RandomState(seed)
    ref = smooth_noise(rng, b, n, n); src = smooth_noise(rng, b, n, n)
    K = np.tile(np.array([[n / 2, 0, n / 2], [0, n / 2, n / 2], [0, 0, 1]], np.float32)[None], (b, 1, 1))
    eye = np.tile(np.eye(4, dtype=np.float32)[None], (b, 1, 1))
    src_pose = eye.copy(); src_pose[:, 0, 3] = -0.064
    tgt_pose = eye.copy(); tgt_pose[:, 0, 3] = -0.03; tgt_pose[:, 1, 3] = 0.01
    th = 0.02
    tgt_pose[:, 0, 0] = np.cos(th); tgt_pose[:, 0, 2] = np.sin(th); tgt_pose[:, 2, 0] = -np.sin(th); tgt_pose[:, 2, 2] = np.cos(th)
    return ref, src, K, eye, src_pose, tgt_pose


def full_config4():
    """BASELINE configs[4]: input_type=PP, 256x256 cube faces, 32 planes, ngf 64, 2 faces: perspective plane sweep at the
    slerp mid-point pose (train.py:118-121) -> network -> assembly -> mpi_render_view (msi.py:527-548, 644-646)."""
    from matryodshka_amd import poses
    cfg = dict(seed=8968, b=2, n=256, d=32, ngf=64, coord=True)
    ref, src, K, eye, src_pose, tgt_pose = pp_inputs(cfg["seed"], cfg["b"], cfg["n"])
    d, ngf = cfg["d"], cfg["ngf"]
    weights = onets.init_weights(6 * d, 2 * d, ngf=ngf, coord_net=True, seed=cfg["seed"], randomize_affine=True)
    o = OracleMSI(weights=weights, coord_net=True, input_type="PP")
    planes = o.inv_depths(1.0, 100.0, d)
    interp_inv = np.linalg.inv(poses.interpolate_pose(eye, src_pose).astype(np.float64)).astype(np.float32)
    pred, net_input = o.infer_msi(src, ref, None, None, eye, src_pose, K, "blend_psv", d, planes, ngf=ngf, ref_pose_inv=interp_inv)
    rel = np.matmul(tgt_pose, interp_inv).astype(np.float32)
    rgb = o.mpi_render_view(pred["rgba_layers"], rel, planes, K)
    out = dict(psv=net_input, rgba_layers=pred["rgba_layers"], rgb=rgb)
    np.savez_compressed(os.path.join(HERE, "full_config4_pp_256x256x32_b2_samples.npz"),
                        cfg=np.array(sorted(cfg.items()), dtype=object), **_samples(out, ("psv", "rgba_layers", "rgb"), seed=5))
    print("wrote config4 samples")


if __name__ == "__main__":
    main()
