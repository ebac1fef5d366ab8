#!/usr/bin/env python
"""Generates the golden fixtures under tests/golden/ from the CPU oracle.

The reference (Python 2 + TensorFlow 1.14) cannot be imported or run in the build container,
and it ships no vectors of its own, so these are REGRESSION vectors of the oracle (pinned by the
analytic KATs in tests/test_oracle_kat.py), not outputs of the reference binary.

    python tests/golden/make_golden.py            # small set (seconds)
    python tests/golden/make_golden.py --full     # + sparse samples of one 640x320x32 frame (minutes)
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
    ap.add_argument("--full", action="store_true")
    a = ap.parse_args()
    for name, cfg in (("small_coord", dict(seed=11, b=1, h=16, w=32, d=4, ngf=8, coord=True)),
                      ("small_wrap", dict(seed=12, b=2, h=16, w=40, d=4, ngf=8, coord=False))):
        inp, out = run(**cfg)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), cfg=np.array(sorted(cfg.items()), dtype=object),
                            **{"in_" + k: v for k, v in inp.items()}, **{"out_" + k: v for k, v in out.items()})
        print("wrote", name, {k: v.shape for k, v in out.items()})
    if a.full:
        cfg = dict(seed=8964, b=1, h=320, w=640, d=32, ngf=64, coord=True)
        inp, out = run(**cfg)
        rng = np.random.RandomState(0)
        samples = {}
        for k in ("psv", "rgba_layers", "rgb", "depth"):
            flat = out[k].reshape(-1)
            idx = rng.randint(0, flat.size, size=4096)
            samples["idx_" + k] = idx
            samples["val_" + k] = flat[idx]
            samples["mean_" + k] = np.float64(flat.astype(np.float64).mean())
        np.savez_compressed(os.path.join(HERE, "full_640x320x32_samples.npz"),
                            cfg=np.array(sorted(cfg.items()), dtype=object), **samples)
        print("wrote full-size samples")


if __name__ == "__main__":
    main()
