#!/usr/bin/env python
"""VERDICT r03 item 6 -- the fp32 ceiling question, numerics half (CPU; the speed half is tools/split3_timing.sh on the GPU box).

Would a 3-way bf16 split with 6 products (h.h, h.m, m.h, h.l, l.h, m.m; fp32 accumulate) of every convolution meet the gate
"every full-size fp32 fixture stage within 2x of the native path's max-abs error"?  The oracle emulates it (oracle/nets.py
conv_split3: six fp32 convolutions of bf16-representable operands) on the configs[1] frame of the committed full-size fixture
(640x320, 32 spheres, ngf 64, CoordNet, seed 8964) and this script reports, per stage, |split3 - fp32 oracle| next to the
error of the native fp32 HIP path against the same oracle (the fixture tests' measured figures).

The same study covers the fp16 form (plan option F32_SPLIT_F16): a 2-way fp16 split (22 significand bits per operand), THREE
products h.h + (h.m' + m'.h) 2^-11 (oracle/nets.py conv_split_f16) -- and, because "error against an fp32 convolution" mixes in that
convolution's own summation error, a second table measures single layers against an fp64 convolution of the same fp32 operands:
plain fp32, six-product bf16 and three-product fp16 side by side.

    python tools/split3_study.py > profiles/r04_split_numerics.txt      (about four minutes of CPU)
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from matryodshka_amd.synthetic import make_inputs   # noqa: E402
from oracle import nets as onets                      # noqa: E402
from oracle.msi import MSI as OracleMSI               # noqa: E402

# max-abs of the native fp32 HIP path against the oracle on this frame (profiles/r04_b_pytest_gpu.log / bench line `parity_max_abs_vs_oracle`)
NATIVE = {"pred (tanh output)": 4.5e-6, "rgba_layers": 9.5e-5, "rgb": 5.1e-5, "depth": 3.9e-6}


def main():
    seed, b, h, w, d, ngf = 8964, 1, 320, 640, 32, 64
    inp = make_inputs(seed, b, h, w)
    weights = onets.init_weights(6 * d, 2 * d, ngf=ngf, coord_net=True, seed=seed, randomize_affine=True)
    o = OracleMSI(weights=weights, coord_net=True)
    planes = o.inv_depths(1.0, 100.0, d)
    src, ref = o.preprocess_image(inp["src_image"]), o.preprocess_image(inp["ref_image"])
    psv = o.format_network_input(ref, src, inp["ref_pose"], inp["src_pose"], planes, inp["intrinsics"])
    out = {}
    for name, kw in (("fp32", {}), ("split3", {"split3_products": True}), ("split_f16", {"split3_products": "f16"})):
        t0 = time.time()
        pred, acts = onets.forward(weights, psv, coord_net=True, return_activations=True, **kw)
        lay = o.assemble(psv, pred, d)
        rgb = o.msi_render_equirect_view(lay["rgba_layers"], inp["tgt_pose_rt"], inp["tgt_pos"], planes, inp["intrinsics"])
        dep = o.msi_render_equirect_depth(lay["rgba_layers"], inp["tgt_pose_rt"], inp["tgt_pos"], planes, inp["intrinsics"])
        out[name] = {"acts": acts, "pred (tanh output)": pred, "rgba_layers": lay["rgba_layers"], "rgb": rgb, "depth": dep}
        print("# %s forward + render: %.0f s" % (name, time.time() - t0), flush=True)
    ok_all = {}
    for tag, title in (("split3", "6-product 3-way bf16 split"), ("split_f16", "3-product 2-way fp16 split")):
        print("# %s vs the fp32 oracle, configs[1] frame (640x320x32, ngf 64, CoordNet, seed %d)" % (title, seed))
        print("# raw convolution outputs, relative to the layer's max |raw|:")
        for k in sorted(x for x in out["fp32"]["acts"] if x.endswith("/raw")):
            a, c = out["fp32"]["acts"][k].astype(np.float64), out[tag]["acts"][k].astype(np.float64)
            print("%-14s max |diff| / max |raw| %.2e   mean |diff| / rms %.2e" % (k[:-4], np.abs(a - c).max() / np.abs(a).max(),
                                                                           np.abs(a - c).mean() / np.sqrt((a * a).mean())))
        print("# stages (gate: max-abs <= 2 x the native fp32 HIP path's max-abs against the same oracle):")
        ok = True
        for k in ("pred (tanh output)", "rgba_layers", "rgb", "depth"):
            a, c = out["fp32"][k].astype(np.float64), out[tag][k].astype(np.float64)
            e = np.abs(a - c)
            gate = 2 * NATIVE[k]
            ok &= e.max() <= gate
            print("%-20s %s max-abs %.2e mean-abs %.2e | native HIP fp32 max-abs %.2e | gate %.2e -> %s"
                  % (k, tag, e.max(), e.mean(), NATIVE[k], gate, "PASS" if e.max() <= gate else "FAIL"))
        print("# numerics gate (%s):" % tag, "PASS" if ok else "FAIL")
        ok_all[tag] = ok

    # single layers against an fp64 convolution of the SAME fp32 operands (the layer's real input of this frame): the error of a
    # plain fp32 convolution (torch, fp32 accumulation) beside the two split forms -- all three relative to max |fp64 result|
    import torch
    import torch.nn.functional as TF
    print("# single layers vs an fp64 convolution of the same fp32 operands (max |diff| / max |out|, mean |diff| / rms):")
    acts = out["fp32"]["acts"]
    for lname, src in (("conv1_2", "conv1_1"), ("conv3_2", "conv3_1"), ("conv4_2", "conv4_1"), ("conv6_2", "conv6_1"), ("conv8_2", "conv8_1")):
        if src not in acts:
            continue
        x = torch.from_numpy(np.ascontiguousarray(np.transpose(acts[src], (0, 3, 1, 2)))).float()
        w = torch.from_numpy(np.ascontiguousarray(np.transpose(weights[lname + "/weights"][:, :, :x.shape[1], :], (3, 2, 0, 1)))).float()
        stride = 2 if lname == "conv1_2" else 1
        rate = 2 if lname.startswith("conv4") else 1
        fn = lambda a, b_: TF.conv2d(a, b_, stride=stride, dilation=rate, padding=rate)   # noqa: E731
        truth = fn(x.double(), w.double())
        scale, rms = float(truth.abs().max()), float(torch.sqrt((truth * truth).mean()))
        row = []
        for tag, y in (("fp32", fn(x, w)), ("bf16 x6", onets.conv_split3(fn, x, w)), ("fp16 x3", onets.conv_split_f16(fn, x, w))):
            d_ = (y.double() - truth).abs()
            row.append("%s %.2e / %.2e" % (tag, float(d_.max()) / scale, float(d_.mean()) / rms))
        print("%-8s K = %4d  | %s" % (lname, 9 * x.shape[1], "  | ".join(row)))


if __name__ == "__main__":
    main()
