#!/usr/bin/env python
"""The per-GPU workloads of BASELINE.json configs[2..4] on one MI355X (bench.py measures configs[1],
the configuration the headline metric is quoted on; these are the parity-test configurations, timed
for DESIGN.md / profiles/).  One JSON line per configuration:

  configs[2]  640x320 ODS, 64 spheres + CoordNet, batch 16, bf16 network (MSI(dtype='bf16'))
  configs[3]  1280x640 ODS, 32 spheres, batch 32 over 8 GPUs = 4 per GPU, fp32 (full pipeline at the high resolution)
  configs[4]  input_type=PP, 6 x 256^2 cube faces, 32 planes, batch 64 over 8 GPUs = 8 faces per GPU, fp32

    python tools/bench_configs.py [--steps K] [--warmup W] [--configs 2 3 4]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from matryodshka_amd import MSI, nets
from tests.util import make_inputs, smooth_noise

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--warmup", type=int, default=3)
ap.add_argument("--configs", type=int, nargs="*", default=[2, 3, 4])
a = ap.parse_args()
dev = torch.device("cuda:0")


def timed(step):
    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.steps):
        step()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / a.steps


def ods(cfg, B, H, W, D, dtype):
    inp = make_inputs(8964, B, H, W)
    model = MSI(weights=nets.init_weights(6 * D, 2 * D, 64, True), coord_net=True, dtype=dtype)
    planes = model.inv_depths(1.0, 100.0, D)
    src_u8 = torch.from_numpy(np.ascontiguousarray(inp["src_image"])).to(dev).contiguous()
    ref_u8 = torch.from_numpy(np.ascontiguousarray(inp["ref_image"])).to(dev).contiguous()
    t = {k: torch.from_numpy(np.ascontiguousarray(inp[k])).to(dev) for k in ("ref_pose", "src_pose", "intrinsics", "tgt_pose_rt", "tgt_pos")}
    rpi = torch.linalg.inv(torch.from_numpy(inp["ref_pose"])).contiguous().to(dev)

    def net_only():
        return model.run_net(net_input[0], 2 * D, 64)

    net_input = [None]

    def step():
        src, ref = model.preprocess_image(src_u8), model.preprocess_image(ref_u8)
        net_input[0] = model.format_network_input(ref, src, t["ref_pose"], t["src_pose"], planes, t["intrinsics"], ref_pose_inv=rpi)
        pred = model.run_net(net_input[0], 2 * D, 64)
        out = model.assemble_layers(net_input[0], pred, D)
        rgb, dep = model.msi_render_equirect_view_and_depth(out["rgba_layers"], t["tgt_pose_rt"], t["tgt_pos"], planes, t["intrinsics"])
        return model.deprocess_image(rgb), model.deprocess_depth_image(dep)

    ms = timed(step)
    ms_net = timed(net_only)
    fl = bench.cnn_flops(H, W, 6 * D, 2 * D, 64, True) * B
    return {"metric": "novel-view frames/sec, %dx%d ODS->%d-sphere MSI infer+render" % (W, H, D), "value": round(B * 1e3 / ms, 2),
            "unit": "frames/s", "n_gpus": 1, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms, 3),
            "higher_is_better": True, "dtype": dtype, "data": "synthetic",
            "config": {"workload": cfg, "height": H, "width": W, "num_spheres": D, "frames_per_step_per_gpu": B},
            "network": {"ms_per_step": round(ms_net, 3), "TFLOPps": round(fl / ms_net / 1e9, 1)}}


def hres(cfg, B, H, W, HH, HW, D):
    """The reference's high-res mode (test.py:283-394): network at HxW, layers re-assembled and rendered at HHxHW."""
    inp, hinp = make_inputs(8964, B, H, W), make_inputs(8965, B, HH, HW)
    model = MSI(weights=nets.init_weights(6 * D, 2 * D, 64, True), coord_net=True)
    planes = model.inv_depths(1.0, 100.0, D)
    g = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev).contiguous()
    src_u8, ref_u8, hsrc_u8, href_u8 = g(inp["src_image"]), g(inp["ref_image"]), g(hinp["src_image"]), g(hinp["ref_image"])
    t = {k: g(inp[k]) for k in ("ref_pose", "src_pose", "intrinsics", "tgt_pose_rt", "tgt_pos")}
    rpi = torch.linalg.inv(torch.from_numpy(inp["ref_pose"])).contiguous().to(dev)

    def step():
        src, ref = model.preprocess_image(src_u8), model.preprocess_image(ref_u8)
        net_input = model.format_network_input(ref, src, t["ref_pose"], t["src_pose"], planes, t["intrinsics"], ref_pose_inv=rpi)
        pred = model.run_net(net_input, 2 * D, 64)
        out = model.assemble_layers(net_input, pred, D, extra_outputs="blend_weights alphas")
        rgb, dep = model.msi_render_equirect_hres(out["blend_weights"], out["alphas"], href_u8, hsrc_u8, t["ref_pose"], t["src_pose"],
                                                  t["tgt_pose_rt"], t["tgt_pos"], planes, t["intrinsics"], ref_pose_inv=rpi)
        return model.deprocess_image(rgb), model.deprocess_depth_image(dep)

    ms = timed(step)
    return {"metric": "novel-view frames/sec, %dx%d network -> %dx%d high-res re-render, %d spheres" % (W, H, HW, HH, D),
            "value": round(B * 1e3 / ms, 2), "unit": "frames/s", "n_gpus": 1, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(ms, 3), "higher_is_better": True, "dtype": "f32", "data": "synthetic",
            "config": {"workload": cfg, "height": H, "width": W, "hres_height": HH, "hres_width": HW, "num_spheres": D,
                       "frames_per_step_per_gpu": B}}


def pp(cfg, B, N, D):
    rng = np.random.RandomState(8964)
    ref, src = smooth_noise(rng, B, N, N), smooth_noise(rng, B, N, N)
    K = np.tile(np.array([[N / 2, 0, N / 2], [0, N / 2, N / 2], [0, 0, 1]], np.float32)[None], (B, 1, 1))
    eye = np.tile(np.eye(4, dtype=np.float32)[None], (B, 1, 1))
    src_pose = eye.copy(); src_pose[:, 0, 3] = -0.064
    tgt_pose = eye.copy(); tgt_pose[:, 0, 3] = -0.03; tgt_pose[:, 1, 3] = 0.01
    model = MSI(weights=nets.init_weights(6 * D, 2 * D, 64, True), coord_net=True, input_type='PP')
    planes = model.inv_depths(1.0, 100.0, D)
    g = {k: torch.from_numpy(v).to(dev).contiguous() for k, v in dict(ref=ref, src=src, K=K, eye=eye, src_pose=src_pose, tgt_pose=tgt_pose).items()}
    Kinv = torch.linalg.inv(torch.from_numpy(K)).contiguous().to(dev)

    def step():
        r, s = model.preprocess_image(g["ref"]), model.preprocess_image(g["src"])
        net_input = model.format_network_input(r, s, g["eye"], g["src_pose"], planes, g["K"], ref_pose_inv=g["eye"])
        pred = model.run_net(net_input, 2 * D, 64)
        out = model.assemble_layers(net_input, pred, D)
        rgb = model.mpi_render_view(out["rgba_layers"], g["tgt_pose"], planes, g["K"], intrinsics_inv=Kinv)
        return model.deprocess_image(rgb)

    ms = timed(step)
    return {"metric": "novel-view cube faces/sec, %dx%d perspective faces -> %d-plane MPI infer+render" % (N, N, D),
            "value": round(B * 1e3 / ms, 2), "unit": "faces/s", "n_gpus": 1, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(ms, 3), "higher_is_better": True, "dtype": "f32", "data": "synthetic",
            "config": {"workload": cfg, "height": N, "width": N, "num_planes": D, "faces_per_step_per_gpu": B}}


for c in a.configs:
    if c == 2:
        print(json.dumps(ods("BASELINE configs[2]: 640x320 ODS, 64 spheres + CoordNet, batch 16, bf16 network", 16, 320, 640, 64, "bf16")), flush=True)
    elif c == 3:
        print(json.dumps(ods("BASELINE configs[3]: 1280x640 ODS, 32 spheres, 4 frames per GPU (batch 32 over 8 GPUs), fp32", 4, 640, 1280, 32, "f32")), flush=True)
        print(json.dumps(hres("BASELINE configs[3], the reference's high-res mode (test.py:283-394): network at 640x320, re-render at 1280x640, "
                              "4 frames per GPU, fp32", 4, 320, 640, 640, 1280, 32)), flush=True)
    elif c == 4:
        print(json.dumps(pp("BASELINE configs[4]: input_type=PP, 256x256 cube faces, 32 planes, 8 faces per GPU (batch 64 over 8 GPUs), fp32", 8, 256, 32)), flush=True)
