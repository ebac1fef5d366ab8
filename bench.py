#!/usr/bin/env python
"""Benchmark of the MSI infer -> render hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): novel-view frames/sec, 640x320 ODS -> 32-sphere MSI infer +
render.  One step = one frame per rank (weak scaling): preprocess the ODS pair ->
2x sphere sweep -> CNN -> RGBA assemble -> equirect RGB + depth render -> deprocess,
inputs already resident in HBM, outputs left in HBM (uint8).  Frames are independent,
so ranks shard frames with no data-path collective; the only collective is the
start-up weight broadcast (RCCL) and the timing barrier / max.

Rank 0 prints ONE JSON line with the metric, `roofline` (the CNN's conv kernels, the
dominant cost, against the fp32 MFMA peak; per-stage detail under `stages`) and
`cpu_baseline` (the CPU oracle timed on this box's host cores; N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

H, W, D, NGF = 320, 640, 32, 64
PEAK_FP32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md "Peak FP32 (matrix)"
PEAK_HBM_GBS = 8000.0           # HBM3E spec peak, same guide


def cnn_layer_flops(h, w, cin, nout, ngf, coord):
    """2*MACs of each of the 18 layers in graph order (conv1_1 ... conv8_2, color_pred)."""
    ex = 1 if coord else 0
    def conv(hh, ww, ci, co, k=9, e=ex):
        return 2 * hh * ww * co * (k * (ci + e))
    def convt(hh, ww, ci, co):                   # 4x4 stride-2 transpose: 2x2 taps per OUTPUT pixel (hh, ww)
        return 2 * hh * ww * co * 4 * ci
    return [
        conv(h, w, cin, ngf),                                   # conv1_1
        conv(h // 2, w // 2, ngf, ngf * 2),                     # conv1_2 (stride 2)
        conv(h // 2, w // 2, ngf * 2, ngf * 2),                 # conv2_1
        conv(h // 4, w // 4, ngf * 2, ngf * 4),                 # conv2_2 (stride 2)
        conv(h // 4, w // 4, ngf * 4, ngf * 4),                 # conv3_1
        conv(h // 4, w // 4, ngf * 4, ngf * 4),                 # conv3_2
        conv(h // 8, w // 8, ngf * 4, ngf * 8),                 # conv3_3 (stride 2)
        conv(h // 8, w // 8, ngf * 8, ngf * 8),                 # conv4_1
        conv(h // 8, w // 8, ngf * 8, ngf * 8),                 # conv4_2
        conv(h // 8, w // 8, ngf * 8, ngf * 8),                 # conv4_3
        convt(h // 4, w // 4, ngf * 16, ngf * 4),               # conv6_1 (skip concat in)
        conv(h // 4, w // 4, ngf * 4, ngf * 4),                 # conv6_2
        conv(h // 4, w // 4, ngf * 4, ngf * 4),                 # conv6_3
        convt(h // 2, w // 2, ngf * 8, ngf * 2),                # conv7_1
        conv(h // 2, w // 2, ngf * 2, ngf * 2),                 # conv7_2
        convt(h, w, ngf * 4, ngf),                              # conv8_1
        conv(h, w, ngf, ngf),                                   # conv8_2
        2 * h * w * ngf * nout,                                 # color_pred (1x1 + bias + tanh)
    ]


def cnn_flops(h, w, cin, nout, ngf, coord):
    """2*MACs of the 18 layers (302.4 GFLOP at 640x320, D=32, CoordNet)."""
    return sum(cnn_layer_flops(h, w, cin, nout, ngf, coord))


def cnn_traffic():
    """HBM bytes per frame of the conv kernels (18 launches) from the committed PMC passes
    (profiles/r01_l_hbm_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs, FETCH
    doubled per the gfx950 note of MI355X_MICROARCH.md); None if the profile is not present."""
    path = os.path.join(ROOT, "profiles", "r01_l_hbm_traffic.json")
    try:
        with open(path) as f:
            k = json.load(f)["kernels"]["conv_igemm_kernel"]
        return int(k["hbm_bytes"])
    except Exception:
        return None


def geometry_bytes(h, w, d):
    """Algorithmic HBM bytes per frame of the HBM-bound stages."""
    img = h * w * 3 * 4
    psv = h * w * 6 * d * 4
    pred = h * w * 2 * d * 4
    rgba = h * w * d * 4 * 4
    return {
        "sweep": 2 * img + psv,                 # read 2 images, write the PSV
        "assemble": psv + pred + rgba,          # read PSV + pred, write the layer stack
        "render": rgba + 2 * img,               # read every texel once, write rgb + depth
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU oracle leg (profiling runs)")
    ap.add_argument("--no-coord-net", action="store_true", help="msi_train_net instead of msi_coord_train_net")
    ap.add_argument("--streams", type=int, default=1,
                    help="HIP streams consecutive frames are issued on (1 = strictly one frame at a time, the "
                         "default and the configuration BASELINE quotes; 2 = software-pipeline independent frames, "
                         "each still batch 1, to fill the tile-quantisation tails of the small layers)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run (one rank per GPU)" % args.gpus)
        raise SystemExit("--gpus %d disagrees with WORLD_SIZE=%d" % (args.gpus, world))
    assert torch.cuda.is_available(), "bench.py needs a HIP device (there is no CPU path)"
    local_rank = local_rank % torch.cuda.device_count()   # (ranks may share a GPU in the gloo functional test)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from matryodshka_amd import MSI, nets
    from matryodshka_amd import dist as mdist

    if world > 1:
        mdist.init_process_group()

    coord = not args.no_coord_net
    cin, nout = 6 * D, 2 * D
    # weights: rank 0 initialises, everyone receives them over RCCL (xGMI)
    weights = nets.init_weights(cin, nout, NGF, coord, seed=8964) if rank == 0 else None
    weights = mdist.broadcast_weights(weights, cin, nout, NGF, coord, dev, src=0) if world > 1 else weights
    models = [MSI(weights=weights, coord_net=coord, device=dev) for _ in range(max(1, args.streams))]
    model = models[0]
    streams = [torch.cuda.current_stream(dev)] + [torch.cuda.Stream(device=dev) for _ in range(args.streams - 1)]
    planes = model.inv_depths(1.0, 100.0, D)

    # synthetic ODS pair, seeded per rank (each rank renders its own frames)
    from tests.util import make_inputs
    inp = make_inputs(8964 + rank, 1, H, W)
    src_u8 = torch.from_numpy(np.ascontiguousarray(inp["src_image"])).to(dev).contiguous()   # resident, in the
    ref_u8 = torch.from_numpy(np.ascontiguousarray(inp["ref_image"])).to(dev).contiguous()   # layout the API takes
    ref_pose = torch.from_numpy(inp["ref_pose"]).to(dev)
    src_pose = torch.from_numpy(inp["src_pose"]).to(dev)
    ref_pose_inv = torch.linalg.inv(torch.from_numpy(inp["ref_pose"])).contiguous().to(dev)   # (LAPACK returns a transposed view)
    intr = torch.from_numpy(inp["intrinsics"]).to(dev)
    tgt_pose_rt = torch.from_numpy(inp["tgt_pose_rt"]).to(dev)
    tgt_pos = torch.from_numpy(inp["tgt_pos"]).to(dev)

    stage_names = ["preprocess", "sweep", "cnn", "assemble", "render", "deprocess"]

    def frame(events=None, model=model):
        def mark():
            if events is not None:
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                events.append(e)
        mark()
        src = model.preprocess_image(src_u8)
        ref = model.preprocess_image(ref_u8)
        mark()
        net_input = model.format_network_input(ref, src, ref_pose, src_pose, planes, intr, ref_pose_inv=ref_pose_inv)
        mark()
        pred = model.run_net(net_input, nout, NGF)
        mark()
        out = model.assemble_layers(net_input, pred, D)
        mark()
        rgb, dep = model.msi_render_equirect_view_and_depth(out["rgba_layers"], tgt_pose_rt, tgt_pos, planes, intr)
        mark()
        rgb8 = model.deprocess_image(rgb)
        dep8 = model.deprocess_depth_image(dep)
        mark()
        return rgb, dep, rgb8, dep8, out

    def step(k, events=None):
        if args.streams == 1:
            return frame(events)
        with torch.cuda.stream(streams[k % args.streams]):      # frame k and k+1 overlap on the device
            return frame(None, models[k % args.streams])

    for k in range(max(args.warmup, args.streams)):
        step(k)
    torch.cuda.synchronize()
    if world > 1:
        mdist.barrier()
    torch.cuda.synchronize()
    all_events = []
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev = []
        result = step(k, ev)
        all_events.append(ev)
    torch.cuda.synchronize()
    if world > 1:
        mdist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    elapsed = mdist.max_over_ranks(elapsed, dev) if world > 1 else elapsed

    if rank != 0:
        return

    if args.streams > 1:
        # per-stage HIP events are only meaningful when frames do not overlap: time one frame alone
        torch.cuda.synchronize()
        all_events = []
        for _ in range(5):
            ev = []
            frame(ev)
            all_events.append(ev)
        torch.cuda.synchronize()
    stage_ms = {}
    for si, name in enumerate(stage_names):
        stage_ms[name] = float(np.mean([ev[si].elapsed_time(ev[si + 1]) for ev in all_events]))
    ms_per_step = elapsed / args.steps * 1e3
    fps = world * args.steps / elapsed

    flops = cnn_flops(H, W, cin, nout, NGF, coord)
    gbytes = geometry_bytes(H, W, D)
    cnn_tflops = flops / (stage_ms["cnn"] * 1e-3) / 1e12
    stages = {k: {"ms": round(v, 4)} for k, v in stage_ms.items()}
    for k in ("sweep", "assemble", "render"):
        gbs = gbytes[k] / (stage_ms[k] * 1e-3) / 1e9
        stages[k].update({"bound": "hbm", "algorithmic_MB": round(gbytes[k] / 1e6, 1),
                          "achieved_GBps": round(gbs, 1), "frac": round(gbs / PEAK_HBM_GBS, 4)})
    stages["cnn"].update({"bound": "mfma", "algorithmic_GFLOP": round(flops / 1e9, 1),
                          "achieved_TFLOPps": round(cnn_tflops, 2),
                          "frac": round(cnn_tflops / PEAK_FP32_MFMA_TFLOPS, 4)})

    line = {
        "metric": "novel-view frames/sec, 640x320 ODS->32-sphere MSI infer+render",
        "value": round(fps, 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BASELINE configs[1]: 640x320 ODS pair, 32 spheres, batch=1 per GPU, fp32, "
                               + ("CoordNet" if coord else "wrap-pad net") + ", infer + RGB&depth render",
                   "height": H, "width": W, "num_spheres": D, "ngf": NGF, "frames_per_step_per_gpu": 1,
                   "parallelism": "frames sharded over %d GPU(s), no data-path collective" % world,
                   "streams_per_gpu": args.streams},
        "roofline": {"kernel": "conv_igemm_kernel (18 launches/frame, fp32 MFMA implicit GEMM; + 17 ln_finish)",
                     "bound": "mfma", "achieved": round(cnn_tflops, 3), "peak": PEAK_FP32_MFMA_TFLOPS,
                     "unit": "TFLOP/s", "frac": round(cnn_tflops / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": cnn_traffic(),
                     "traffic_note": "HBM bytes per frame of the 18 conv launches, profiles/r01_l_hbm_traffic.json "
                                     "(separate --pmc FETCH_SIZE / WRITE_SIZE passes, gfx950 FETCH correction)",
                     "launches_per_frame": 18, "algorithmic_flops_per_frame": flops,
                     "ms_per_frame": round(stage_ms["cnn"], 4),
                     "timed": "HIP events around msi_net_forward_f32 on the launch stream inside the timed region "
                              "(18 conv + 17 fix-up + 17 ln_apply launches: conservative for the conv kernel alone)"},
        "stages": stages,
    }

    if world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"], parity = cpu_baseline(model, weights, inp, planes, coord, result)
        line["parity_max_abs_vs_oracle"] = parity
    else:
        line["cpu_baseline"] = None
    print(json.dumps(line), flush=True)


def cpu_baseline(model, weights, inp, planes, coord, gpu_result):
    """The CPU oracle ("port": the reference itself needs Python 2 + TF 1.14 and cannot run here)
    on the same workload, timed on this box's host cores: a bounded sample of 2 frames (~10-30 s)
    after a small warm-up that creates the thread pool / conv primitives.  torch-CPU conv runs on
    min(cores, 64) threads (256 threads are 10x slower on this host: oversubscription), the numpy
    geometry is single-threaded."""
    from oracle.msi import MSI as OracleMSI
    from oracle import nets as onets
    cores = os.cpu_count() or 1
    threads = min(cores, 64)
    torch.set_num_threads(threads)
    o = OracleMSI(weights=weights, coord_net=coord)
    wsmall = onets.init_weights(24, 8, 8, coord)
    onets.forward(wsmall, np.zeros((1, 16, 32, 24), np.float32), coord_net=coord)      # warm-up only

    def one_frame():
        pred_o, _ = o.infer_msi(inp["src_image"], inp["ref_image"], None, None, inp["ref_pose"], inp["src_pose"],
                                inp["intrinsics"], "blend_psv", D, planes, ngf=NGF)
        rgb_o = o.msi_render_equirect_view(pred_o["rgba_layers"], inp["tgt_pose_rt"], inp["tgt_pos"], planes, inp["intrinsics"])
        dep_o = o.msi_render_equirect_depth(pred_o["rgba_layers"], inp["tgt_pose_rt"], inp["tgt_pos"], planes, inp["intrinsics"])
        o.deprocess_image(rgb_o)
        o.deprocess_depth_image(dep_o)
        return pred_o, rgb_o, dep_o

    nframes = 2
    t0 = time.perf_counter()
    for _ in range(nframes):
        pred_o, rgb_o, dep_o = one_frame()
    t = (time.perf_counter() - t0) / nframes
    rgb, dep, _, _, out = gpu_result
    parity = {
        "rgba_layers": float(np.abs(out["rgba_layers"].cpu().numpy() - pred_o["rgba_layers"]).max()),
        "rgb": float(np.abs(rgb.cpu().numpy() - rgb_o).max()),
        "depth": float(np.abs(dep.cpu().numpy() - dep_o).max()),
    }
    base = {"value": round(1.0 / t, 5), "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": "%d frames of the same workload (640x320, 32 spheres, infer + rgb & depth render), %.1f s per "
                      "frame; torch-CPU conv on %d of %d host cores, numpy geometry single-threaded"
                      % (nframes, t, threads, cores)}
    return base, parity


if __name__ == "__main__":
    main()
