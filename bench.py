#!/usr/bin/env python
"""Benchmark of the MSI infer -> render hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 1|2|3|4]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): novel-view frames/sec, 640x320 ODS -> 32-sphere MSI infer + render
(--config 1, the default and the configuration the metric is quoted on).  One step = one frame per rank
(weak scaling): preprocess the ODS pair -> 2x sphere sweep -> CNN -> RGBA assemble -> equirect RGB + depth
render -> deprocess, inputs already resident in HBM, outputs left in HBM (uint8).  Frames are independent,
so ranks shard frames with no data-path collective; the only collective is the start-up weight broadcast
(RCCL) and the timing barrier / max.

The other BASELINE configurations are parity-test cases; `--config` times them with the same protocol:
  2  640x320 ODS, 64 spheres + CoordNet, batch 16 per GPU, bf16 network                   (weak scaling)
  3  1280x640 ODS, 32 spheres, batch 32 per step SHARDED over the ranks, fp32              (strong scaling)
  4  input_type=PP, 256x256 cube faces, 32 planes, batch 64 faces per step sharded, fp32   (strong scaling)
A rank's shard (dist.shard_frames) is run as one batch; `value` = frames of ALL ranks / max-over-ranks time.

Rank 0 prints ONE JSON line with the metric, `roofline` (the CNN's conv kernels, the dominant cost, against
the MFMA peak of the compute type; per-stage detail under `stages`) and `cpu_baseline` (the CPU oracle timed on
this box's host cores; config 1 at N=1 only).
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

NGF = 64
PEAK_FP32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md "Peak FP32 (matrix)"
PEAK_BF16_MFMA_TFLOPS = 2500.0  # same guide, dense bf16 MFMA
PEAK_HBM_GBS = 8000.0           # HBM3E spec peak, same guide

CONFIGS = {
    1: dict(kind="ods", h=320, w=640, d=32, dtype="f32", per_rank=1, total=None, scaling="weak",
            name="BASELINE configs[1]: 640x320 ODS pair, 32 spheres, batch=1 per GPU, fp32"),
    2: dict(kind="ods", h=320, w=640, d=64, dtype="bf16", per_rank=16, total=None, scaling="weak",
            name="BASELINE configs[2]: 640x320 ODS, 64 spheres + CoordNet, batch=16 per GPU, bf16 network"),
    3: dict(kind="ods", h=640, w=1280, d=32, dtype="f32", per_rank=None, total=32, scaling="strong",
            name="BASELINE configs[3]: 1280x640 high_res ODS, 32 spheres, batch=32 sharded over the GPUs, fp32"),
    4: dict(kind="pp", h=256, w=256, d=32, dtype="f32", per_rank=None, total=64, scaling="strong",
            name="BASELINE configs[4]: input_type=PP 256x256 cube faces, 32 planes, batch=64 faces sharded over the GPUs, fp32"),
}


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


def is_f16_split(kernel_name):
    """A plan kernel name (NetPlan.layer_kernel) of the fp16 three-product form: the x3 kernels' last template argument is the
    number of operand planes (2 = fp16 h | m', 3 = bf16 h | m | l)."""
    return "_x3_kernel" in kernel_name and kernel_name.split("(")[0].rstrip("> ").replace("<", ",").split(",")[-1].strip() == "2"


def split_peak(kernel_name):
    """Dense MFMA peak (fp32-equivalent TFLOP/s) of the instruction a plan's layer runs on."""
    if is_f16_split(kernel_name):
        return PEAK_BF16_MFMA_TFLOPS / 3.0
    if "_x3_kernel" in kernel_name:
        return PEAK_BF16_MFMA_TFLOPS / 6.0
    return PEAK_FP32_MFMA_TFLOPS


def conv_flops(h, w, cin, nout, ngf, coord):
    """2*MACs of the 17 convolutions conv1_1 ... conv8_2 WITHOUT color_pred: the 1x1 head runs inside the fused tail,
    after the event that closes the roofline interval (300.7 GFLOP at 640x320, D=32, CoordNet)."""
    return sum(cnn_layer_flops(h, w, cin, nout, ngf, coord)[:-1])


def conv_algorithmic_bytes(h, w, cin, ngf, coord, nb, act_bytes=4, w_bytes=4):
    """HBM bytes one forward of the 17 convolutions has to move at least: every layer reads its input activation(s)
    once and writes its raw output once (+ the in-place LayerNorm pass where one is launched is NOT counted: it belongs
    to ln_apply), and the packed weights are read once per forward.  What `roofline.traffic_ratio` divides by."""
    c = ngf
    dims = [  # (in pixels scale, cin, out pixels scale, cout, taps)
        (1, cin, 1, c, 9), (1, c, 4, 2 * c, 9), (4, 2 * c, 4, 2 * c, 9), (4, 2 * c, 16, 4 * c, 9),
        (16, 4 * c, 16, 4 * c, 9), (16, 4 * c, 16, 4 * c, 9), (16, 4 * c, 64, 8 * c, 9),
        (64, 8 * c, 64, 8 * c, 9), (64, 8 * c, 64, 8 * c, 9), (64, 8 * c, 64, 8 * c, 9),
        (64, 16 * c, 16, 4 * c, 16), (16, 4 * c, 16, 4 * c, 9), (16, 4 * c, 16, 4 * c, 9),
        (16, 8 * c, 4, 2 * c, 16), (4, 2 * c, 4, 2 * c, 9), (4, 4 * c, 1, c, 16), (1, c, 1, c, 9)]
    act = sum(h * w // si * ci + h * w // so * co for si, ci, so, co, _ in dims) * act_bytes * nb
    wts = sum(t * ci * co for _, ci, _, co, t in dims) * w_bytes
    return act + wts



def csrc_hash():
    """sha256 (first 16 hex digits) over the kernel sources the library is built from -- what a traffic profile is
    stamped with (tools/hbm_traffic.py) and what this run compares it to (the GPU box has no .git)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "matryodshka_amd", "csrc", "*.hip")) +
                   glob.glob(os.path.join(ROOT, "matryodshka_amd", "csrc", "*.cpp")) +
                   glob.glob(os.path.join(ROOT, "matryodshka_amd", "csrc", "*.h")) +
                   [os.path.join(ROOT, "include", "msi_hip.h")])
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def cnn_traffic(config=1, coord=True):
    """HBM bytes per STEP of the conv launches (every kernel whose name starts with conv / convt) from the newest committed
    PMC passes of this configuration (profiles/r*_hbm_traffic*.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate
    runs, FETCH doubled per the gfx950 note of MI355X_MICROARCH.md, made by tools/hbm_traffic.py, which stamps the hash
    of the kernel sources it measured).  Returns (bytes, source text, stale): stale = the profile was measured with
    different kernel sources than the ones this run uses (or carries no hash: rounds 1-2); (None, None, None) if no
    profile of the configuration is present."""
    import glob
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic*.json")))
    for path in reversed(paths):
        try:
            with open(path) as f:
                j = json.load(f)
            if int(j.get("config", 1)) != config or bool(j.get("coord_net", True)) != bool(coord):
                continue
            conv = [v["hbm_bytes"] for k, v in j["kernels"].items() if k.startswith("conv")]
            stale = j.get("csrc_sha") != csrc_hash()
            return int(sum(conv)), "%s (commit %s, csrc_sha %s)" % (
                os.path.relpath(path, ROOT), j.get("git_commit", "not recorded"), j.get("csrc_sha", "not recorded")), stale
        except Exception:
            continue
    return None, None, None


def geometry_bytes(h, w, d, psv_bytes=4):
    """Algorithmic HBM bytes per frame of the HBM-bound stages."""
    img = h * w * 3 * 4
    psv = h * w * 6 * d * psv_bytes
    pred = h * w * 2 * d * 4
    rgba = h * w * d * 4 * 4
    return {
        "sweep": 2 * img + psv,                 # read 2 images, write the PSV
        "assemble": psv + pred + rgba,          # read PSV + pred, write the layer stack
        "render": rgba + 2 * img,               # read every texel once, write rgb + depth
    }


_ORIG_AFFINITY = None


def sustained_matrix_rate(dev, seconds=0.25):
    """What THIS part's matrix pipes sustain on CHANGING bf16 operands with nothing else running (msi_probe_matrix_rate; tools/ubench/clock_probe.hip,
    profiles/r06_clock.txt): dense bf16 PFLOP/s, and the shader clock (s_memtime cycles per s_memrealtime second) it ran at.  The 2.5 PFLOP/s dense peak is a
    constant-operand figure at 2.4 GHz; under changing operands the clock itself falls (power), so `roofline.frac_of_sustained` prices the convolutions
    against what the silicon delivers before a single byte is fed to it.  ~0.6 s, after the timed regions."""
    from matryodshka_amd import _native as N
    nwg = 2 * torch.cuda.get_device_properties(dev).multi_processor_count
    ticks = torch.zeros((nwg, 2), dtype=torch.int64, device=dev)
    sink = torch.zeros(4, dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    out = {}
    for changing in (1, 0):
        iters = 20000
        for attempt in range(3):                   # calibrate the iteration count to ~`seconds`, then measure
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            N.check(N.lib.msi_probe_matrix_rate(changing, iters, nwg, ticks.data_ptr(), sink.data_ptr(), stream), "msi_probe_matrix_rate")
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1)
            if attempt < 2:
                iters = max(1000, int(iters * seconds * 1e3 / max(ms, 1e-3)))
        t = ticks.cpu().numpy().astype(np.float64)
        ghz = float(np.median(t[:, 0] / (t[:, 1] / 100e6))) / 1e9
        out["changing" if changing else "constant"] = {"pflops": round(nwg * 4 * iters * 12 * 2 * 32 * 32 * 16 / (ms * 1e-3) / 1e15, 4), "shader_clock_ghz": round(ghz, 3),
                                                        "ms": round(ms, 1)}
    return out


def pin_to_gpu_numa_node(device_index):
    """Pin this rank's launch thread (and the threads it starts) to the CPUs local to its GPU: the PCI device's
    `local_cpulist` under /sys (eight GPUs hang off two sockets; a rank launching from the far socket pays a cross-socket
    hop per kernel launch, and batch-1 frames are 30 launches of ~3 ms).  Returns a dict for the JSON line; never fatal."""
    global _ORIG_AFFINITY
    info = {"pinned": False}
    try:
        _ORIG_AFFINITY = os.sched_getaffinity(0)
        prop = torch.cuda.get_device_properties(device_index)
        bus = "%04x:%02x:%02x.0" % (getattr(prop, "pci_domain_id", 0), prop.pci_bus_id, prop.pci_device_id)
        base = "/sys/bus/pci/devices/" + bus
        with open(base + "/numa_node") as f:
            info["numa_node"] = int(f.read().strip())
        with open(base + "/local_cpulist") as f:
            text = f.read().strip()
        cpus = set()
        for part in text.split(","):
            if "-" in part:
                lo, hi = part.split("-")
                cpus.update(range(int(lo), int(hi) + 1))
            elif part:
                cpus.add(int(part))
        allowed = os.sched_getaffinity(0)
        cpus &= allowed
        info["pci_bus"] = bus
        if cpus and info["numa_node"] >= 0 and len(cpus) < len(allowed):
            os.sched_setaffinity(0, cpus)
            info["pinned"] = True
        info["cpus"] = len(cpus) if cpus else len(allowed)
    except Exception as e:      # no sysfs entry (containers), old torch, ...: run unpinned
        info["error"] = "%s: %s" % (type(e).__name__, e)
    return info


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: re-run this very command line as N ranks under
    torch.distributed.run (one process per GPU, rendezvous on 127.0.0.1, a free port), hand its streams through
    (rank 0 prints the one JSON line) and return its exit code.  With fewer than N visible devices RCCL cannot be
    used (it refuses two ranks on one device): refuse loudly unless MSI_DIST_BACKEND (gloo: functional runs) is set."""
    import socket
    import subprocess
    if "--dist-check" not in sys.argv and not os.environ.get("MSI_DIST_BACKEND") and torch.cuda.device_count() < n:
        print("bench.py --gpus %d: only %d HIP device(s) visible; one rank per GPU over RCCL needs %d "
              "(MSI_DIST_BACKEND=gloo lets ranks share a device for a functional run)" % (n, torch.cuda.device_count(), n),
              file=sys.stderr)
        return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC only on this pool (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.call(cmd, env=env)


def dist_check(args, rank, local_rank, world):
    """--dist-check: what `--gpus N` adds to the one-GPU bench and nothing else (see the flag's help)."""
    from matryodshka_amd import nets
    from matryodshka_amd import dist as mdist
    cfg = CONFIGS[args.config]
    use_gpu = torch.cuda.is_available()
    dev = torch.device("cuda", local_rank % torch.cuda.device_count()) if use_gpu else torch.device("cpu")
    use_pg = world > 1 or args.force_process_group
    if use_pg:
        if use_gpu:
            torch.cuda.set_device(dev)
        mdist.init_process_group()
    cin, nout, ngf = 24, 8, 8
    weights = nets.init_weights(cin, nout, ngf, True, seed=8964) if rank == 0 else None
    if use_pg:
        weights = mdist.broadcast_weights(weights, cin, nout, ngf, True, dev, src=0)
    blob_sum = float(nets.flatten_params(weights, cin, nout, ngf, True).astype(np.float64).sum())
    lo, hi, total = mdist.step_frames(cfg["total"], cfg["per_rank"], rank, world)
    if use_pg:
        mdist.barrier()
    t = mdist.max_over_ranks(1.0 + rank, dev) if use_pg else 1.0
    sums = mdist.gather_floats(blob_sum, dev) if use_pg else [blob_sum]
    ranges = mdist.gather_ranges(lo, hi, dev) if use_pg else [(lo, hi)]
    if rank == 0:
        print(json.dumps({"dist_check": True, "n_gpus": world, "config": args.config,
                          "world_size_process_group": torch.distributed.get_world_size() if use_pg else 1,
                          "backend": torch.distributed.get_backend() if use_pg else None,
                          "frames_per_step": total, "frame_ranges_per_rank": ranges, "max_over_ranks": t,
                          "weights_equal_on_all_ranks": len(set(sums)) == 1}), flush=True)
    if use_pg:
        torch.distributed.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=1, choices=sorted(CONFIGS),
                    help="BASELINE.json configs[N]; 1 (default) is the configuration the metric is quoted on")
    ap.add_argument("--repeats", type=int, default=4,
                    help="extra timed regions of --steps steps after the contract one (reported under `repeats`; the "
                         "headline `value` is always the first region right after the warm-up)")
    ap.add_argument("--substreams", type=int, default=1,
                    help="ODS configurations 2 / 3: a step's batch of this rank runs as S sub-batches on S HIP streams (the "
                         "VALU-bound sweep / tail / render of one sub-batch overlap the MFMA-bound network of another); "
                         "1 (default) = the whole batch as one batch on one stream.  Reported separately (config.substreams)")
    ap.add_argument("--no-settle", dest="settle", action="store_false",
                    help="skip the untimed settle regions before the contract region (see `settle_regions_ms`)")
    ap.add_argument("--prewarm", type=float, default=None,
                    help="seconds of untimed steps before the --warmup steps (default 0.5 at --config 1; 4.0 at the batched "
                         "configurations, whose first second under load contains a slow transient: 19 vs 11.4 ms per step at config 2)")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU oracle leg (profiling runs)")
    ap.add_argument("--no-coord-net", action="store_true", help="msi_train_net instead of msi_coord_train_net")
    ap.add_argument("--arithmetic", default=None, choices=["native", "split3", "split_f16"],
                    help="fp32 configurations: how the stride-1 3x3 layers multiply -- native = v_mfma_f32_32x32x2_f32; split3 = 3-way bf16 "
                         "split of both operands, SIX products on the bf16 MFMA, fp32 accumulation (fp32-grade: dropped terms < 2^-26 of a "
                         "product; plan option F32_SPLIT3); split_f16 = 2-way fp16 split (22 significand bits), THREE products on the fp16 MFMA "
                         "(plan option F32_SPLIT_F16; operands limited to the fp16 range, flagged in the status word).  Default: the library's "
                         "default plan.  Reported as config.arithmetic")
    ap.add_argument("--no-alt-arithmetic", action="store_true",
                    help="skip the second timed region (N = 1, fp32 configurations, default plan only): the same workload with the opt-in "
                         "three-product fp16 split (plan option F32_SPLIT_F16), reported as `alt_arithmetic` -- never as `value`")
    ap.add_argument("--net-opt", action="append", default=[], metavar="K=V",
                    help="tuning: msi_net_plan_set_option(K, V) on every plan of the run (integers; include/msi_hip.h MSI_NET_OPT_*); reported as config.net_options")
    ap.add_argument("--strong-frames", type=int, default=8,
                    help="config 1: after the contract region, also time a FIXED batch of this many frames sharded over the "
                         "ranks (strong-scaling reading of the same path, reported under `strong_scaling`; 0 = skip)")
    ap.add_argument("--streams", type=int, default=1,
                    help="HIP streams consecutive frames are issued on (1 = strictly one frame at a time, the "
                         "default and the configuration BASELINE quotes; 2 = software-pipeline independent frames, "
                         "each still batch 1, to fill the tile-quantisation tails of the small layers; config 1 only)")
    ap.add_argument("--force-process-group", action="store_true",
                    help="initialise torch.distributed even at --gpus 1 (world_size 1; backend nccl = RCCL on a GPU box) and run the weight broadcast, "
                         "barriers, MAX-over-ranks and gathers through it exactly as an N-rank run does: first contact with RCCL on a one-GPU box "
                         "(`distributed.backend` and a measured `weight_broadcast_ms` on the line; VERDICT r05 item 4)")
    ap.add_argument("--no-sustained-probe", action="store_true",
                    help="skip msi_probe_matrix_rate (roofline.frac_of_sustained / sustained_matrix_rate become null): profiling runs, where the probe's 0.5 s of "
                         "matrix-only kernels would sit in the kernel table")
    ap.add_argument("--dist-check", action="store_true",
                    help="only the N-rank plumbing of this file: launch, rendezvous, weight broadcast, frame ranges, barrier and "
                         "max-over-ranks -- no frame loop, no kernels; runs without a GPU on gloo (tests/test_dist_cpu.py) and "
                         "prints its own one-line JSON (never a benchmark line: no `value`)")
    args = ap.parse_args()
    cfg = CONFIGS[args.config]
    if args.prewarm is None:
        args.prewarm = 0.5 if args.config == 1 else 4.0

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus))                # `python bench.py --gpus N` typed directly: become N ranks
    if args.gpus != world:
        raise SystemExit("--gpus %d disagrees with WORLD_SIZE=%d" % (args.gpus, world))
    use_pg = world > 1 or args.force_process_group
    if use_pg and "WORLD_SIZE" not in os.environ:               # --force-process-group typed without a launcher: a one-rank rendezvous on a free local port
        import socket
        s_ = socket.socket()
        s_.bind(("127.0.0.1", 0))
        os.environ.update({"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(s_.getsockname()[1])})
        s_.close()
    if args.dist_check:
        return dist_check(args, rank, local_rank, world)
    assert torch.cuda.is_available(), "bench.py needs a HIP device (there is no CPU path)"
    local_rank = local_rank % torch.cuda.device_count()   # (ranks may share a GPU in the gloo functional test)
    torch.cuda.set_device(local_rank)                      # one process per GPU: rank r drives device LOCAL_RANK
    dev = torch.device("cuda", local_rank)
    affinity = pin_to_gpu_numa_node(local_rank)
    if args.streams > 1 and args.config != 1:
        raise SystemExit("--streams applies to --config 1")
    if args.substreams > 1 and (args.config not in (2, 3) or args.streams > 1):
        raise SystemExit("--substreams applies to --config 2 / 3 (ODS batches) without --streams")

    from matryodshka_amd import MSI, nets
    from matryodshka_amd import dist as mdist

    if use_pg:
        mdist.init_process_group()
        # the data path of an N-GPU run is RCCL over xGMI; anything else must be asked for explicitly (functional tests)
        if not os.environ.get("MSI_DIST_BACKEND"):
            assert torch.distributed.get_backend() == "nccl", "bench.py --gpus N runs on RCCL (backend nccl); got %s" % torch.distributed.get_backend()

    H, W, D = cfg["h"], cfg["w"], cfg["d"]
    coord = not args.no_coord_net
    cin, nout = 6 * D, 2 * D
    # frames of this rank per step: its shard of the step's batch (strong scaling) or a fixed batch (weak scaling)
    lo, hi, frames_total = mdist.step_frames(cfg["total"], cfg["per_rank"], rank, world)
    B = hi - lo

    # weights: rank 0 initialises, everyone receives them over RCCL (xGMI); timed, outside the frame loop
    weights = nets.init_weights(cin, nout, NGF, coord, seed=8964) if rank == 0 else None
    broadcast_ms = None
    if use_pg:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        weights = mdist.broadcast_weights(weights, cin, nout, NGF, coord, dev, src=0)
        torch.cuda.synchronize()
        broadcast_ms = (time.perf_counter() - t0) * 1e3
    models = [MSI(weights=weights, coord_net=coord, device=dev, dtype=cfg["dtype"],
                  input_type="PP" if cfg["kind"] == "pp" else "ODS") for _ in range(max(1, args.streams, args.substreams))]
    model = models[0]
    from matryodshka_amd import _native as _N
    if args.arithmetic is not None:
        if cfg["dtype"] != "f32":
            raise SystemExit("--arithmetic applies to the fp32 configurations")
        for mm in models:
            mm.net_options[_N.NET_OPT_F32_SPLIT3] = 0 if args.arithmetic == "native" else 0x3ffff
            mm.net_options[_N.NET_OPT_F32_SPLIT_F16] = 0x3ffff if args.arithmetic == "split_f16" else 0
    for kv in args.net_opt:
        k_, v_ = kv.split("=")
        for mm in models:
            mm.net_options[int(k_)] = int(v_, 0)
    streams = [torch.cuda.current_stream(dev)] + [torch.cuda.Stream(device=dev) for _ in range(max(args.streams, args.substreams) - 1)]
    planes = model.inv_depths(1.0, 100.0, D)

    # synthetic inputs, seeded per global frame index (every rank renders its own frames), resident in HBM
    from matryodshka_amd.synthetic import make_inputs
    g = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev).contiguous()
    inp = None
    if B > 0 and cfg["kind"] == "ods":
        parts = [make_inputs(8964 + f, 1, H, W) for f in range(lo, hi)]
        inp = {k: np.concatenate([p[k] for p in parts], axis=0) for k in parts[0]}
        src_u8, ref_u8 = g(inp["src_image"]), g(inp["ref_image"])   # in the layout the API takes
        ref_pose, src_pose, intr = g(inp["ref_pose"]), g(inp["src_pose"]), g(inp["intrinsics"])
        ref_pose_inv = g(np.linalg.inv(inp["ref_pose"].astype(np.float64)).astype(np.float32))
        tgt_pose_rt, tgt_pos = g(inp["tgt_pose_rt"]), g(inp["tgt_pos"])
    elif B > 0:
        from matryodshka_amd import poses
        from matryodshka_amd.synthetic import pp_inputs
        parts = [pp_inputs(8964 + f, 1, H) for f in range(lo, hi)]
        ref, src, K, eye, spose, tpose = (np.concatenate([p[i] for p in parts], axis=0) for i in range(6))
        interp_inv = np.linalg.inv(poses.interpolate_pose(eye, spose).astype(np.float64)).astype(np.float32)   # train.py:118-121
        rel = np.matmul(tpose, interp_inv).astype(np.float32)                                                  # msi.py:644-646
        pp = dict(ref=g(ref), src=g(src), K=g(K), eye=g(eye), src_pose=g(spose), interp_inv=g(interp_inv), rel=g(rel),
                  Kinv=g(np.linalg.inv(K.astype(np.float64)).astype(np.float32)))

    stage_names = ["preprocess", "sweep", "cnn", "assemble", "render", "deprocess"]   # cnn = the convolutions (+ ln_apply);
    # assemble = head + layer assembly (one fused kernel on the fp32 blend_psv path, head launch + K3 otherwise)

    def frame(events=None, model=model, cnn_events=None, sl=None):
        """One step of this rank: its B frames as one batch (sl: a sub-batch of them, --substreams).  `events`: a HIP event at
        every stage boundary; `cnn_events`: only around the network (the roofline kernel) -- what the timed region records."""
        if B == 0:
            return None
        if sl is not None:   # (ODS only) contiguous views of the resident inputs
            src_u8_, ref_u8_, ref_pose_, src_pose_, intr_ = src_u8[sl], ref_u8[sl], ref_pose[sl], src_pose[sl], intr[sl]
            ref_pose_inv_, tgt_pose_rt_, tgt_pos_ = ref_pose_inv[sl], tgt_pose_rt[sl], tgt_pos[sl]
        elif cfg["kind"] == "ods":
            src_u8_, ref_u8_, ref_pose_, src_pose_, intr_ = src_u8, ref_u8, ref_pose, src_pose, intr
            ref_pose_inv_, tgt_pose_rt_, tgt_pos_ = ref_pose_inv, tgt_pose_rt, tgt_pos
        def mark():
            if events is not None:
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                events.append(e)
        def mark_cnn():
            if cnn_events is not None:
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                cnn_events.append(e)
        mark()
        if cfg["kind"] == "ods":
            src, ref = model.preprocess_image_pair(src_u8_, ref_u8_)
            mark()
            net_input = model.format_network_input(ref, src, ref_pose_, src_pose_, planes, intr_, ref_pose_inv=ref_pose_inv_)
        else:
            src = model.preprocess_image(pp["src"])
            ref = model.preprocess_image(pp["ref"])
            mark()
            net_input = model.format_network_input(ref, src, pp["eye"], pp["src_pose"], planes, pp["K"], ref_pose_inv=pp["interp_inv"])
        mark(); mark_cnn()
        # network + layer assembly: on the fp32 blend_psv path the head and the assembly are ONE fused HBM-bound kernel; the
        # event between the 17 convolutions and that tail keeps the MFMA roofline of the conv kernel clean
        mid = torch.cuda.Event(enable_timing=True)
        mid.record()                                            # (creates the handle; re-recorded by the library)
        out = model.infer_layers(net_input, D, NGF, event_after_convs=mid)
        if events is not None:
            events.append(mid)
        if cnn_events is not None:
            cnn_events.append(mid)
        mark()
        if cfg["kind"] == "ods":
            rgb, dep = model.msi_render_equirect_view_and_depth(out["rgba_layers"], tgt_pose_rt_, tgt_pos_, planes, intr_)
            mark()
            rgb8, dep8 = model.deprocess_image_and_depth(rgb, dep)
        else:
            rgb = model.mpi_render_view(out["rgba_layers"], pp["rel"], planes, pp["K"], intrinsics_inv=pp["Kinv"])
            dep = None
            mark()
            rgb8, dep8 = model.deprocess_image(rgb), None
        mark()
        return rgb, dep, rgb8, dep8, out

    sub_slices = [slice(B * i // args.substreams, B * (i + 1) // args.substreams) for i in range(args.substreams)]

    def step(k, cnn_events=None):
        if args.substreams > 1:                                  # the batch as S sub-batches, one per stream (whole step's work)
            r = None
            for i, sl in enumerate(sub_slices):
                if sl.stop > sl.start:
                    with torch.cuda.stream(streams[i]):
                        r = frame(None, models[i], cnn_events=cnn_events, sl=sl)
            return r
        if args.streams == 1:
            return frame(None, cnn_events=cnn_events)
        with torch.cuda.stream(streams[k % args.streams]):      # frame k and k+1 overlap on the device
            return frame(None, models[k % args.streams])

    def timed_region(nsteps):
        """EXACTLY nsteps steps between barrier + synchronize on both sides; max over ranks."""
        torch.cuda.synchronize()
        if use_pg:
            mdist.barrier()
        torch.cuda.synchronize()
        ev = []
        t0 = time.perf_counter()
        for k in range(nsteps):
            result = step(k, ev if args.streams == 1 else None)
        torch.cuda.synchronize()
        t_own = time.perf_counter()                              # this rank's own finish time (before the closing barrier)
        if use_pg:
            mdist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        own.append(t_own - t0)
        elapsed = mdist.max_over_ranks(elapsed, dev) if use_pg else elapsed
        # one (start, end) event pair per network forward; --substreams S: S forwards per step, each timed on its own stream --
        # their SUM is the step's network time (sub-batches overlap with other stages, not with each other's convolutions)
        per_fwd = [ev[2 * i].elapsed_time(ev[2 * i + 1]) for i in range(len(ev) // 2)]
        nfwd = max(1, sum(1 for sl in sub_slices if sl.stop > sl.start)) if args.substreams > 1 else 1
        cnn_ms = [sum(per_fwd[i:i + nfwd]) for i in range(0, len(per_fwd) - nfwd + 1, nfwd)]
        return elapsed, cnn_ms, result

    # clock / allocator pre-warm (untimed, disclosed as `prewarm_s`): an idle MI355X needs ~0.1 s under load to reach
    # its sustained clocks -- a 20-step region is 60 ms -- and the first frames create the plan, workspaces and tables
    t_pre, k = time.perf_counter(), 0
    while time.perf_counter() - t_pre < args.prewarm:
        step(k); k += 1
        if k % 8 == 0:
            torch.cuda.synchronize()
    for k in range(max(args.warmup, args.streams)):
        step(k)
    own = []                                                       # this rank's own time of every region
    # settle (untimed, disclosed as `settle_regions_ms`): whole regions of the contract's own form -- no synchronisation inside --
    # until two consecutive ones agree to 3 % (at most four, at most ~3 s).  configs[2] was seen to run its first ~0.8 s of
    # unsynchronised regions 27 % slow on some boxes even after the pre-warm loop above (which synchronises every 8 steps):
    # 13.38, 13.42, 10.54, 10.55, 10.56 ms per step in consecutive regions (profiles/r03_g_config2_transient_bench.json); what is
    # reported is the steady state, and the regions after the contract one (`repeats`) show that it is one
    settle = []
    if args.settle:
        while len(settle) < 4 and sum(settle) < 3.0:               # (decided on the max-over-ranks region times only: every rank
            settle.append(timed_region(args.steps)[0])             # takes the same decision -- no rank-local clock in the condition)
            if len(settle) >= 2 and abs(settle[-1] - settle[-2]) <= 0.03 * settle[-2]:
                break
        own.clear()
    elapsed, cnn_ms, result = timed_region(args.steps)             # the contract region: `value` comes from here
    repeats = [timed_region(args.steps)[0] for _ in range(max(0, args.repeats))]
    per_rank_ms = mdist.gather_floats(own[0] / args.steps * 1e3, dev) if use_pg else [own[0] / args.steps * 1e3]

    # strong-scaling reading of the same path (config 1): a FIXED batch of --strong-frames frames sharded over the ranks,
    # each rank running its shard as consecutive batch-1 frames; barrier + synchronize on both sides, max over ranks
    strong = None
    if args.config == 1 and args.strong_frames > 0 and args.streams == 1:
        slo, shi = mdist.shard_frames(args.strong_frames, rank, world)
        times = []
        for _ in range(5):
            torch.cuda.synchronize()
            if use_pg:
                mdist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(shi - slo):
                step(k)
            torch.cuda.synchronize()
            if use_pg:
                mdist.barrier()
            torch.cuda.synchronize()
            t = time.perf_counter() - t0
            times.append(mdist.max_over_ranks(t, dev) if use_pg else t)
        t = float(np.median(times))
        strong = {"frames": args.strong_frames, "frames_of_rank0": shi - slo, "ms": round(t * 1e3, 4),
                  "frames_per_s": round(args.strong_frames / t, 3), "regions_ms": [round(x * 1e3, 4) for x in times],
                  "note": "fixed batch sharded over the ranks (dist.shard_frames), median of 5 regions; compare across --gpus N "
                          "for the strong-scaling curve (`value` is the weak-scaling one)"}

    # the opt-in arithmetic, same workload, same region form (N = 1 only: it is a second reading, not part of the contract): the
    # three-product fp16 split is NOT the arithmetic `value` is quoted on (22-bit operands)
    alt, alt_result = None, None
    if (world == 1 and B > 0 and cfg["dtype"] == "f32" and args.arithmetic is None and not args.net_opt and not args.no_alt_arithmetic
            and args.streams == 1 and args.substreams == 1):
        alt_model = MSI(weights=weights, coord_net=coord, device=dev, dtype=cfg["dtype"], input_type="PP" if cfg["kind"] == "pp" else "ODS")
        alt_model.net_options[_N.NET_OPT_F32_SPLIT_F16] = 0x3ffff
        for k in range(max(args.warmup, 3)):
            frame(None, alt_model)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(args.steps):
            alt_result = frame(None, alt_model)
        torch.cuda.synchronize()
        alt_t = time.perf_counter() - t0
        alt_plan = alt_model._plan(B, H, W, cin, nout, NGF)
        alt = {"name": "split_f16", "value": round(frames_total * args.steps / alt_t, 3), "unit": "faces/s" if cfg["kind"] == "pp" else "frames/s",
               "ms_per_step": round(alt_t / args.steps * 1e3, 4), "steps": args.steps,
               "layers_on_the_fp16_form": sum(is_f16_split(alt_plan.layer_kernel(i)[0]) for i in range(17)),
               "network_status": int(alt_model.network_status()),
               "note": "plan option F32_SPLIT_F16 (bench.py --arithmetic split_f16): 2-way fp16 split of both operands (22 significand bits), THREE products on the fp16 MFMA, "
                       "fp32 accumulation -- measured against fp64 it has the error of a plain fp32 convolution (profiles/r04_split_numerics.txt), operands limited to the "
                       "fp16 range (status word); opt-in, reported beside the default, never as `value`"}
    ranges = mdist.gather_ranges(lo, hi, dev) if use_pg else [(lo, hi)]
    nccl_world = torch.distributed.get_world_size() if use_pg else 1
    backend = torch.distributed.get_backend() if use_pg else None
    if use_pg:                       # the last collective is behind us: every rank leaves the group together (rank 0 goes on alone with the per-stage pass)
        mdist.barrier()
        torch.distributed.destroy_process_group()
    if rank != 0:
        return

    # per-stage HIP events: a separate, untimed pass (one frame batch at a time on the launch stream)
    torch.cuda.synchronize()
    all_events = []
    for _ in range(5):
        ev = []
        frame(ev)
        all_events.append(ev)
    torch.cuda.synchronize()
    stage_ms = {}
    for si, name in enumerate(stage_names):
        stage_ms[name] = float(np.mean([ev[si].elapsed_time(ev[si + 1]) for ev in all_events])) if B > 0 else 0.0
    ms_per_step = elapsed / args.steps * 1e3
    fps = frames_total * args.steps / elapsed
    cnn_ms_timed = float(np.mean(cnn_ms)) if cnn_ms else stage_ms["cnn"]   # (streams > 1: from the separate pass)

    bf16 = cfg["dtype"] == "bf16"
    peak = PEAK_BF16_MFMA_TFLOPS if bf16 else PEAK_FP32_MFMA_TFLOPS
    flops = conv_flops(H, W, cin, nout, NGF, coord)                # per frame: the 17 convolutions the interval contains
    gbytes = geometry_bytes(H, W, D, 2 if bf16 else 4)
    nb = max(B, 1)
    cnn_tflops = flops * nb / (cnn_ms_timed * 1e-3) / 1e12
    # the peak the convolutions are priced against: per layer the dense MFMA peak of the instruction it runs on -- bf16 plans
    # 2 500 TFLOP/s; fp32 layers 157.3 on v_mfma_f32_32x32x2_f32, or 2 500 / 6 = 416.7 fp32-equivalent TFLOP/s where the plan
    # runs the six-product 3-way bf16 split (conv_halo_x3_kernel) -- blended by the layers' flops: peak = flops / sum(flops_l / peak_l)
    layer_fl = cnn_layer_flops(H, W, cin, nout, NGF, coord)[:-1]
    if B > 0 and not bf16:
        kplan = model._plan(B, H, W, cin, nout, NGF)
        layer_peak = [split_peak(kplan.layer_kernel(i)[0]) for i in range(17)]
        peak = sum(layer_fl) / sum(f / pk for f, pk in zip(layer_fl, layer_peak))
    stages = {k: {"ms": round(v, 4)} for k, v in stage_ms.items()}
    for k in ("sweep", "assemble", "render"):
        if stage_ms[k] > 0:
            gbs = gbytes[k] * nb / (stage_ms[k] * 1e-3) / 1e9
            stages[k].update({"bound": "hbm", "algorithmic_MB": round(gbytes[k] * nb / 1e6, 1),
                              "achieved_GBps": round(gbs, 1), "frac": round(gbs / PEAK_HBM_GBS, 4)})
    stages["cnn"].update({"bound": "mfma", "algorithmic_GFLOP": round(flops * nb / 1e9, 1),
                          "achieved_TFLOPps": round(flops * nb / (max(stage_ms["cnn"], 1e-9) * 1e-3) / 1e12, 2)})
    traffic, traffic_src, traffic_stale = cnn_traffic(args.config, coord)
    conv_bytes = conv_algorithmic_bytes(H, W, cin, NGF, coord, nb, act_bytes=2 if bf16 else 4, w_bytes=2 if bf16 else 4)
    if args.substreams > 1:   # the per-stage pass runs the whole batch on one stream: not the configuration that was timed
        stages["note"] = "per-stage pass = whole batch on ONE stream (substreams only changes the timed region)"

    sustained = None if args.no_sustained_probe else sustained_matrix_rate(dev)
    unit = "faces/s" if cfg["kind"] == "pp" else "frames/s"
    arithmetic = None
    nx2 = nx3 = 0
    if B > 0:
        plan = model._plan(B, H, W, cin, nout, NGF)
        kern = [plan.layer_kernel(i)[0] for i in range(17)]
        nx2 = sum(is_f16_split(k) for k in kern)
        nx3 = sum("_x3_kernel" in k for k in kern) - nx2
        arithmetic = ("bf16 operands, fp32 accumulate (v_mfma_f32_32x32x16_bf16)" if bf16 else
                      "fp32: %d of 17 convolutions as a 2-way fp16 split with 3 products on the fp16 MFMA (22-bit operands), %d as a 3-way bf16 split "
                      "with 6 products on the bf16 MFMA (both fp32 accumulate, fp32-grade), %d on the native fp32 MFMA" % (nx2, nx3, 17 - nx2 - nx3)
                      if nx2 + nx3 else "native fp32 MFMA (v_mfma_f32_32x32x2_f32)")
    metric = "novel-view frames/sec, 640x320 ODS->32-sphere MSI infer+render" if args.config == 1 else \
        "novel-view %s, %dx%d %s->%d-%s infer+render" % (unit.replace("/s", "/sec"), W, H, "PP face" if cfg["kind"] == "pp" else "ODS",
                                                        D, "plane MPI" if cfg["kind"] == "pp" else "sphere MSI")
    line = {
        "metric": metric,
        "value": round(fps, 3), "unit": unit, "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "prewarm_s": args.prewarm, "settle_regions_ms": [round(x / args.steps * 1e3, 4) for x in settle], "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
        "scaling": cfg["scaling"], "vs_baseline": None, "dtype": cfg["dtype"], "data": "synthetic",
        "config": {"workload": cfg["name"] + ", " + ("CoordNet" if coord else "wrap-pad net") + ", infer + "
                               + ("MPI render" if cfg["kind"] == "pp" else "RGB&depth render"),
                   "baseline_config_index": args.config, "height": H, "width": W,
                   "num_spheres": D, "ngf": NGF, "frames_per_step": frames_total, "frames_per_step_rank0": B,
                   "parallelism": "frames sharded over %d GPU(s) (dist.shard_frames), no data-path collective" % world,
                   "streams_per_gpu": args.streams, "substreams": args.substreams,
                   "arithmetic": arithmetic, "net_options": args.net_opt or None},
        "distributed": {"world_size_env": world, "world_size_process_group": nccl_world, "backend": backend,
                        "frame_ranges_per_rank": ranges,
                        "per_rank_ms_per_step": [round(x, 4) for x in per_rank_ms],
                        "rank_skew": {"max_ms": round(max(per_rank_ms), 4), "min_ms": round(min(per_rank_ms), 4),
                                      "max_over_min": round(max(per_rank_ms) / max(min(per_rank_ms), 1e-9), 4)},
                        "cpu_affinity_rank0": affinity,
                        "weight_broadcast_ms": None if broadcast_ms is None else round(broadcast_ms, 3),
                        "devices_visible": torch.cuda.device_count(), "device_of_rank0": torch.cuda.get_device_name(dev)},
        "repeats": {"ms_per_step": [round(r / args.steps * 1e3, 4) for r in [elapsed] + repeats],
                    "median_ms_per_step": round(float(np.median([elapsed] + repeats)) / args.steps * 1e3, 4),
                    "note": "region 0 is the contract region `value` / `ms_per_step` are computed from"},
        "roofline": {"kernel": "conv_halo*_kernel + conv_igemm_kernel (the 3x3 / 4x4 conv launches of one forward, %s implicit GEMM)" % ("bf16 MFMA" if bf16 else str(arithmetic)),
                     "bound": "mfma", "achieved": round(cnn_tflops, 3), "peak": round(peak, 2),
                     "peak_note": "dense MFMA peak of the instruction each layer runs on, blended by the layers' flops (fp32 MFMA 157.3; six-product bf16 split 2500 / 6 = 416.7, three-product fp16 split 2500 / 3 = 833.3 fp32-equivalent; bf16 2500 TFLOP/s) -- all quoted at the 2.4 GHz peak clock "
                                  "on constant operands.  On operands that change between consecutive MFMAs the part itself clocks 1.76-1.89 GHz and a matrix-only loop sustains 1.72-1.91 PFLOP/s (sustained_matrix_rate below is THIS box's figure; "
                                  "s_memtime counts real shader cycles, the driver's sclk reading is stale on this pool: profiles/r06_clock.txt); the six-product kernels run at that same clock, so what separates `frac_of_sustained` from 1 is skeleton "
                                  "(launch ramp, tile quantisation, prologue / epilogue: profiles/r06_residency.txt), not power",
                     "unit": "TFLOP/s", "frac": round(cnn_tflops / peak, 4),
                     "frac_of_fp32_mfma_peak": None if bf16 else round(cnn_tflops / PEAK_FP32_MFMA_TFLOPS, 4),
                     "frac_of_sustained": round(cnn_tflops / (peak * sustained["changing"]["pflops"] * 1e3 / PEAK_BF16_MFMA_TFLOPS), 4) if (sustained and (bf16 or nx2 + nx3 == 17)) else None,
                     "sustained_matrix_rate": None if sustained is None else dict(sustained, note="msi_probe_matrix_rate on THIS box after the timed regions: matrix-only loop (no memory / LDS traffic), bf16 MFMA back to back, on operands that "
                                                   "change between consecutive instructions vs constant ones; dense PFLOP/s and the shader clock (s_memtime cycles per s_memrealtime second). "
                                                   "frac_of_sustained = achieved / (peak x changing.pflops / 2.5): an EXTRA key, `frac` stays on the nominal peak (profiles/r06_clock.txt)"),
                     "traffic": traffic,
                     "traffic_stale": traffic_stale,
                     "algorithmic_bytes": conv_bytes,
                     "traffic_ratio": None if traffic is None else round(traffic / conv_bytes, 3),
                     "traffic_note": None if traffic is None else
                     "HBM bytes per step (%d frame(s)) of the conv launches, %s (separate --pmc FETCH_SIZE / WRITE_SIZE passes, "
                     "gfx950 FETCH correction)%s" % (nb, traffic_src, "; STALE: measured with other kernel sources than this "
                     "run's (csrc_sha %s)" % csrc_hash() if traffic_stale else ""),
                     "launches_per_frame": 17, "algorithmic_flops_per_launch_set": flops * nb,
                     "flops_note": "conv1_1 ... conv8_2 only; color_pred (the 1x1 head, %.2f GFLOP per frame) runs in the fused tail "
                                   "after the closing event and is not counted" % (cnn_layer_flops(H, W, cin, nout, NGF, coord)[-1] / 1e9),
                     "ms_per_forward": round(cnn_ms_timed, 4),
                     "timed": "HIP events around msi_net_plan_forward on the launch stream INSIDE the timed region, mean of "
                              "%d forwards (conv launches + the remaining ln_apply launches: conservative for the conv kernels alone)" % max(len(cnn_ms), 1)},
        "stages": stages,
        "strong_scaling": strong,
        "alt_arithmetic": alt,
    }

    if world == 1 and args.config == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"], parity, alt_parity = cpu_baseline(model, weights, inp, planes, coord, result, D, alt_result)
        line["parity_max_abs_vs_oracle"] = parity
        if alt is not None:
            alt["parity_max_abs_vs_oracle"] = alt_parity
    else:
        line["cpu_baseline"] = None
    print(json.dumps(line), flush=True)


def cpu_baseline(model, weights, inp, planes, coord, gpu_result, D, alt_result=None):
    """The CPU oracle ("port": the reference itself needs Python 2 + TF 1.14 and cannot run here)
    on the same workload, timed on this box's host cores: a bounded sample of 2 frames (~10-30 s)
    after a small warm-up that creates the thread pool / conv primitives.  torch-CPU conv runs on
    min(cores, 64) threads (256 threads are 10x slower on this host: oversubscription), the numpy
    geometry is single-threaded."""
    from oracle.msi import MSI as OracleMSI
    from oracle import nets as onets
    if _ORIG_AFFINITY:                      # the CPU leg runs on the whole host again, not on the launch thread's NUMA node
        os.sched_setaffinity(0, _ORIG_AFFINITY)
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
    def par(res):
        rgb, dep, _, _, out = res
        return {"rgba_layers": float(np.abs(out["rgba_layers"].cpu().numpy() - pred_o["rgba_layers"]).max()),
                "rgb": float(np.abs(rgb.cpu().numpy() - rgb_o).max()),
                "depth": float(np.abs(dep.cpu().numpy() - dep_o).max())}
    parity = par(gpu_result)
    alt_parity = par(alt_result) if alt_result is not None else None
    base = {"value": round(1.0 / t, 5), "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": "%d frames of the same workload (640x320, 32 spheres, infer + rgb & depth render), %.1f s per "
                      "frame; torch-CPU conv on %d of %d host cores, numpy geometry single-threaded"
                      % (nframes, t, threads, cores)}
    return base, parity, alt_parity


if __name__ == "__main__":
    main()
