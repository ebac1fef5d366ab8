"""First contact with RCCL on the one GPU a test box has.

Every other multi-rank test forces `gloo` (two ranks cannot share a device under RCCL), so until round 6 the `nccl` branch of
matryodshka_amd/dist.py -- `device_id=`, collectives on DEVICE tensors -- had never executed anywhere and the driver's 8-GPU run would
have been its first execution.  world_size 1 is enough to go through all of it: communicator creation on the device, the flat 68-MB
weight broadcast, the all_gathers of the frame ranges / per-rank times, the MAX all_reduce and the barrier of the timing protocol.
Each case runs in its own process (a process group is process-wide state)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import json, os, sys, time
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.getcwd())
from matryodshka_amd import dist as mdist, nets
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
mdist.init_process_group()                      # no backend given: nccl (= RCCL) because a HIP device is present
assert dist.get_backend() == "nccl", dist.get_backend()
cin, nout, ngf = 192, 64, 64                    # BASELINE configs[1]: the real 16.98 M-parameter blob
w0 = nets.init_weights(cin, nout, ngf, True, seed=8964)
torch.cuda.synchronize(); t0 = time.perf_counter()
w = mdist.broadcast_weights(w0, cin, nout, ngf, True, dev, src=0)
torch.cuda.synchronize(); first_ms = (time.perf_counter() - t0) * 1e3
t0 = time.perf_counter()
w = mdist.broadcast_weights(w0, cin, nout, ngf, True, dev, src=0)
torch.cuda.synchronize(); second_ms = (time.perf_counter() - t0) * 1e3
blob0 = nets.flatten_params(w0, cin, nout, ngf, True)
blob = nets.flatten_params(w, cin, nout, ngf, True)
# the collectives really ran on device tensors
t = torch.arange(8, dtype=torch.float32, device=dev)
dist.broadcast(t, src=0)
dist.all_reduce(t, op=dist.ReduceOp.SUM)
lo, hi, total = mdist.step_frames(32, None, 0, 1)
out = {"backend": dist.get_backend(), "world": dist.get_world_size(), "numel": int(blob.size), "bytes": int(blob.nbytes),
       "equal": bool(np.array_equal(blob, blob0)), "first_ms": first_ms, "second_ms": second_ms,
       "ranges": mdist.gather_ranges(lo, hi, dev), "floats": mdist.gather_floats(1.25, dev), "max": mdist.max_over_ranks(3.5, dev),
       "allreduce": t.cpu().tolist(), "coll_device": str(mdist._coll_device(dev))}
mdist.barrier()
dist.destroy_process_group()
print("RCCL_N1 " + json.dumps(out), flush=True)
"""


def _env():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = {k: v for k, v in os.environ.items() if k not in ("MSI_DIST_BACKEND",)}
    env.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def test_rccl_world_size_1_runs_every_collective_of_dist_py_on_the_device():
    p = subprocess.run([sys.executable, "-c", WORKER], env=_env(), cwd=ROOT, timeout=600, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    lines = [l for l in p.stdout.splitlines() if l.startswith("RCCL_N1 ")]
    assert p.returncode == 0 and len(lines) == 1, (p.returncode, p.stdout[-2000:], p.stderr[-4000:])
    j = json.loads(lines[0][len("RCCL_N1 "):])
    print("RCCL world_size 1: 67.9-MB weight broadcast %.1f ms (first, with communicator set-up), %.1f ms (second)" % (j["first_ms"], j["second_ms"]))
    assert j["backend"] == "nccl" and j["world"] == 1 and j["coll_device"].startswith("cuda")
    assert j["numel"] == 16980160                                        # the reference-width network: 67.9 MB of fp32
    assert j["bytes"] == 4 * j["numel"] and j["equal"] is True
    assert j["ranges"] == [[0, 32]] and j["floats"] == [1.25] and j["max"] == 3.5
    assert j["allreduce"] == [float(i) for i in range(8)]


def test_bench_line_over_a_forced_rccl_process_group():
    """`bench.py --force-process-group`: the N = 1 line carries distributed.backend "nccl", a measured weight_broadcast_ms, and the barriers /
    MAX-over-ranks of the timed region went through RCCL."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "MSI_DIST_BACKEND")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--force-process-group", "--steps", "5", "--warmup", "2", "--prewarm", "0.2",
                        "--repeats", "1", "--no-settle", "--no-cpu-baseline", "--no-alt-arithmetic", "--strong-frames", "0"],
                       env=env, cwd=ROOT, timeout=900, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert p.returncode == 0 and len(lines) == 1, (p.returncode, p.stdout[-2000:], p.stderr[-4000:])
    j = json.loads(lines[0])
    d = j["distributed"]
    assert d["backend"] == "nccl" and d["world_size_process_group"] == 1 and d["frame_ranges_per_rank"] == [[0, 1]]
    assert d["weight_broadcast_ms"] is not None and d["weight_broadcast_ms"] > 0
    assert j["n_gpus"] == 1 and j["value"] > 0 and j["steps"] == 5
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "rccl_n1_bench.json"), "w") as f:
            f.write(lines[0] + "\n")
