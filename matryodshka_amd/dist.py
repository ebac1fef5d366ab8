"""Multi-GPU plumbing: one process per GPU, frames sharded over ranks, ONE collective
(the start-up weight broadcast; `nccl` == RCCL over xGMI on ROCm, `gloo` on CPU for
the tests).  The reference is single-device; frames are independent --
LayerNorm is per sample, sweep and render per frame -- so there is no per-frame
collective to add.
"""
import os

import numpy as np
import torch
import torch.distributed as dist


def init_process_group(backend=None):
    if dist.is_initialized():
        return
    if backend is None:
        # MSI_DIST_BACKEND=gloo lets two ranks share one GPU (functional test of the N>1 path on a
        # 1-GPU box; RCCL refuses duplicate devices)
        backend = os.environ.get("MSI_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    kwargs = {}
    if backend == "nccl":
        kwargs["device_id"] = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    dist.init_process_group(backend=backend, **kwargs)


def shard_frames(num_frames, rank, world_size):
    """Contiguous frame range [lo, hi) of `rank`; sizes differ by at most one."""
    base, rem = divmod(int(num_frames), int(world_size))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def step_frames(total, per_rank, rank, world_size):
    """Global frame indices [lo, hi) this rank renders in ONE bench step and the step's total:
    strong scaling (`total` frames per step sharded over the ranks, BASELINE configs[3]/[4]) or weak scaling
    (`per_rank` frames on every rank, configs[1]/[2])."""
    if total is not None:
        lo, hi = shard_frames(total, rank, world_size)
        return lo, hi, int(total)
    return rank * per_rank, (rank + 1) * per_rank, per_rank * world_size


def gather_ranges(lo, hi, device):
    """[(lo, hi)] of every rank (one small all_gather; used by the tests and by bench.py's report)."""
    t = torch.tensor([int(lo), int(hi)], dtype=torch.int64, device=_coll_device(device))
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [(int(o[0]), int(o[1])) for o in out]


def gather_floats(value, device):
    """[value of rank 0, rank 1, ...] (one small all_gather: bench.py's per-rank step times)."""
    t = torch.tensor([float(value)], dtype=torch.float64, device=_coll_device(device))
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(o.item()) for o in out]


def _coll_device(device):
    """Collectives run on the GPU with nccl (RCCL) and on the host with gloo."""
    return torch.device("cpu") if dist.get_backend() == "gloo" else device


def broadcast_blob(blob, numel, device, src=0):
    """Broadcast a flat fp32 parameter blob (numpy on `src`, ignored elsewhere)."""
    device = _coll_device(device)
    if dist.get_rank() == src:
        t = torch.from_numpy(np.ascontiguousarray(blob, dtype=np.float32)).to(device)
        assert t.numel() == numel
    else:
        t = torch.empty(numel, dtype=torch.float32, device=device)
    dist.broadcast(t, src=src)
    return t.cpu().numpy()


def broadcast_weights(weights, in_channels, num_outputs, ngf, coord_net, device, src=0):
    """All ranks end up with rank `src`'s weight dict (one flat broadcast, ~68 MB fp32 at
    the reference width).  Needs the native library only for the variable table."""
    from . import nets
    shapes = nets.variable_shapes(in_channels, num_outputs, ngf, coord_net)
    numel = int(sum(int(np.prod(s)) for _, s in shapes))
    blob = nets.flatten_params(weights, in_channels, num_outputs, ngf, coord_net) if dist.get_rank() == src else None
    blob = broadcast_blob(blob, numel, device, src)
    return nets.unflatten_params(blob, in_channels, num_outputs, ngf, coord_net)


def barrier():
    if dist.is_initialized():
        dist.barrier()


def max_over_ranks(value, device):
    t = torch.tensor([float(value)], dtype=torch.float64, device=_coll_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
