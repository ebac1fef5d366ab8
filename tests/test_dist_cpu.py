"""The N>1 path on CPU: world_size-2 gloo processes exercise frame sharding, the flat weight
broadcast and the max-over-ranks timing reduction of matryodshka_amd/dist.py (on the GPU box the
same code runs with backend nccl == RCCL over xGMI)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from matryodshka_amd import dist as mdist, nets
    mdist.init_process_group(backend="gloo")
    cin, nout, ngf = 24, 8, 8
    w = nets.init_weights(cin, nout, ngf, True, seed=100 + rank)      # ranks start out different
    w = mdist.broadcast_weights(w if rank == 0 else None, cin, nout, ngf, True, torch.device("cpu"), src=0)
    blob = nets.flatten_params(w, cin, nout, ngf, True)
    lo, hi = mdist.shard_frames(7, rank, world)
    t = mdist.max_over_ranks(1.0 + rank, torch.device("cpu"))
    # the sharded bench loop (bench.py --config 3 / 4: B = 32 / 64 frames per step over the ranks; --config 1 / 2: weak)
    loops = {}
    for name, total, per_rank in (("config3", 32, None), ("config4", 64, None), ("config1", None, 1), ("odd", 7, None)):
        flo, fhi, ftotal = mdist.step_frames(total, per_rank, rank, world)
        rendered = list(range(flo, fhi))                 # what frame() would be called with
        loops[name] = (mdist.gather_ranges(flo, fhi, torch.device("cpu")), ftotal, rendered)
    mdist.barrier()
    q.put((rank, float(blob.sum()), blob.size, (lo, hi), t, loops))
    dist.destroy_process_group()


def test_shard_frames_partitions_exactly():
    from matryodshka_amd.dist import shard_frames
    for n in (0, 1, 7, 8, 32, 65):
        for world in (1, 2, 3, 8):
            parts = [shard_frames(n, r, world) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(parts[:-1], parts[1:]))
            sizes = [hi - lo for lo, hi in parts]
            assert max(sizes) - min(sizes) <= 1


def test_two_rank_gloo_broadcast_and_reduce(native_lib):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from matryodshka_amd import nets
    ref = nets.flatten_params(nets.init_weights(24, 8, 8, True, seed=100), 24, 8, 8, True)
    assert res[0][1] == res[1][1] == float(ref.sum()) and res[0][2] == ref.size   # both hold rank 0's weights
    assert res[0][3] == (0, 4) and res[1][3] == (4, 7)
    assert res[0][4] == res[1][4] == 2.0
    # sharded loop: every rank sees the same partition; the union of the ranks' frames is exactly the step's batch
    for name, total in (("config3", 32), ("config4", 64), ("config1", 2), ("odd", 7)):
        ranges0, total0, frames0 = res[0][5][name]
        ranges1, total1, frames1 = res[1][5][name]
        assert ranges0 == ranges1 and total0 == total1 == total
        assert sorted(frames0 + frames1) == list(range(total)) and not set(frames0) & set(frames1)
        assert [tuple(r) for r in ranges0] == [(frames0[0], frames0[-1] + 1), (frames1[0], frames1[-1] + 1)]


def _run_bench(args, timeout=300):
    """`python bench.py <args>` exactly as a driver types it: no launcher, no RANK / WORLD_SIZE in the environment."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["MSI_DIST_BACKEND"] = "gloo"
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, env=env, cwd=root, timeout=timeout,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert p.returncode == 0 and len(lines) == 1, (p.returncode, p.stdout[-2000:], p.stderr[-4000:])
    return json.loads(lines[0])


def test_bench_gpus_2_launches_itself(native_lib):
    """`python bench.py --gpus 2` must become two ranks by itself (it used to exit).  --dist-check
    runs everything `--gpus N` adds to the one-GPU bench -- launch, rendezvous, weight broadcast, frame ranges, barrier,
    max over ranks -- without the frame loop, so it runs here on gloo without a GPU; the full line is checked on the GPU
    box by tests/test_gpu_pipeline.py::test_bench_gpus_2_end_to_end_on_one_gpu."""
    j = _run_bench(["--gpus", "2", "--dist-check", "--config", "3"])
    assert j["world_size_process_group"] == 2 and j["n_gpus"] == 2 and j["backend"] == "gloo"
    assert j["frame_ranges_per_rank"] == [[0, 16], [16, 32]] and j["frames_per_step"] == 32
    assert j["max_over_ranks"] == 2.0 and j["weights_equal_on_all_ranks"] is True
    j = _run_bench(["--gpus", "2", "--dist-check"])                 # config 1: weak scaling, one frame per rank
    assert j["frame_ranges_per_rank"] == [[0, 1], [1, 2]] and j["frames_per_step"] == 2


def test_bench_refuses_rccl_on_too_few_devices():
    """Without MSI_DIST_BACKEND the N-rank run is RCCL, one rank per GPU: fewer visible devices is a loud error with
    exit code 2, never a silent single-rank number."""
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() >= 2:
        return
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MSI_DIST_BACKEND")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env, cwd=root, timeout=120,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert p.returncode == 2 and "HIP device(s) visible" in p.stderr and not p.stdout.strip()
