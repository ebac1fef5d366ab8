"""The plans bench.py actually times ("plan coverage hole").

The network's plan depends on the BATCH: which kernel a layer takes (stride-2 halo kernel only when tiles x batch >= 3 CUs,
bf16 128x128 tiles only when >= 4 tiles per CU) and how its tiles are split.  The full-size fixtures hold b = 1 / 2 frames
(tests/test_golden.py); bench.py --config 2 / 3 / 4 runs b = 16 / 32 / 64 per GPU and b = 4 / 8 as the 8-GPU shard of
configs[3] / [4].  Here the fixture's frames are REPLICATED to those batches, the DEFAULT plan runs the whole batch, EVERY
frame is compared with the committed dense oracle samples at the fixture's tolerance (on the device: the b = 32 layer
stack is 13 GB), and msi_net_plan_layer_kernel says which instantiations ran -- the lists below are the ones in
profiles/*_config{2,3,4}_kernel_stats.txt, so every kernel variant of the bench tables appears in a parity test at
that shape.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 1e-3

# graph order conv1_1 ... conv8_2 (the head runs inside head_assemble_kernel on the blend_psv path)
# (r04: the stride-1 halo layers of an fp32 plan run the six-product bf16 split by default -- conv_halo_x3_kernel)
# (the kernels' last template argument: operand planes -- 3 = bf16 h | m | l, six products; 2 = fp16 h | m', three products)
# (r05: the stride-1, rate-1 layers of the six-product form whose grid is >= 3 tiles of 8 x 16 pixels per CU run conv_halo8_x3_kernel -- plan option X3_TILE8:
#  every such layer at the batches of configs[3] / [4]; conv1_1, conv2_1, conv7_2, conv8_2 at configs[1])
# (r05: the rate-2 layers conv4_x run on row-parity tiles -- plan option X3_ROWPAR, first template argument 3: dilation 2 along W, row stride 2 along H)
def f32_big_grid(np_, wide=(0, 2, 4, 5, 11, 12, 14, 16), wide_ct=(10, 13, 15), wide_s2=(1, 3, 6)):
    c = lambda r, a: "conv_halo_x3_kernel<%d, %d, %d>" % (r, a, np_)      # noqa: E731
    s2, ct = "conv_halo_s2_x3_kernel<1, %d>" % np_, "convt_halo_x3_kernel<%d>" % np_
    k = [c(1, 0), s2, c(1, 1), s2, c(1, 1), c(1, 1), s2, c(3, 1), c(3, 1), c(3, 1), ct, c(1, 1), c(1, 1), ct, c(1, 1), ct, c(1, 1)]
    if np_ == 3:
        for i in wide:
            k[i] = "conv_halo8_x3_kernel<%d, 3>" % (0 if i == 0 else 1)
        for i in wide_ct:          # (r05: the conv-transposes and the stride-2 layers on the 8 x 16-pixel tile under the same grid rule)
            k[i] = "convt_halo8_x3_kernel"
        for i in wide_s2:
            k[i] = "conv_halo8_s2_x3_kernel<1>"
    return k


DEFAULT_PLANES = 3          # (plan option F32_SPLIT_F16 -- the three-product fp16 form, 2 planes -- is opt-in)
F32_BIG_GRID = f32_big_grid(DEFAULT_PLANES)
F32_CONFIG1 = f32_big_grid(DEFAULT_PLANES, wide=(0, 2, 14, 16), wide_ct=(15,), wide_s2=(1,))
BF16_CONFIG2 = ["conv_halo_bf16_kernel<256, 64, 1, 0, 4>", "conv_halo_bf16_s2_kernel<1, 4>", "conv_halo_bf16_kernel<128, 128, 1, 0, 8>",
                "conv_halo_bf16_s2_kernel<1, 4>", "conv_halo_bf16_kernel<128, 128, 1, 0, 8>", "conv_halo_bf16_kernel<128, 128, 1, 1, 8>",
                "conv_halo_bf16_s2_kernel<1, 4>", "conv_halo_bf16_kernel<128, 128, 2, 0, 8>", "conv_halo_bf16_kernel<128, 128, 2, 1, 8>",
                "conv_halo_bf16_kernel<128, 128, 2, 1, 8>", "convt_halo_bf16_kernel<128, 128, 0>", "conv_halo_bf16_kernel<128, 128, 1, 1, 8>",
                "conv_halo_bf16_kernel<128, 128, 1, 1, 8>", "convt_halo_bf16_kernel<128, 128, 0>", "conv_halo_bf16_kernel<128, 128, 1, 1, 8>",
                "convt_halo_bf16_kernel<128, 64, 0>", "conv_halo_bf16_kernel<256, 64, 1, 1, 4>"]


def _fixture(name):
    path = os.path.join(HERE, name)
    if not os.path.exists(path):
        pytest.skip("%s not generated (tests/golden/make_golden.py --configs ...)" % name)
    z = np.load(path, allow_pickle=True)
    return z, {k: v for k, v in z["cfg"]}


def _tile(inp, reps):
    return {k: np.concatenate([v] * reps, axis=0) for k, v in inp.items()}


def _check_every_frame(z, key, tensor, bfix, max_tol, mean_tol=None, seed_bump=0):
    """tensor [reps * bfix, H, W, ...] on the device: every group of bfix frames against the fixture's dense samples."""
    import torch
    from tests.util import stratified_index
    extra = z["extra_pixels"] if "extra_pixels" in z.files else None
    shape = (bfix,) + tuple(tensor.shape[1:])
    assert shape == tuple(int(v) for v in z["shape_" + key]), (key, shape)
    idx = torch.from_numpy(stratified_index(shape, extra, int(z["sample_seed"]) + seed_bump)).to(tensor.device)
    want = torch.from_numpy(z["val_" + key].astype(np.float64)).to(tensor.device)
    assert idx.numel() == want.numel()
    worst, worst_mean = 0.0, 0.0
    for r in range(tensor.shape[0] // bfix):
        got = tensor[r * bfix:(r + 1) * bfix].contiguous().reshape(-1)[idx].double()
        err = (got - want).abs()
        assert bool(torch.isfinite(got).all()), (key, r)
        worst, worst_mean = max(worst, float(err.max())), max(worst_mean, float(err.mean()))
        assert float(err.max()) <= max_tol, (key, "frame group", r, float(err.max()))
        if mean_tol is not None:
            assert float(err.mean()) <= mean_tol, (key, "frame group", r, float(err.mean()))
    return worst, worst_mean, int(idx.numel()), tensor.shape[0]


def _plan_kernels(model, b, h, w, d, ngf=64):
    plan = model._plan(b, h, w, 6 * d, 2 * d, ngf)
    return [plan.layer_kernel(i) for i in range(17)]


def _ods_batch(cfg, batch, dtype="f32"):
    import torch
    from matryodshka_amd import MSI
    from matryodshka_amd.synthetic import make_inputs
    from oracle import nets as onets      # weights are INPUTS, drawn from the documented seeded generator
    bfix, h, w, d, ngf, seed = (int(cfg[k]) for k in ("b", "h", "w", "d", "ngf", "seed"))
    assert batch % bfix == 0
    inp = _tile(make_inputs(seed, bfix, h, w), batch // bfix)
    weights = onets.init_weights(6 * d, 2 * d, ngf=ngf, coord_net=True, seed=seed, randomize_affine=True)
    m = MSI(weights=weights, coord_net=True, dtype=dtype)
    planes = m.inv_depths(1.0, 100.0, d)
    pred, net_input = m.infer_msi(torch.from_numpy(inp["src_image"]), torch.from_numpy(inp["ref_image"]), None, None,
                                  inp["ref_pose"], inp["src_pose"], inp["intrinsics"], "blend_psv", d, planes, ngf=ngf)
    rgb, dep = m.msi_render_equirect_view_and_depth(pred["rgba_layers"], inp["tgt_pose_rt"], inp["tgt_pos"], planes,
                                                    inp["intrinsics"])
    torch.cuda.synchronize()
    assert m.network_status() == 0
    return m, dict(psv=net_input, rgba_layers=pred["rgba_layers"], rgb=rgb, depth=dep), (bfix, h, w, d)


def test_config2_bench_batch_16_bf16_every_frame():
    """bench.py --config 2: 640x320, 64 spheres, batch 16, bf16 -- the fixture's two frames eight times."""
    z, cfg = _fixture("full_config2_bf16_640x320x64_b2_samples.npz")
    m, got, (bfix, h, w, d) = _ods_batch(cfg, 16, dtype="bf16")
    kern = _plan_kernels(m, 16, h, w, d)
    assert [k[0] for k in kern] == BF16_CONFIG2, kern
    assert all(k[2] == 0 for k in kern)                                  # whole tiles only at this grid
    rep = {"psv": _check_every_frame(z, "psv", got["psv"].float(), bfix, 2.0 ** -7, 1e-5)}
    for k in ("rgba_layers", "rgb", "depth"):
        rep[k] = _check_every_frame(z, k, got[k], bfix, 6e-2, 3e-3)
    print("config2 b=16 (max, mean, samples per group, frames):", rep)


@pytest.mark.parametrize("batch", [32, 4])
def test_config3_bench_batches_every_frame(batch):
    """bench.py --config 3: 1280x640, 32 spheres, fp32; 32 frames on one GPU, 4 = a rank's shard on 8 GPUs."""
    import torch
    z, cfg = _fixture("full_config3_1280x640x32_samples.npz")
    m, got, (bfix, h, w, d) = _ods_batch(cfg, batch)
    kern = _plan_kernels(m, batch, h, w, d)
    assert [k[0] for k in kern] == F32_BIG_GRID, kern
    assert all(k[2] == 0 for k in kern if not k[0].startswith(("convt_halo8_x3_kernel", "conv_halo8_s2_x3_kernel")))   # (whole tiles; the 8-row conv-transpose / stride-2 tiles of a rank's 4-frame shard end in a K-range tail)
    rep = {k: _check_every_frame(z, k, got[k], bfix, TOL) for k in ("psv", "rgba_layers", "rgb", "depth")}
    print("config3 b=%d (max, mean, samples per group, frames):" % batch, rep)
    del got
    torch.cuda.empty_cache()


@pytest.mark.parametrize("batch", [64, 8])
def test_config4_bench_batches_every_face(batch):
    """bench.py --config 4: PP cube faces 256x256, 32 planes, fp32; 64 faces on one GPU, 8 = a rank's shard on 8 GPUs.
    (With the native arithmetic -- plan option F32_SPLIT3 = 0 -- conv3_3 takes conv_halo_s2_kernel at these batches and the tap
    kernel in the b = 2 fixture test; the default six-product plan runs the halo form at every batch.)"""
    import torch
    from matryodshka_amd import MSI, poses
    from oracle import nets as onets
    from tests.golden.make_golden import pp_inputs
    z, cfg = _fixture("full_config4_pp_256x256x32_b2_samples.npz")
    bfix, n, d, ngf, seed = (int(cfg[k]) for k in ("b", "n", "d", "ngf", "seed"))
    reps = batch // bfix
    ref, src, K, eye, src_pose, tgt_pose = (np.concatenate([a] * reps, axis=0) for a in pp_inputs(seed, bfix, n))
    weights = onets.init_weights(6 * d, 2 * d, ngf=ngf, coord_net=True, seed=seed, randomize_affine=True)
    m = MSI(weights=weights, coord_net=True, input_type="PP")
    planes = m.inv_depths(1.0, 100.0, d)
    interp_inv = np.linalg.inv(poses.interpolate_pose(eye, src_pose).astype(np.float64)).astype(np.float32)
    pred, net_input = m.infer_msi(torch.from_numpy(src), torch.from_numpy(ref), None, None, eye, src_pose, K, "blend_psv",
                                  d, planes, ngf=ngf, ref_pose_inv=interp_inv)
    rgb = m.mpi_render_view(pred["rgba_layers"], np.matmul(tgt_pose, interp_inv).astype(np.float32), planes, K)
    torch.cuda.synchronize()
    assert m.network_status() == 0
    kern = _plan_kernels(m, batch, n, n, d)
    # (the conv-transposes take the 8-row tile where their grid is >= 3 tiles per CU: conv7_1 / conv8_1 at 8 faces, not conv6_1)
    assert [k[0] for k in kern] == (F32_BIG_GRID if batch >= 64 else f32_big_grid(DEFAULT_PLANES, wide_ct=(13, 15), wide_s2=(1, 3))), kern
    native = MSI(weights=weights, coord_net=True, input_type="PP")      # (the native-arithmetic plan still switches kernels with the batch)
    native.net_options[__import__("matryodshka_amd")._native.NET_OPT_F32_SPLIT3] = 0
    assert [k[0] for k in _plan_kernels(native, 2, n, n, d)][6] == "conv_igemm_kernel<64, 64, 0, 0>"
    assert [k[0] for k in _plan_kernels(native, batch, n, n, d)][6] == "conv_halo_s2_kernel<1>"
    got = dict(psv=net_input, rgba_layers=pred["rgba_layers"], rgb=rgb)
    rep = {k: _check_every_frame(z, k, got[k], bfix, TOL) for k in ("psv", "rgba_layers", "rgb")}
    print("config4 b=%d (max, mean, samples per group, faces):" % batch, rep)


def test_config1_plan_is_the_profiled_one():
    """configs[1] (batch 1): the plan of profiles/r04_*_config1_kernel_stats.txt -- fourteen 3x3 layers through the six-product
    bf16 split (conv_halo_x3_kernel / conv_halo_s2_x3_kernel) and the conv-transposes on convt_halo_x3_kernel, split tiles
    handed off inside the launch; with F32_SPLIT3 = 0 the r03 plan (conv3_3 on the tap kernel).  Parity at this shape:
    tests/test_golden.py::test_gpu_matches_full_size_samples, tests/test_gpu_split3.py."""
    from matryodshka_amd import _native as N, nets
    plan = N.NetPlan(nets.make_desc(1, 320, 640, 192, 64, 64, True, "f32"))
    k = [plan.layer_kernel(i) for i in range(18)]
    assert [x[0] for x in k[:17]] == F32_CONFIG1 and k[17][0] == "conv_igemm_kernel<64, 64, 2, 0>", k
    assert [x[2] for x in k[:17]] == [64, 32, 32, 32, 32, 32, 400, 400, 400, 400, 400, 32, 32, 32, 32, 32, 64], k   # (conv1_2 / conv8_1 on the 8-row tile: 800 tiles, 32 of them cut)
    plan.set_option(N.NET_OPT_F32_SPLIT3, 0)
    k = [plan.layer_kernel(i)[0] for i in range(17)]
    assert k[6] == "conv_igemm_kernel<64, 64, 0, 0>" and k[0] == "conv_halo_kernel<1, 0>" and k[1] == "conv_halo_s2_kernel<1>" and k[10] == "conv_igemm_kernel<64, 64, 1, 0>", k
