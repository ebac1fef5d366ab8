"""End-to-end GPU parity (infer_msi -> render, through the MSI class / C ABI) against the CPU
oracle on the shapes of the other BASELINE configs:
  * configs[2]-like: D = 64 spheres (Cin = 384, 128 outputs), batch > 1 (fp32 here; the bf16
    variant of that config is not built yet -- DESIGN.md section 8);
  * configs[3]-like: 1280x640 high_res input (reduced depth/width so the oracle finishes in seconds);
  * the non-CoordNet network (msi_train_net: wrap padding).
Tolerance 1e-3 max-abs on every float stage (north_star), uint8 within 1 LSB."""
import numpy as np
import pytest

from tests.util import make_inputs

pytestmark = pytest.mark.gpu
TOL = 1e-3


def _both(b, h, w, d, ngf, coord, seed):
    import torch
    from matryodshka_amd import MSI
    from oracle import nets as onets
    from oracle.msi import MSI as OracleMSI
    inp = make_inputs(seed, b, h, w)
    weights = onets.init_weights(6 * d, 2 * d, ngf=ngf, coord_net=coord, seed=seed, randomize_affine=True)
    m, o = MSI(weights=weights, coord_net=coord), OracleMSI(weights=weights, coord_net=coord)
    planes = m.inv_depths(1.0, 100.0, d)
    pred, net_input = m.infer_msi(torch.from_numpy(inp["src_image"]), torch.from_numpy(inp["ref_image"]), None, None,
                                  inp["ref_pose"], inp["src_pose"], inp["intrinsics"], "blend_psv", d, planes, ngf=ngf)
    rgb, dep = m.msi_render_equirect_view_and_depth(pred["rgba_layers"], inp["tgt_pose_rt"], inp["tgt_pos"], planes,
                                                    inp["intrinsics"])
    pred_o, net_input_o = o.infer_msi(inp["src_image"], inp["ref_image"], None, None, inp["ref_pose"], inp["src_pose"],
                                      inp["intrinsics"], "blend_psv", d, planes, ngf=ngf)
    rgb_o = o.msi_render_equirect_view(pred_o["rgba_layers"], inp["tgt_pose_rt"], inp["tgt_pos"], planes, inp["intrinsics"])
    dep_o = o.msi_render_equirect_depth(pred_o["rgba_layers"], inp["tgt_pose_rt"], inp["tgt_pos"], planes, inp["intrinsics"])
    errs = {
        "psv": np.abs(net_input.cpu().numpy() - net_input_o).max(),
        "rgba": np.abs(pred["rgba_layers"].cpu().numpy() - pred_o["rgba_layers"]).max(),
        "rgb": np.abs(rgb.cpu().numpy() - rgb_o).max(),
        "depth": np.abs(dep.cpu().numpy() - dep_o).max(),
        "rgb_u8": np.abs(m.deprocess_image(rgb).cpu().numpy().astype(int) - o.deprocess_image(rgb_o).astype(int)).max(),
    }
    return errs


@pytest.mark.parametrize("b,h,w,d,ngf,coord", [
    (2, 32, 64, 64, 16, True),     # configs[2] shape family: 64 spheres, batch > 1
    (1, 640, 1280, 4, 8, True),    # configs[3] resolution (high_res 1280x640)
    (2, 24, 48, 8, 16, False),     # msi_train_net (no CoordNet): wrap padding
    (3, 16, 40, 16, 12, True),     # odd channel counts: channel-tail and M-tail paths
])
def test_pipeline_matches_oracle(b, h, w, d, ngf, coord):
    errs = _both(b, h, w, d, ngf, coord, seed=100 + d)
    for k in ("psv", "rgba", "rgb", "depth"):
        assert errs[k] <= TOL, (k, errs)
    assert errs["rgb_u8"] <= 1, errs


def test_batch_elements_are_independent():
    """B frames are B independent B=1 evaluations: frame i of a batch == the same frame alone."""
    import torch
    from matryodshka_amd import MSI
    from oracle import nets as onets
    b, h, w, d, ngf = 3, 16, 32, 8, 16
    inp = make_inputs(77, b, h, w)
    weights = onets.init_weights(6 * d, 2 * d, ngf=ngf, coord_net=True, seed=5)
    m = MSI(weights=weights, coord_net=True)
    planes = m.inv_depths(1.0, 100.0, d)

    def run(sl):
        pred, _ = m.infer_msi(torch.from_numpy(inp["src_image"][sl]), torch.from_numpy(inp["ref_image"][sl]), None, None,
                              inp["ref_pose"][sl], inp["src_pose"][sl], inp["intrinsics"][sl], "blend_psv", d, planes, ngf=ngf)
        return m.msi_render_equirect_view(pred["rgba_layers"], inp["tgt_pose_rt"][sl], inp["tgt_pos"][sl], planes,
                                          inp["intrinsics"][sl]).clone()
    full = run(slice(0, b))
    for i in range(b):
        one = run(slice(i, i + 1))
        assert torch.equal(full[i:i + 1], one), i     # bit-identical: no cross-sample reduction anywhere


def test_high_res_rerender_matches_plane_by_plane_oracle():
    """test.py:283-394 (BASELINE configs[3] 'high_res'): low-res inference, then re-render at a
    higher resolution from upsampled blend weights / alphas.  The oracle follows the reference's
    per-plane host loop; the HIP path does it in one fused pass."""
    import torch
    from matryodshka_amd import MSI
    from oracle import nets as onets
    from oracle.msi import MSI as OracleMSI
    b, h, w, d, ngf = 1, 16, 32, 8, 16
    hh, hw = 40, 88                                     # non-integer scale on purpose
    inp = make_inputs(31, b, h, w)
    hres = make_inputs(32, b, hh, hw)
    weights = onets.init_weights(6 * d, 2 * d, ngf=ngf, coord_net=True, seed=31, randomize_affine=True)
    m, o = MSI(weights=weights, coord_net=True), OracleMSI(weights=weights, coord_net=True)
    planes = m.inv_depths(1.0, 100.0, d)
    pred, _ = m.infer_msi(torch.from_numpy(inp["src_image"]), torch.from_numpy(inp["ref_image"]), None, None,
                          inp["ref_pose"], inp["src_pose"], inp["intrinsics"], "blend_psv", d, planes,
                          extra_outputs="blend_weights alphas", ngf=ngf)
    rgb, dep = m.msi_render_equirect_hres(pred["blend_weights"], pred["alphas"], torch.from_numpy(hres["ref_image"]),
                                          torch.from_numpy(hres["src_image"]), inp["ref_pose"], inp["src_pose"],
                                          inp["tgt_pose_rt"], inp["tgt_pos"], planes, inp["intrinsics"])
    pred_o, _ = o.infer_msi(inp["src_image"], inp["ref_image"], None, None, inp["ref_pose"], inp["src_pose"],
                            inp["intrinsics"], "blend_psv", d, planes, extra_outputs="blend_weights alphas", ngf=ngf)
    rgb_o, dep_o = o.render_hres(pred_o["blend_weights"], pred_o["alphas"], hres["ref_image"], hres["src_image"],
                                 inp["ref_pose"], inp["src_pose"], inp["tgt_pose_rt"], inp["tgt_pos"], planes,
                                 inp["intrinsics"])
    assert tuple(rgb.shape) == rgb_o.shape == (b, hh, hw, 3)
    assert np.abs(rgb.cpu().numpy() - rgb_o).max() <= TOL
    assert np.abs(dep.cpu().numpy() - dep_o).max() <= TOL


def _pp_inputs(seed, b, n):
    """data_loader.py:205-226 (input_type PP): fx = cx = W/2, fy = cy = H/2; source shifted along -x
    by the input offset, target by the target offset; 256x256 cube faces in configs[4]."""
    rng = np.random.RandomState(seed)
    from tests.util import smooth_noise
    ref = smooth_noise(rng, b, n, n); src = smooth_noise(rng, b, n, n)
    K = np.tile(np.array([[n / 2, 0, n / 2], [0, n / 2, n / 2], [0, 0, 1]], np.float32)[None], (b, 1, 1))
    eye = np.tile(np.eye(4, dtype=np.float32)[None], (b, 1, 1))
    src_pose = eye.copy(); src_pose[:, 0, 3] = -0.064
    tgt_pose = eye.copy(); tgt_pose[:, 0, 3] = -0.03; tgt_pose[:, 1, 3] = 0.01
    th = 0.02
    tgt_pose[:, 0, 0] = np.cos(th); tgt_pose[:, 0, 2] = np.sin(th); tgt_pose[:, 2, 0] = -np.sin(th); tgt_pose[:, 2, 2] = np.cos(th)
    return ref, src, K, eye, src_pose, tgt_pose


def test_pp_cube_face_path_matches_oracle():
    """BASELINE configs[4] (input_type=PP): perspective plane sweep -> CNN -> assemble -> homography
    (MPI) render, per cube face; here 2 faces of 32x32 with 8 planes against the oracle."""
    import torch
    from matryodshka_amd import MSI
    from oracle import nets as onets
    from oracle.msi import MSI as OracleMSI
    b, n, d, ngf = 2, 32, 8, 16
    ref, src, K, eye, src_pose, tgt_pose = _pp_inputs(5, b, n)
    weights = onets.init_weights(6 * d, 2 * d, ngf=ngf, coord_net=True, seed=3, randomize_affine=True)
    m = MSI(weights=weights, coord_net=True, input_type='PP')
    o = OracleMSI(weights=weights, coord_net=True, input_type='PP')
    planes = m.inv_depths(1.0, 100.0, d)
    pred, net_input = m.infer_msi(torch.from_numpy(src), torch.from_numpy(ref), None, None, eye, src_pose, K,
                                  "blend_psv", d, planes, ngf=ngf)
    out = m.mpi_render_view(pred["rgba_layers"], tgt_pose, planes, K)
    pred_o, net_input_o = o.infer_msi(src, ref, None, None, eye, src_pose, K, "blend_psv", d, planes, ngf=ngf)
    out_o = o.mpi_render_view(pred_o["rgba_layers"], tgt_pose, planes, K)
    assert np.abs(net_input.cpu().numpy() - net_input_o).max() <= TOL
    assert np.abs(pred["rgba_layers"].cpu().numpy() - pred_o["rgba_layers"]).max() <= TOL
    assert np.abs(out.cpu().numpy() - out_o).max() <= TOL


def test_mpi_render_identity_and_zero_padding():
    """KAT for tf.contrib.resampler semantics: identity pose reproduces the
    over-composite exactly; a pure x-shift brings in zeros (not wrapped texels) at the border."""
    import torch
    from matryodshka_amd import MSI
    from tests.util import random_rgba
    b, n, d = 1, 16, 3
    rgba = random_rgba(8, b, n, n, d)
    m = MSI()
    K = np.array([[[n / 2, 0, n / 2], [0, n / 2, n / 2], [0, 0, 1]]], np.float32)
    planes = [4.0, 2.0, 1.0]
    eye = np.eye(4, dtype=np.float32)[None]
    out = m.mpi_render_view(torch.from_numpy(rgba).cuda(), eye, planes, K).cpu().numpy()
    exp = rgba[..., 0, :3]
    for i in range(1, d):
        a = rgba[..., i, 3:]
        exp = rgba[..., i, :3] * a + exp * (1 - a)
    assert np.abs(out - exp).max() < 1e-5
    # x-translation of 0.5 at depth 1 with fx = n/2 shifts by exactly 4 px: four border columns
    # must come out as exact zeros (zero padding), NOT as wrapped texels
    from oracle.msi import MSI as OracleMSI
    shift = eye.copy(); shift[0, 0, 3] = 0.5
    one = rgba[..., :1, :].copy(); one[..., 3] = 1.0
    got = m.mpi_render_view(torch.from_numpy(one).cuda(), shift, [1.0], K).cpu().numpy()
    ref = OracleMSI(input_type='PP').mpi_render_view(one, shift, [1.0], K)
    assert np.abs(got - ref).max() < 1e-5
    zero_cols = [j for j in range(n) if not got[0, :, j].any()]
    assert len(zero_cols) == 4 and (zero_cols == [0, 1, 2, 3] or zero_cols == [n - 4, n - 3, n - 2, n - 1])


def test_hres_rerender_on_a_bf16_model_uses_an_fp32_volume():
    """msi_render_equirect_hres on a dtype='bf16' model used to hand a bf16 high-res volume to the
    fp32 assembly.  The high-res volume is fp32 whatever the model's dtype: same result as on an fp32 model."""
    import torch
    from matryodshka_amd import MSI
    b, h, w, d = 1, 16, 32, 8
    hh, hw = 32, 64
    inp = make_inputs(41, b, h, w)
    hres = make_inputs(42, b, hh, hw)
    rng = np.random.RandomState(0)
    bw = torch.from_numpy(rng.uniform(0, 1, (b, h, w, d)).astype(np.float32)).cuda()
    al = torch.from_numpy(rng.uniform(0, 1, (b, h, w, d)).astype(np.float32)).cuda()
    outs = []
    for dtype in ("f32", "bf16"):
        m = MSI(coord_net=True, dtype=dtype)
        planes = m.inv_depths(1.0, 100.0, d)
        outs.append(m.msi_render_equirect_hres(bw, al, torch.from_numpy(hres["ref_image"]), torch.from_numpy(hres["src_image"]),
                                               inp["ref_pose"], inp["src_pose"], inp["tgt_pose_rt"], inp["tgt_pos"], planes,
                                               inp["intrinsics"]))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert bool(torch.isfinite(outs[1][0]).all())


@pytest.mark.parametrize("scheme,coord", [("blend_bg", True), ("blend_bg_psv", False), ("alpha_only", True)])
def test_infer_msi_colour_schemes_match_oracle(scheme, coord):
    """FLAGS.which_color_pred (test.py:55-56, msi.py:166-275) end to end: network with 2D+3 / 3D+3 / D outputs,
    layer assembly, render."""
    import torch
    from matryodshka_amd import MSI
    from oracle import nets as onets
    from oracle.msi import MSI as OracleMSI
    b, h, w, d, ngf = 2, 16, 40, 8, 16
    nout = {"blend_bg": 2 * d + 3, "blend_bg_psv": 3 * d + 3, "alpha_only": d}[scheme]
    inp = make_inputs(51, b, h, w)
    weights = onets.init_weights(6 * d, nout, ngf=ngf, coord_net=coord, seed=9, randomize_affine=True)
    m, o = MSI(weights=weights, coord_net=coord), OracleMSI(weights=weights, coord_net=coord)
    planes = m.inv_depths(1.0, 100.0, d)
    pred, _ = m.infer_msi(torch.from_numpy(inp["src_image"]), torch.from_numpy(inp["ref_image"]), None, None,
                          inp["ref_pose"], inp["src_pose"], inp["intrinsics"], scheme, d, planes,
                          extra_outputs="blend_weights alphas", ngf=ngf)
    pred_o, _ = o.infer_msi(inp["src_image"], inp["ref_image"], None, None, inp["ref_pose"], inp["src_pose"],
                            inp["intrinsics"], scheme, d, planes, extra_outputs="blend_weights alphas", ngf=ngf)
    assert set(pred) == set(pred_o)
    for k in pred_o:
        assert np.abs(pred[k].cpu().numpy() - pred_o[k]).max() <= TOL, k
    rgb = m.msi_render_equirect_view(pred["rgba_layers"], inp["tgt_pose_rt"], inp["tgt_pos"], planes, inp["intrinsics"])
    rgb_o = o.msi_render_equirect_view(pred_o["rgba_layers"], inp["tgt_pose_rt"], inp["tgt_pos"], planes, inp["intrinsics"])
    assert np.abs(rgb.cpu().numpy() - rgb_o).max() <= TOL


def test_bench_gpus_2_end_to_end_on_one_gpu():
    """`python bench.py --gpus 2 --steps 3` as the driver types it (no launcher): bench.py re-runs
    itself as two ranks; on this one-GPU box they share the device over gloo (RCCL refuses two ranks on one device).
    The line must be the contract's, from a 2-rank process group, with one frame per rank per step."""
    from tests.test_dist_cpu import _run_bench
    j = _run_bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--prewarm", "0.2", "--repeats", "0", "--no-settle",
                    "--strong-frames", "2"], timeout=900)
    assert j["n_gpus"] == 2 and j["distributed"]["world_size_process_group"] == 2 and j["distributed"]["backend"] == "gloo"
    assert j["distributed"]["frame_ranges_per_rank"] == [[0, 1], [1, 2]] and j["config"]["frames_per_step"] == 2
    assert j["steps"] == 3 and j["value"] > 0 and abs(j["value"] - 2 * 1e3 / j["ms_per_step"]) <= 0.01 * j["value"]
    assert j["roofline"]["frac"] > 0 and j["cpu_baseline"] is None and j["scaling"] == "weak"
    j = _run_bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--prewarm", "0.2", "--repeats", "0", "--no-settle",
                    "--config", "3"], timeout=900)
    assert j["distributed"]["frame_ranges_per_rank"] == [[0, 16], [16, 32]] and j["scaling"] == "strong"
