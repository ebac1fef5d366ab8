"""The test.py-equivalent harness (matryodshka_amd/harness.py): a synthetic 'Replica-style' sample
(camera .txt line + three .jpeg files at 2x the working resolution) must produce the reference's
output files (test.py:209-281), and the written PNGs must equal the direct API results."""
import os

import numpy as np
import pytest

from tests.util import make_inputs

pytestmark = pytest.mark.gpu


def test_harness_writes_reference_outputs(tmp_path):
    from PIL import Image
    from matryodshka_amd import harness
    h, w, d, ngf = 16, 32, 4, 8
    inp = make_inputs(3, 1, 2 * h, 2 * w)
    img_dir = tmp_path / "images"; img_dir.mkdir()
    rng = np.random.RandomState(0)
    for name in ("000", "001", "002"):
        arr = np.clip(rng.uniform(0, 255, size=(2 * h, 2 * w, 3)), 0, 255).astype(np.uint8)
        Image.fromarray(arr).save(str(img_dir / ("room_0_pos%s.jpeg" % name)), quality=95)
    cam = tmp_path / "cams.txt"
    cam.write_text("room_0 000 001 002 0.032 0.01 -0.02 0.03\n")
    out_root = tmp_path / "out"
    n = harness.main(["--cameras_glob", str(cam), "--image_dir", str(img_dir), "--output_root", str(out_root),
                      "--experiment_name", "exp", "--height", str(h), "--width", str(w), "--num_msi_planes", str(d),
                      "--num_psv_planes", str(d), "--ngf", str(ngf),
                      "--test_outputs", "src_image_ref_image_tgt_image_psv_rgba_layers_blend_weights_alphas"])
    assert n == 1
    sample = out_root / "exp" / "room_0_000001002"
    expected = ["tgt_image_room_0_000001002.png", "output_tgt_room_0_000001002.png", "output_depth_room_0_000001002.png",
                "src_image_room_0_000001002.png", "ref_image_room_0_000001002.png", "blend_weights.npy", "alphas.npy"]
    expected += ["psv_plane_%.3d.png" % j for j in range(d)] + ["blend_weight_%.3d.png" % j for j in range(d)]
    expected += ["msi_alpha_%.2d.png" % j for j in range(d)] + ["msi_rgb_%.2d.png" % j for j in range(d)]
    for f in expected:
        assert (sample / f).exists(), f
    assert (out_root / "exp" / "step.txt").read_text() == "0"
    bw = np.load(str(sample / "blend_weights.npy"))
    assert bw.shape == (1, h, w, d) and 0.0 <= bw.min() and bw.max() <= 1.0
    out = np.asarray(Image.open(str(sample / "output_tgt_room_0_000001002.png")))
    assert out.shape == (h, w, 3) and out.dtype == np.uint8
    # the area resize of the harness: ref_image png == box mean of the 2x jpeg
    ref_png = np.asarray(Image.open(str(sample / "ref_image_room_0_000001002.png"))).astype(int)
    jpg = np.asarray(Image.open(str(img_dir / "room_0_pos000.jpeg")).convert("RGB"), dtype=np.float32)
    box = jpg.reshape(h, 2, w, 2, 3).mean(axis=(1, 3))
    assert np.abs(ref_png - np.clip(box, 0, 255).astype("uint8").astype(int)).max() <= 1


def test_harness_restores_a_tf_checkpoint(tmp_path):
    """--checkpoint: the variables come from a TF V2 checkpoint (written here by tf_checkpoint.write_checkpoint),
    step.txt records its global step (test.py:225-229), and the outputs equal a run on the same weights via --weights."""
    from PIL import Image
    from matryodshka_amd import harness, nets, tf_checkpoint
    h, w, d, ngf = 16, 32, 4, 8
    img_dir = tmp_path / "images"; img_dir.mkdir()
    rng = np.random.RandomState(1)
    for name in ("000", "001", "002"):
        Image.fromarray(rng.randint(0, 255, size=(h, w, 3)).astype(np.uint8)).save(str(img_dir / ("s_pos%s.jpeg" % name)), quality=95)
    cam = tmp_path / "cams.txt"
    cam.write_text("s 000 001 002 0.032 0.01 -0.02 0.03\n")
    weights = nets.init_weights(6 * d, 2 * d, ngf, True, seed=5)
    ckpt_dir = tmp_path / "ckpt"; ckpt_dir.mkdir()
    tensors = {"net/" + k: v for k, v in weights.items()}
    tensors["global_step"] = np.array(1234, np.int64)
    tf_checkpoint.write_checkpoint(str(ckpt_dir / "model.ckpt-1234"), tensors)
    np.savez(str(tmp_path / "w.npz"), **weights)
    common = ["--cameras_glob", str(cam), "--image_dir", str(img_dir), "--height", str(h), "--width", str(w),
              "--num_msi_planes", str(d), "--num_psv_planes", str(d), "--ngf", str(ngf), "--test_outputs", "tgt_image_alphas"]
    assert harness.main(common + ["--output_root", str(tmp_path / "a"), "--experiment_name", "e", "--checkpoint", str(ckpt_dir)]) == 1
    assert harness.main(common + ["--output_root", str(tmp_path / "b"), "--experiment_name", "e", "--weights", str(tmp_path / "w.npz")]) == 1
    assert (tmp_path / "a" / "e" / "step.txt").read_text() == "1234"
    fa = tmp_path / "a" / "e" / "s_000001002"
    fb = tmp_path / "b" / "e" / "s_000001002"
    assert np.array_equal(np.load(str(fa / "alphas.npy")), np.load(str(fb / "alphas.npy")))
    assert np.array_equal(np.asarray(Image.open(str(fa / "output_tgt_s_000001002.png"))),
                          np.asarray(Image.open(str(fb / "output_tgt_s_000001002.png"))))


@pytest.mark.parametrize("coord,scheme", [(True, "blend_psv"), (False, "blend_psv"), (True, "blend_bg_psv")])
def test_harness_outputs_equal_the_oracle(tmp_path, coord, scheme):
    """SURVEY 8f-2 / test.py:231-281: the written PNGs / NPYs against the CPU oracle run on the same decoded and
    area-resized images: uint8 outputs equal up to 1 LSB on < 0.1 % of the pixels (DESIGN.md tolerances); the network
    variant is inferred from the weights (CoordNet has one more input channel)."""
    from PIL import Image
    from matryodshka_amd import harness
    from oracle import nets as onets
    from oracle.msi import MSI as OracleMSI
    h, w, d, ngf = 32, 64, 8, 16
    nout = {"blend_psv": 2 * d, "blend_bg_psv": 3 * d + 3}[scheme]
    img_dir = tmp_path / "images"; img_dir.mkdir()
    from tests.util import smooth_noise
    rng = np.random.RandomState(5)
    for name in ("000", "001", "002"):
        arr = (smooth_noise(rng, 1, 2 * h, 2 * w)[0] * 255).astype(np.uint8)
        Image.fromarray(arr).save(str(img_dir / ("apt_pos%s.jpeg" % name)), quality=95)
    cam = tmp_path / "cams.txt"
    cam.write_text("apt 000 001 002 0.032 0.02 -0.01 0.03\n")
    weights = onets.init_weights(6 * d, nout, ngf=ngf, coord_net=coord, seed=13, randomize_affine=True)
    np.savez(str(tmp_path / "w.npz"), **weights)
    outputs = "src_image_ref_image_tgt_image_psv_rgba_layers_blend_weights_alphas_src_output_image_ref_output_image_psp"
    assert harness.main(["--cameras_glob", str(cam), "--image_dir", str(img_dir), "--output_root", str(tmp_path / "o"),
                         "--experiment_name", "e", "--height", str(h), "--width", str(w), "--num_msi_planes", str(d),
                         "--num_psv_planes", str(d), "--ngf", str(ngf), "--weights", str(tmp_path / "w.npz"), "--which_color_pred", scheme,
                         "--test_outputs", outputs]) == 1
    sample = tmp_path / "o" / "e" / "apt_000001002"
    ref, src = (harness.load_image(str(img_dir / ("apt_pos%s.jpeg" % n)), h, w)[None] for n in ("000", "001"))
    o = OracleMSI(weights=weights, coord_net=coord)
    planes = o.inv_depths(1.0, 100.0, d)
    eye = np.eye(4, dtype=np.float32)[None]
    intr = np.array([[[0.032, 0, 0], [0, 1, 0], [0, 0, 1]]], np.float32)
    pos = np.array([[0.02, -0.01, 0.03]], np.float32)
    pred, psv = o.infer_msi(src, ref, None, None, eye, eye, intr, scheme, d, planes, extra_outputs="blend_weights alphas", ngf=ngf)

    def png(name):
        return np.asarray(Image.open(str(sample / name))).astype(int)

    def close(name, want_u8):
        got = png(name)
        diff = np.abs(got - want_u8.astype(int))
        assert diff.max() <= 1 and (diff > 0).mean() < 1e-3, (name, diff.max(), (diff > 0).mean())

    tag = "apt_000001002"
    close("output_tgt_%s.png" % tag, o.deprocess_image(o.msi_render_equirect_view(pred["rgba_layers"], eye, pos, planes, intr))[0])
    close("output_depth_%s.png" % tag, o.deprocess_depth_image(o.msi_render_equirect_depth(pred["rgba_layers"], eye, pos, planes, intr))[0])
    close("output_src_%s.png" % tag, o.deprocess_image(o.msi_render_ods_view(pred["rgba_layers"], -1, eye, pos, planes, intr))[0])
    close("output_ref_%s.png" % tag, o.deprocess_image(o.msi_render_ods_view(pred["rgba_layers"], 1, eye, pos, planes, intr))[0])
    for vw in range(4):
        want = o.deprocess_image(o.msi_render_perspective_view(pred["rgba_layers"], eye, pos, planes, intr, viewing_window=vw))[0]
        close("output_ptgt%d_%s.png" % (vw, tag), want)
    u8 = lambda x: np.clip(x, 0, 255).astype("uint8")          # utils.write_image (utils.py:76-81)
    for i in (0, d // 2, d - 1):
        close("msi_alpha_%.2d.png" % i, u8(pred["rgba_layers"][0, :, :, i, 3] * 255.0))
        close("msi_rgb_%.2d.png" % i, u8((pred["rgba_layers"][0, :, :, i, :3] + 1.) / 2. * 255))
        close("psv_plane_%.3d.png" % i, u8((psv[0, :, :, i * 3:(i + 1) * 3] + 1.) / 2. * 255))
        close("blend_weight_%.3d.png" % i, u8(pred["blend_weights"][0, :, :, i] * 255.0))
    assert np.abs(np.load(str(sample / "blend_weights.npy")) - pred["blend_weights"]).max() <= 1e-3
    assert np.abs(np.load(str(sample / "alphas.npy")) - pred["alphas"]).max() <= 1e-3


def _write_sample_images(img_dir, scene, h, w, seed, names=("000", "001", "002")):
    from PIL import Image
    from tests.util import smooth_noise
    img_dir.mkdir(exist_ok=True)
    rng = np.random.RandomState(seed)
    for name in names:
        arr = (smooth_noise(rng, 1, h, w)[0] * 255).astype(np.uint8)
        Image.fromarray(arr).save(str(img_dir / ("%s_pos%s.jpeg" % (scene, name))), quality=95)


def _close_png(path, want_u8):
    from PIL import Image
    got = np.asarray(Image.open(str(path))).astype(int)
    diff = np.abs(got - np.asarray(want_u8).astype(int))
    assert diff.max() <= 1 and (diff > 0).mean() < 1e-3, (str(path), diff.max(), (diff > 0).mean())


def test_harness_high_res_mode_equals_the_oracle(tmp_path):
    """--test_type high_res (test.py:283-394): the low-res pass saves blend_weights.npy / alphas.npy, the high-res pass
    re-assembles the layers from the high-res sweep volume and writes output_hrestgt_* / output_hresdepth_* (test.py:383-394:
    (x + 1) / 2 * 255 and depth * 255 through write_image) -- against the oracle's per-plane loop; then
    high_res_only re-renders from the saved files alone, and on_video + prefix names the directories (test.py:209-217)."""
    from matryodshka_amd import harness
    from oracle import nets as onets
    from oracle.msi import MSI as OracleMSI
    h, w, hh, hw, d, ngf = 32, 64, 64, 128, 8, 16
    _write_sample_images(tmp_path / "lo", "office_0", h, w, 7)
    _write_sample_images(tmp_path / "hi", "office_0", 2 * hh, 2 * hw, 8)            # area-resized 2x down by the loader
    cam = tmp_path / "cams.txt"
    cam.write_text("office_0 000 001 002 0.032 0.02 -0.01 0.03\n")
    weights = onets.init_weights(6 * d, 2 * d, ngf=ngf, coord_net=True, seed=17, randomize_affine=True)
    np.savez(str(tmp_path / "w.npz"), **weights)
    common = ["--cameras_glob", str(cam), "--image_dir", str(tmp_path / "lo"), "--hres_image_dir", str(tmp_path / "hi"),
              "--output_root", str(tmp_path / "o"), "--experiment_name", "e", "--height", str(h), "--width", str(w),
              "--hres_height", str(hh), "--hres_width", str(hw), "--num_msi_planes", str(d), "--num_psv_planes", str(d),
              "--ngf", str(ngf), "--weights", str(tmp_path / "w.npz")]
    assert harness.main(common + ["--test_type", "on_video_high_res", "--prefix", "supp"]) == 1
    tag = "video_supp_office_0_000001002"
    sample = tmp_path / "o" / "e" / tag
    assert (sample / ("output_tgt_%s.png" % tag)).exists() and (sample / "blend_weights.npy").exists()
    ref, src = (harness.load_image(str(tmp_path / "lo" / ("office_0_pos%s.jpeg" % n)), h, w)[None] for n in ("000", "001"))
    href, hsrc = (harness.load_image(str(tmp_path / "hi" / ("office_0_pos%s.jpeg" % n)), hh, hw)[None] for n in ("000", "001"))
    o = OracleMSI(weights=weights, coord_net=True)
    planes = o.inv_depths(1.0, 100.0, d)
    eye = np.eye(4, dtype=np.float32)[None]
    intr = np.array([[[0.032, 0, 0], [0, 1, 0], [0, 0, 1]]], np.float32)
    pos = np.array([[0.02, -0.01, 0.03]], np.float32)
    pred, _ = o.infer_msi(src, ref, None, None, eye, eye, intr, "blend_psv", d, planes, extra_outputs="blend_weights alphas", ngf=ngf)
    hrgb, hdep = o.render_hres(pred["blend_weights"], pred["alphas"], href, hsrc, eye, eye, eye, pos, planes, intr)
    u8 = lambda x: np.clip(x, 0, 255).astype("uint8")                                # utils.write_image (utils.py:76-81)
    _close_png(sample / ("output_hrestgt_%s.png" % tag), u8(((hrgb[0] + 1.) / 2.) * 255.))
    _close_png(sample / ("output_hresdepth_%s.png" % tag), u8(hdep[0] * 255.))
    # high_res_only: nothing but the re-render, from the saved .npy files
    before = (sample / ("output_hrestgt_%s.png" % tag)).read_bytes()
    os.remove(str(sample / ("output_hrestgt_%s.png" % tag)))
    low_mtime = os.stat(str(sample / ("output_tgt_%s.png" % tag))).st_mtime_ns
    assert harness.main(common + ["--test_type", "on_video_high_res_only", "--prefix", "supp"]) == 1
    assert (sample / ("output_hrestgt_%s.png" % tag)).read_bytes() == before
    assert os.stat(str(sample / ("output_tgt_%s.png" % tag))).st_mtime_ns == low_mtime      # the low-res pass did not run again
    # evaluate.py --video now finds what the harness wrote
    from matryodshka_amd import evaluate
    cam.write_text("office_0 000 001 002 0.032 0.02 -0.01 0.03\noffice_0 001 002 000 0.032 0.0 0.01 0.02\n")
    assert harness.main(common + ["--test_type", "on_video", "--test_outputs", "tgt_image"]) == 2
    table = evaluate.main(["--result_root", str(tmp_path / "o"), "--model_names", "e", "--output_table", str(tmp_path / "t.json"),
                           "--video", "--videos", "office_0"])
    assert table["examples"] == [] and len(table["consecutive"][0]) == 2          # three video frames of one scene: two pairs


def test_harness_pp_mode_equals_the_oracle(tmp_path):
    """--input_type PP (test.py:51, data_loader.py:205-226): perspective camera lines `scene ref src tgt input_offset tgt_offset`,
    plane sweep at the slerp mid-point pose, mpi_render_view through tgt_pose @ interp_pose_inv (msi.py:644-646)."""
    from matryodshka_amd import harness
    from oracle import nets as onets, poses as oposes
    from oracle.msi import MSI as OracleMSI
    n, d, ngf = 64, 8, 16
    _write_sample_images(tmp_path / "img", "room_1", n, n, 9)
    cam = tmp_path / "cams.txt"
    cam.write_text("room_1 000 001 002 0.064 0.03\n")
    weights = onets.init_weights(6 * d, 2 * d, ngf=ngf, coord_net=True, seed=19, randomize_affine=True)
    np.savez(str(tmp_path / "w.npz"), **weights)
    assert harness.main(["--cameras_glob", str(cam), "--image_dir", str(tmp_path / "img"), "--output_root", str(tmp_path / "o"),
                         "--experiment_name", "e", "--height", str(n), "--width", str(n), "--num_msi_planes", str(d),
                         "--num_psv_planes", str(d), "--ngf", str(ngf), "--weights", str(tmp_path / "w.npz"),
                         "--input_type", "PP", "--test_outputs", "tgt_image_rgba_layers_alphas"]) == 1
    tag = "room_1_000001002"
    sample = tmp_path / "o" / "e" / tag
    ref, src = (harness.load_image(str(tmp_path / "img" / ("room_1_pos%s.jpeg" % k)), n, n)[None] for k in ("000", "001"))
    o = OracleMSI(weights=weights, coord_net=True, input_type="PP")
    planes = o.inv_depths(1.0, 100.0, d)
    eye = np.eye(4, dtype=np.float32)[None]
    src_pose, tgt_pose = eye.copy(), eye.copy()
    src_pose[0, 0, 3], tgt_pose[0, 0, 3] = -0.064, -0.03
    K = np.array([[[n / 2, 0, n / 2], [0, n / 2, n / 2], [0, 0, 1]]], np.float32)
    interp_inv = np.linalg.inv(oposes.interpolate_pose(eye, src_pose).astype(np.float64)).astype(np.float32)
    pred, _ = o.infer_msi(src, ref, None, None, eye, src_pose, K, "blend_psv", d, planes, extra_outputs="alphas", ngf=ngf,
                          ref_pose_inv=interp_inv)
    rgb = o.mpi_render_view(pred["rgba_layers"], np.matmul(tgt_pose, interp_inv).astype(np.float32), planes, K)
    _close_png(sample / ("output_tgt_%s.png" % tag), o.deprocess_image(rgb)[0])
    u8 = lambda x: np.clip(x, 0, 255).astype("uint8")
    for i in (0, d - 1):
        _close_png(sample / ("msi_alpha_%.2d.png" % i), u8(pred["rgba_layers"][0, :, :, i, 3] * 255.0))
    assert np.abs(np.load(str(sample / "alphas.npy")) - pred["alphas"]).max() <= 1e-3
    assert not (sample / ("output_depth_%s.png" % tag)).exists()                    # the MPI path has no depth render


def test_harness_restores_a_hand_assembled_bundle_and_matches_the_oracle(tmp_path):
    """SURVEY 8f-1 / test.py:191-202: the variables of a real (small) network in a TF V2 bundle assembled BYTE BY BYTE by the
    test (two data shards, Adam slots, one conv weight partitioned along its first axis) -- not written by
    tf_checkpoint.write_checkpoint -- restored by `harness --checkpoint`, and the written outputs compared with the ORACLE
    run on the same arrays (not with another run of the product)."""
    from matryodshka_amd import harness
    from oracle import nets as onets
    from oracle.msi import MSI as OracleMSI
    from tests.test_tf_checkpoint import _assemble_fixture
    h, w, d, ngf = 32, 64, 8, 16
    _write_sample_images(tmp_path / "img", "hotel_0", h, w, 11)
    cam = tmp_path / "cams.txt"
    cam.write_text("hotel_0 000 001 002 0.032 0.01 0.02 -0.03\n")
    weights = onets.init_weights(6 * d, 2 * d, ngf=ngf, coord_net=True, seed=23, randomize_affine=True)
    tensors = {"net/" + k: v for k, v in weights.items() if k != "conv2_1/weights"}
    for k in ("conv1_1/weights", "conv4_2/weights", "color_pred/biases"):           # optimizer slots a trained bundle carries
        tensors["net/" + k + "/Adam"] = np.zeros_like(weights[k])
        tensors["net/" + k + "/Adam_1"] = np.ones_like(weights[k])
    tensors["beta1_power"] = np.array(0.9 ** 7, np.float32)
    tensors["beta2_power"] = np.array(0.999 ** 7, np.float32)
    tensors["global_step"] = np.array(400000, np.int64)
    ckpt = tmp_path / "ckpt"; ckpt.mkdir()
    prefix, _ = _assemble_fixture(ckpt, tensors=tensors, part=weights["conv2_1/weights"], part_name="net/conv2_1/weights", split=2)
    assert harness.main(["--cameras_glob", str(cam), "--image_dir", str(tmp_path / "img"), "--output_root", str(tmp_path / "o"),
                         "--experiment_name", "e", "--height", str(h), "--width", str(w), "--num_msi_planes", str(d),
                         "--num_psv_planes", str(d), "--ngf", str(ngf), "--checkpoint", str(ckpt),
                         "--test_outputs", "tgt_image_alphas_blend_weights"]) == 1
    assert (tmp_path / "o" / "e" / "step.txt").read_text() == "400000"
    tag = "hotel_0_000001002"
    sample = tmp_path / "o" / "e" / tag
    ref, src = (harness.load_image(str(tmp_path / "img" / ("hotel_0_pos%s.jpeg" % k)), h, w)[None] for k in ("000", "001"))
    o = OracleMSI(weights=weights, coord_net=True)
    planes = o.inv_depths(1.0, 100.0, d)
    eye = np.eye(4, dtype=np.float32)[None]
    intr = np.array([[[0.032, 0, 0], [0, 1, 0], [0, 0, 1]]], np.float32)
    pos = np.array([[0.01, 0.02, -0.03]], np.float32)
    pred, _ = o.infer_msi(src, ref, None, None, eye, eye, intr, "blend_psv", d, planes, extra_outputs="blend_weights alphas", ngf=ngf)
    _close_png(sample / ("output_tgt_%s.png" % tag), o.deprocess_image(o.msi_render_equirect_view(pred["rgba_layers"], eye, pos, planes, intr))[0])
    _close_png(sample / ("output_depth_%s.png" % tag), o.deprocess_depth_image(o.msi_render_equirect_depth(pred["rgba_layers"], eye, pos, planes, intr))[0])
    assert np.abs(np.load(str(sample / "alphas.npy")) - pred["alphas"]).max() <= 1e-3
    assert np.abs(np.load(str(sample / "blend_weights.npy")) - pred["blend_weights"]).max() <= 1e-3


@pytest.mark.parametrize("coord", [False, True])
def test_harness_at_baseline_config0_size_equals_the_oracle(tmp_path, coord):
    """BASELINE configs[0] at ITS OWN workload: test.py:87-281 on one 640x320 ODS pair, 32 spheres,
    ngf 64 -- 1280x640 JPEGs on disk -> area resize -> HIP path -> the reference's PNG / NPY files, compared with the CPU
    oracle on the same decoded + resized images (<= 1 LSB on < 0.1 % of the pixels; blend_weights.npy / alphas.npy 1e-3).
    coord = False is test.py:52's default network (msi_train_net), True the released ODS models' (msi_coord_train_net)."""
    import torch
    from PIL import Image
    from matryodshka_amd import harness
    from oracle import nets as onets
    from oracle.msi import MSI as OracleMSI
    h, w, d, ngf = 320, 640, 32, 64
    _write_sample_images(tmp_path / "test_640x320", "apartment_0", 2 * h, 2 * w, 21)
    cam = tmp_path / "cams.txt"
    cam.write_text("apartment_0 000 001 002 0.032 0.02 -0.01 0.03\n")
    weights = onets.init_weights(6 * d, 2 * d, ngf=ngf, coord_net=coord, seed=22, randomize_affine=True)
    np.savez(str(tmp_path / "w.npz"), **weights)
    args = ["--cameras_glob", str(cam), "--image_dir", str(tmp_path / "test_640x320"), "--output_root", str(tmp_path / "o"),
            "--experiment_name", "e", "--weights", str(tmp_path / "w.npz")]       # everything else: the harness' (= test.py's) defaults
    assert harness.main(args) == 1
    tag = "apartment_0_000001002"
    sample = tmp_path / "o" / "e" / tag
    ref, src = (harness.load_image(str(tmp_path / "test_640x320" / ("apartment_0_pos%s.jpeg" % n)), h, w)[None] for n in ("000", "001"))
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    o = OracleMSI(weights=weights, coord_net=coord)
    planes = o.inv_depths(1.0, 100.0, d)
    eye = np.eye(4, dtype=np.float32)[None]
    intr = np.array([[[0.032, 0, 0], [0, 1, 0], [0, 0, 1]]], np.float32)
    pos = np.array([[0.02, -0.01, 0.03]], np.float32)
    pred, psv = o.infer_msi(src, ref, None, None, eye, eye, intr, "blend_psv", d, planes, extra_outputs="blend_weights alphas", ngf=ngf)
    _close_png(sample / ("output_tgt_%s.png" % tag), o.deprocess_image(o.msi_render_equirect_view(pred["rgba_layers"], eye, pos, planes, intr))[0])
    _close_png(sample / ("output_depth_%s.png" % tag), o.deprocess_depth_image(o.msi_render_equirect_depth(pred["rgba_layers"], eye, pos, planes, intr))[0])
    u8 = lambda x: np.clip(x, 0, 255).astype("uint8")          # utils.write_image (utils.py:76-81)
    for i in (0, d // 2, d - 1):
        _close_png(sample / ("msi_alpha_%.2d.png" % i), u8(pred["rgba_layers"][0, :, :, i, 3] * 255.0))
        _close_png(sample / ("msi_rgb_%.2d.png" % i), u8((pred["rgba_layers"][0, :, :, i, :3] + 1.) / 2. * 255))
        _close_png(sample / ("blend_weight_%.3d.png" % i), u8(pred["blend_weights"][0, :, :, i] * 255.0))
    for f in ["msi_alpha_%.2d.png" % i for i in range(d)] + ["msi_rgb_%.2d.png" % i for i in range(d)] + \
             ["src_image_%s.png" % tag, "ref_image_%s.png" % tag, "tgt_image_%s.png" % tag]:
        assert (sample / f).exists(), f
    bw, al = np.load(str(sample / "blend_weights.npy")), np.load(str(sample / "alphas.npy"))
    assert bw.shape == al.shape == (1, h, w, d)
    e_bw, e_al = np.abs(bw - pred["blend_weights"]).max(), np.abs(al - pred["alphas"]).max()
    print("configs[0] size, coord=%s: blend_weights %.2e alphas %.2e" % (coord, e_bw, e_al))
    assert e_bw <= 1e-3 and e_al <= 1e-3


def test_harness_flags_unreliable_samples_and_continues(tmp_path):
    """a forward whose LayerNorm statistics leave the fixed-point window (here: a NaN weight -> MSI_NET_STATUS_LN_OVERFLOW) and a
    target position outside the innermost sphere must NOT abort the run: every sample's files are written, the flagged ones carry UNRELIABLE.txt,
    the remaining samples are processed; --strict aborts at the first."""
    from PIL import Image
    from matryodshka_amd import harness, nets
    from matryodshka_amd._native import MsiError
    h, w, d, ngf = 16, 32, 4, 8
    img_dir = tmp_path / "images"; img_dir.mkdir()
    rng = np.random.RandomState(0)
    for name in ("000", "001", "002", "003"):
        arr = np.clip(rng.uniform(0, 255, size=(2 * h, 2 * w, 3)), 0, 255).astype(np.uint8)
        Image.fromarray(arr).save(str(img_dir / ("room_0_pos%s.jpeg" % name)), quality=95)
    cam = tmp_path / "cams.txt"
    # second sample: target 2.5 units from the origin with the innermost sphere at radius 1 (spherical.py:316-318 takes sqrt of a negative number there)
    cam.write_text("room_0 000 001 002 0.032 0.01 -0.02 0.03\nroom_0 000 001 003 0.032 2.5 0.0 0.0\n")
    weights = nets.init_weights(6 * d, 2 * d, ngf, False)
    good = tmp_path / "good.npz"; np.savez(str(good), **weights)
    bad_w = {k: v.copy() for k, v in weights.items()}
    bad_w["conv3_1/weights"][0, 0, 0, 0] = np.nan
    bad = tmp_path / "bad.npz"; np.savez(str(bad), **bad_w)
    common = ["--cameras_glob", str(cam), "--image_dir", str(img_dir), "--height", str(h), "--width", str(w), "--num_msi_planes", str(d),
              "--num_psv_planes", str(d), "--ngf", str(ngf), "--test_outputs", "tgt_image_blend_weights_alphas"]
    # (a) healthy weights: sample 1 clean, sample 2 flagged by the domain guard; both directories exist, the run goes on
    n = harness.main(common + ["--output_root", str(tmp_path / "o1"), "--experiment_name", "e", "--weights", str(good)])
    assert n == 2
    s1, s2 = tmp_path / "o1" / "e" / "room_0_000001002", tmp_path / "o1" / "e" / "room_0_000001003"
    assert (s1 / "output_tgt_room_0_000001002.png").exists() and not (s1 / "UNRELIABLE.txt").exists()
    assert (s2 / "UNRELIABLE.txt").exists() and "innermost sphere" in (s2 / "UNRELIABLE.txt").read_text()
    # (b) a NaN weight: the status word fires on every forward; the files are written all the same and flagged
    n = harness.main(common + ["--output_root", str(tmp_path / "o2"), "--experiment_name", "e", "--weights", str(bad)])
    assert n == 2
    t1 = tmp_path / "o2" / "e" / "room_0_000001002"
    assert (t1 / "UNRELIABLE.txt").exists() and (t1 / "output_tgt_room_0_000001002.png").exists() and (t1 / "blend_weights.npy").exists()
    # (c) --strict: the first flagged sample aborts
    with pytest.raises((MsiError, ValueError)):
        harness.main(common + ["--output_root", str(tmp_path / "o3"), "--experiment_name", "e", "--weights", str(bad), "--strict"])
