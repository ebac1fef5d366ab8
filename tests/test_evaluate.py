"""KATs of the eval.py counterpart (matryodshka_amd/evaluate.py: tf.image.ssim / psnr semantics, eval.py:127-174)."""
import json
import os

import numpy as np
import pytest
from scipy import signal

from matryodshka_amd import evaluate as E


def test_identical_images():
    x = np.random.RandomState(0).uniform(0, 255, (32, 48, 3))
    assert E.psnr(x, x) == float("inf")
    assert abs(E.ssim(x, x) - 1.0) < 1e-12


def test_psnr_known_values():
    x = np.full((16, 16, 3), 100.0)
    assert abs(E.psnr(x, x + 5.0) - 20 * np.log10(255.0 / 5.0)) < 1e-9
    y = x.copy(); y[0, 0, 0] += 255.0            # one wrong sample of 768
    assert abs(E.psnr(x, y) - 10 * np.log10(768.0)) < 1e-9


def test_ssim_constant_images_is_the_luminance_term():
    a, b = 50.0, 80.0
    c1 = (0.01 * 255) ** 2
    got = E.ssim(np.full((20, 24, 3), a), np.full((20, 24, 3), b))
    assert abs(got - (2 * a * b + c1) / (a * a + b * b + c1)) < 1e-12


def test_ssim_against_a_direct_2d_formulation():
    """Independent restatement: full 2-D window through scipy.signal.correlate2d, per channel."""
    rng = np.random.RandomState(3)
    x = rng.uniform(0, 255, (24, 30, 3))
    y = np.clip(x + rng.normal(0, 20, x.shape), 0, 255)
    win = E._gauss_window()
    assert abs(win.sum() - 1.0) < 1e-15 and win.shape == (11, 11) and np.allclose(win, win.T)
    c1, c2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2
    per_c = []
    for c in range(3):
        f = lambda im: signal.correlate2d(im, win, mode="valid")
        mx, my = f(x[..., c]), f(y[..., c])
        sxy = f(x[..., c] * y[..., c]) - mx * my
        sxx_yy = f(x[..., c] ** 2 + y[..., c] ** 2) - mx * mx - my * my
        per_c.append((((2 * mx * my + c1) / (mx * mx + my * my + c1)) * ((2 * sxy + c2) / (sxx_yy + c2))).mean())
    assert abs(E.ssim(x, y) - np.mean(per_c)) < 1e-10
    assert E.ssim(x, y) < 0.99
    with pytest.raises(ValueError):
        E.ssim(x[:8], y[:8])                     # smaller than the 11x11 window


def _write_example(root, ex, rng, depth_level):
    from PIL import Image
    d = root / ex
    d.mkdir(parents=True)
    tgt = rng.randint(0, 255, (16, 32, 3)).astype(np.uint8)
    out = np.clip(tgt.astype(int) + 3, 0, 255).astype(np.uint8)
    Image.fromarray(tgt).save(str(d / ("tgt_image_%s.png" % ex)))
    Image.fromarray(out).save(str(d / ("output_tgt_%s.png" % ex)))
    Image.fromarray(np.full((16, 32, 3), depth_level, np.uint8)).save(str(d / ("output_depth_%s.png" % ex)))


def test_result_tree_evaluation(tmp_path):
    """eval.py:62-78 / :102-126: video frames are NOT scored for SSIM / PSNR, and consecutive pairs never cross scenes."""
    rng = np.random.RandomState(1)
    root = tmp_path / "m"
    for ex in ("room_0_000001002", "office_0_003004005"):             # plain test examples
        _write_example(root, ex, rng, 0)
    for k, ex in enumerate(("video_room_0_001", "video_room_0_002", "video_room_0_003")):
        _write_example(root, ex, rng, 10 * k)
    for k, ex in enumerate(("video_office_0_001", "video_office_0_002")):
        _write_example(root, ex, rng, 100 + 7 * k)
    (root / "step.txt").write_text("0")
    table = E.main(["--result_root", str(tmp_path), "--model_names", "m", "--output_table", str(tmp_path / "t.json"),
                    "--video", "--videos", "room_0 office_0"])
    assert table["examples"] == ["office_0_003004005", "room_0_000001002"]          # no 'video' entry, no step.txt
    assert all(30 < p[0] < 45 for p in table["psnr"]) and all(0.9 < s[0] <= 1 for s in table["ssim"])
    cons = table["consecutive"]
    assert [len(c) for c in cons] == [2, 1]                                         # pairs per scene, none across scenes
    assert cons[0][0]["frames"] == ["video_room_0_001", "video_room_0_002"] and cons[1][0]["frames"] == ["video_office_0_001", "video_office_0_002"]
    assert abs(cons[0][0]["diffs"][0][1] - 10.0) < 1e-6 and abs(cons[1][0]["diffs"][0][1] - 7.0) < 1e-6
    assert json.load(open(str(tmp_path / "t.json")))["model_names"] == ["m"]
    # a directory without tgt_image_* is a clear error, not an IndexError
    (root / "broken_example").mkdir()
    with pytest.raises(FileNotFoundError):
        E.main(["--result_root", str(tmp_path), "--model_names", "m", "--output_table", str(tmp_path / "t2.json")])
