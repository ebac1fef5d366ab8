"""Quality metrics of the reference's eval.py on written result files (host numpy; no device work).

    python -m matryodshka_amd.evaluate --result_root results --model_names msi-hip --output_table eval.json

eval.py:127-145 `evaluate_one`: per example directory, `tgt_image_*` (ground truth) against `output_tgt_*` (render):
  tf.image.ssim(pred, tgt, max_val=255) and tf.image.psnr(pred, tgt, max_val=255)
eval.py:147-174 `evaluate_consecutive_one`: mean absolute frame-to-frame difference of `output_tgt_*` / `output_depth_*`
of consecutive video frames (temporal consistency).  E-LPIPS (eval.py:138, the vendored elpips/ network + weights) is
out of scope: it is a training / evaluation network of its own, not part of the infer -> render path.

tf.image.ssim / psnr semantics restated here are [TF-knowledge] (TF 1.14 image_ops_impl.py): 11x11 Gaussian window,
sigma 1.5, normalised by softmax of -(x^2+y^2)/(2 sigma^2), VALID depth-wise filtering, k1 = 0.01, k2 = 0.03,
luminance = (2 mu_x mu_y + c1) / (mu_x^2 + mu_y^2 + c1), cs = (2 (E[xy] - mu_x mu_y) + c2) / (E[x^2 + y^2] - mu_x^2 - mu_y^2 + c2),
SSIM = mean over window positions per channel, then mean over channels; PSNR = 20 log10(max) - 10 log10(mean squared error).
"""
import argparse
import glob
import json
import os

import numpy as np


def _gauss_window(size=11, sigma=1.5):
    coords = np.arange(size, dtype=np.float64) - (size - 1) / 2.0
    g = -0.5 * coords * coords / (sigma * sigma)
    g2 = g[None, :] + g[:, None]
    e = np.exp(g2 - g2.max())
    return e / e.sum()          # softmax over the 121 taps


def _filter_valid(x, win):
    """VALID cross-correlation of [H,W,C] with the separable window (its rows and columns are proportional)."""
    k = win.shape[0]
    col = win.sum(axis=1)                       # 1-D factors of the (exactly separable) normalised Gaussian
    row = win.sum(axis=0)
    h, w, _ = x.shape
    tmp = np.zeros((h - k + 1, w, x.shape[2]))
    for i in range(k):
        tmp += col[i] * x[i:i + h - k + 1]
    out = np.zeros((h - k + 1, w - k + 1, x.shape[2]))
    for j in range(k):
        out += row[j] * tmp[:, j:j + w - k + 1]
    return out


def ssim(img1, img2, max_val=255.0, filter_size=11, filter_sigma=1.5, k1=0.01, k2=0.03):
    """tf.image.ssim for one image pair [H,W,C] (eval.py:139)."""
    x = np.asarray(img1, dtype=np.float64)
    y = np.asarray(img2, dtype=np.float64)
    if x.shape != y.shape or x.ndim != 3:
        raise ValueError("ssim: images must be [H,W,C] and agree")
    if min(x.shape[0], x.shape[1]) < filter_size:
        raise ValueError("ssim: image smaller than the %dx%d window" % (filter_size, filter_size))
    win = _gauss_window(filter_size, filter_sigma)
    c1, c2 = (k1 * max_val) ** 2, (k2 * max_val) ** 2
    mx, my = _filter_valid(x, win), _filter_valid(y, win)
    num0, den0 = mx * my * 2.0, mx * mx + my * my
    lum = (num0 + c1) / (den0 + c1)
    num1 = _filter_valid(x * y, win) * 2.0
    den1 = _filter_valid(x * x + y * y, win)
    cs = (num1 - num0 + c2) / (den1 - den0 + c2)
    return float((lum * cs).mean(axis=(0, 1)).mean())


def psnr(img1, img2, max_val=255.0):
    """tf.image.psnr (eval.py:140); inf for identical images."""
    x = np.asarray(img1, dtype=np.float64)
    y = np.asarray(img2, dtype=np.float64)
    mse = float(((x - y) ** 2).mean())
    if mse == 0.0:
        return float("inf")
    return float(20.0 * np.log10(max_val) - 10.0 * np.log10(mse))


def load_image(path):
    """eval.py load_image: the PNG as float values 0..255, [H,W,3]."""
    from PIL import Image
    return np.asarray(Image.open(path).convert("RGB"), dtype=np.float32)


def evaluate_one(result_root, model_name, example):
    """eval.py:127-145 without the E-LPIPS term: (ssim, psnr) of output_tgt_* against tgt_image_*."""
    d = os.path.join(result_root, model_name, example)
    tgts, preds = sorted(glob.glob(os.path.join(d, "tgt_image_*"))), sorted(glob.glob(os.path.join(d, "output_tgt_*")))
    if not tgts or not preds:
        raise FileNotFoundError("%s: needs tgt_image_* and output_tgt_* (run the harness with 'tgt_image' in --test_outputs)" % d)
    tgt, pred = load_image(tgts[0]), load_image(preds[0])
    return ssim(pred, tgt, 255.0), psnr(pred, tgt, 255.0)


def _pick(files):
    """eval.py:155-158: of two candidates prefer the one whose name contains 'blurred'."""
    files = sorted(files)
    if len(files) > 1 and "blurred" in files[1]:
        return files[1]
    return files[0]


def evaluate_consecutive_one(result_root, model_name, pair):
    """eval.py:147-174: mean |frame1 - frame2| of the rendered target and depth images of two consecutive frames."""
    d1, d2 = (os.path.join(result_root, model_name, e) for e in pair)
    t1, t2 = (load_image(_pick(glob.glob(os.path.join(d, "output_tgt_*")))) for d in (d1, d2))
    z1, z2 = (load_image(_pick(glob.glob(os.path.join(d, "output_depth_*")))) for d in (d1, d2))
    return float(np.abs(t1 - t2).mean()), float(np.abs(z1 - z2).mean())


def _example_dirs(result_root, model_name):
    root = os.path.join(result_root, model_name)
    return [e for e in os.listdir(root) if not e.endswith(".txt") and os.path.isdir(os.path.join(root, e))]


def collect_examples(result_root, model_names):
    """eval.py:62-78: the NON-video example directories present for every model (`step.txt` and every entry whose name
    contains 'video' are skipped; an example missing for some model is an error there: `assert not skipped`)."""
    counts = {}
    for m in model_names:
        for e in _example_dirs(result_root, m):
            if "video" in e:
                continue
            counts[e] = counts.get(e, 0) + 1
    skipped = sorted(k for k, v in counts.items() if v != len(model_names))
    if skipped:
        raise ValueError("examples missing for some model: %s" % ", ".join(skipped[:5]))
    return sorted(counts)


def collect_video_consecutive_examples(result_root, model_names, scene_names):
    """eval.py:102-126: per scene (an entry is a frame of scene s when its name contains 'video' and s), the frame
    directories present for every model sorted by name, as consecutive pairs WITHIN the scene: [scene][pair] -> (e0, e1)."""
    out = []
    for scene in scene_names:
        counts = {}
        for m in model_names:
            for e in _example_dirs(result_root, m):
                if "video" in e and scene in e:
                    counts[e] = counts.get(e, 0) + 1
        skipped = sorted(k for k, v in counts.items() if v != len(model_names))
        if skipped:
            raise ValueError("video frames missing for some model: %s" % ", ".join(skipped[:5]))
        frames = sorted(counts)
        out.append([(frames[j], frames[j + 1]) for j in range(len(frames) - 1)])
    return out


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--result_root", default="results")
    ap.add_argument("--model_names", default="msi-hip", help="comma-separated experiment names under result_root")
    ap.add_argument("--output_table", default="eval.json")
    ap.add_argument("--video", action="store_true",
                    help="eval_type on_video (eval.py:185-215): frame-to-frame differences of the 'video' examples, per scene")
    ap.add_argument("--videos", default="room_0 room_2 office_0 apartment_0", help="scene names of the video examples (eval.py:43)")
    a = ap.parse_args(argv)
    models = [m for m in a.model_names.split(",") if m]
    examples = collect_examples(a.result_root, models)
    table = {"model_names": models, "examples": examples, "ssim": [], "psnr": []}
    for e in examples:
        scores = [evaluate_one(a.result_root, m, e) for m in models]
        table["ssim"].append([s[0] for s in scores])
        table["psnr"].append([s[1] for s in scores])
    table["mean_ssim"] = [float(np.mean([r[i] for r in table["ssim"]])) for i in range(len(models))] if examples else []
    table["mean_psnr"] = [float(np.mean([r[i] for r in table["psnr"]])) for i in range(len(models))] if examples else []
    if a.video:
        scenes = [s for s in a.videos.split(" ") if s]
        pairs = collect_video_consecutive_examples(a.result_root, models, scenes)
        table["video_scenes"] = scenes
        table["consecutive"] = [[{"frames": list(pr), "diffs": [list(evaluate_consecutive_one(a.result_root, m, pr)) for m in models]}
                                 for pr in scene_pairs] for scene_pairs in pairs]
    with open(a.output_table, "w") as f:
        json.dump(table, f)
    print("Output written to %s" % a.output_table)
    return table


if __name__ == "__main__":
    main()
