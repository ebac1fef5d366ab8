"""test.py-equivalent inference harness (reference test.py:87-281) on the MI355X path.

    python -m matryodshka_amd.harness --cameras_glob 'glob/test/ods/*.txt' --image_dir glob/test_640x320 \
        --weights weights.npz --output_root results --experiment_name ods-wotemp-elpips-coord

Reads the reference's camera files (one sample per line: `scene_id id_ref id_src id_tgt baseline tx
ty tz`, datasets.py:413-425), loads `<image_dir>/<scene>_pos<id>.jpeg` (datasets.py:539), area-resizes
to --height x --width (datasets.py:513-515), runs infer_msi + the equirect RGB / depth render
(test.py:137-159) and writes the reference's output files (test.py:209-281):
  <output_root>/<experiment>/<scene>_<ref><src><tgt>/{tgt_image,output_tgt,output_depth}_<dir>.png,
  src_image/ref_image, psv_plane_%.3d.png, blend_weights.npy, blend_weight_%.3d.png, alphas.npy,
  msi_alpha_%.2d.png, msi_rgb_%.2d.png, and <output_root>/<experiment>/step.txt.
`--checkpoint` reads a TF V2 checkpoint directly (tf_checkpoint.py), `--weights` an .npz of the TF variables
(see nets.variable_shapes); without either Xavier-initialised
weights are used (there is no network access for the pretrained checkpoint), step.txt then says 0.
Host I/O only: all arithmetic is in libmsi_hip.so.
"""
import argparse
import glob
import os

import numpy as np


def write_image(filename, image):
    """utils.write_image (utils.py:76-81): clip to [0,255], uint8, PNG."""
    from PIL import Image
    arr = np.clip(np.asarray(image), 0, 255).astype("uint8")
    if arr.ndim == 3 and arr.shape[2] == 1:
        arr = arr[:, :, 0]
    Image.fromarray(arr).save(filename)


def resize_area(img, height, width):
    """tf.image.resize_area on a float image [H,W,3]: exact box mean for integer factors (the
    Replica test set is rendered at a multiple of 640x320), PIL box filter otherwise."""
    h, w, c = img.shape
    if h == height and w == width:
        return img.astype(np.float32)
    if h % height == 0 and w % width == 0:
        fy, fx = h // height, w // width
        return img.reshape(height, fy, width, fx, c).mean(axis=(1, 3)).astype(np.float32)
    from PIL import Image
    chans = [np.asarray(Image.fromarray(img[..., k].astype(np.float32), mode="F").resize((width, height), Image.BOX))
             for k in range(c)]
    return np.stack(chans, axis=-1).astype(np.float32)


def load_image(path, height, width):
    from PIL import Image
    img = np.asarray(Image.open(path).convert("RGB"), dtype=np.float32) * np.float32(1.0 / 255.0)
    return resize_area(img, height, width)


def parse_camera_files(pattern):
    """Yields (scene_id, (id_ref, id_src, id_tgt), baseline, tgt_pos[3]) per non-empty line."""
    for path in sorted(glob.glob(pattern)):
        with open(path) as f:
            for line in f.read().split("\n"):
                parts = line.split(" ")
                if len(parts) < 8:
                    continue
                yield parts[0], tuple(parts[1:4]), float(parts[4]), [float(x) for x in parts[5:8]]


def run_sample(model, images, baseline, tgt_pos, planes, num_planes, ngf, test_outputs, output_dir, dirname,
               which_color_pred="blend_psv", jitter_pose=None):
    """One iteration of the loop at test.py:199-281.  images = (ref, src, tgt) float [H,W,3] in [0,1]
    (image order ref, src, tgt: data_loader.py:134-136)."""
    import torch
    ref, src, tgt = (torch.from_numpy(np.ascontiguousarray(x[None])) for x in images)
    eye = np.eye(4, dtype=np.float32)[None]
    intr = np.array([[[baseline, 0, 0], [0, 1, 0], [0, 0, 1]]], dtype=np.float32)       # data_loader.py:160
    pos = np.asarray(tgt_pos, dtype=np.float32)[None]
    outs, net_input = model.infer_msi(src, ref, None, None, eye, eye, intr, which_color_pred, num_planes, planes,
                                      extra_outputs="blend_weights alphas psv", ngf=ngf)
    jouts = None
    if jitter_pose is not None:      # test.py:141-147: second inference with the sweep rotated by jitter_pose^-1
        jinv = np.linalg.inv(np.asarray(jitter_pose, dtype=np.float64)).astype(np.float32)
        jouts, _ = model.infer_msi(src, ref, None, None, eye, eye, intr, which_color_pred, num_planes, planes,
                                   extra_outputs="blend_weights alphas psv", ngf=ngf, jitter_pose_inv=jinv)
    os.makedirs(output_dir, exist_ok=True)
    if "tgt_image" in test_outputs:
        rgb, dep = model.msi_render_equirect_view_and_depth(outs["rgba_layers"], eye, pos, planes, intr)
        write_image(os.path.join(output_dir, "tgt_image_%s.png" % dirname), tgt[0].numpy() * 255.0)
        write_image(os.path.join(output_dir, "output_tgt_%s.png" % dirname), model.deprocess_image(rgb)[0].cpu().numpy())
        write_image(os.path.join(output_dir, "output_depth_%s.png" % dirname),
                    model.deprocess_depth_image(dep)[0].cpu().numpy())
        if jouts is not None:        # test.py:160-165 renders through the jitter pose; :237-240 (the reference reads a
            # 'jitter_output_depth' it never produces -- here the depth is rendered the same way)
            jrgb, jdep = model.msi_render_equirect_view_and_depth(jouts["rgba_layers"], np.asarray(jitter_pose, np.float32),
                                                                  pos, planes, intr)
            write_image(os.path.join(output_dir, "jitter_output_tgt_%s.png" % dirname),
                        model.deprocess_image(jrgb)[0].cpu().numpy())
            write_image(os.path.join(output_dir, "jitter_output_depth_%s.png" % dirname),
                        model.deprocess_depth_image(jdep)[0].cpu().numpy())
    if "src_image" in test_outputs:
        write_image(os.path.join(output_dir, "src_image_%s.png" % dirname), src[0].numpy() * 255.0)
    if "ref_image" in test_outputs:
        write_image(os.path.join(output_dir, "ref_image_%s.png" % dirname), ref[0].numpy() * 255.0)
    if "src_output_image" in test_outputs:
        o = model.msi_render_ods_view(outs["rgba_layers"], -1, eye, pos, planes, intr)
        write_image(os.path.join(output_dir, "output_src_%s.png" % dirname), model.deprocess_image(o)[0].cpu().numpy())
    if "ref_output_image" in test_outputs:
        o = model.msi_render_ods_view(outs["rgba_layers"], 1, eye, pos, planes, intr)
        write_image(os.path.join(output_dir, "output_ref_%s.png" % dirname), model.deprocess_image(o)[0].cpu().numpy())
    if "psp" in test_outputs:
        for vw in range(4):
            o = model.msi_render_perspective_view(outs["rgba_layers"], eye, pos, planes, intr, viewing_window=vw)
            write_image(os.path.join(output_dir, "output_ptgt%d_%s.png" % (vw, dirname)),
                        model.deprocess_image(o)[0].cpu().numpy())
    if "psv" in test_outputs:
        psv = outs["psv"].cpu().numpy()
        for j in range(num_planes):
            write_image(os.path.join(output_dir, "psv_plane_%.3d.png" % j), (psv[0, :, :, j * 3:(j + 1) * 3] + 1.) / 2. * 255)
    if "blend" in which_color_pred and "blend_weights" in test_outputs:      # test.py:262
        bw = outs["blend_weights"].cpu().numpy()
        np.save(os.path.join(output_dir, "blend_weights.npy"), bw)
        for i in range(num_planes):
            write_image(os.path.join(output_dir, "blend_weight_%.3d.png" % i), bw[0, :, :, i] * 255.0)
    if "alphas" in test_outputs:
        np.save(os.path.join(output_dir, "alphas.npy"), outs["alphas"].cpu().numpy())
    if "rgba_layers" in test_outputs:
        rgba = outs["rgba_layers"].cpu().numpy()
        for i in range(num_planes):
            write_image(os.path.join(output_dir, "msi_alpha_%.2d.png" % i), rgba[0, :, :, i, 3] * 255.0)
            write_image(os.path.join(output_dir, "msi_rgb_%.2d.png" % i), (rgba[0, :, :, i, :3] + 1.) / 2. * 255)
        if jouts is not None:        # test.py:276-280
            jr = jouts["rgba_layers"].cpu().numpy()
            for i in range(num_planes):
                write_image(os.path.join(output_dir, "jitter_msi_alpha_%.2d.png" % i), jr[0, :, :, i, 3] * 255.0)
                write_image(os.path.join(output_dir, "jitter_msi_rgb_%.2d.png" % i), (jr[0, :, :, i, :3] + 1.) / 2. * 255)
    return outs


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--cameras_glob", default="glob/test/ods/*.txt")
    ap.add_argument("--image_dir", default="glob/test_640x320")
    ap.add_argument("--output_root", default="results")
    ap.add_argument("--experiment_name", default="msi-hip")
    ap.add_argument("--weights", default="", help=".npz of TF variables (net/<layer>/weights ...)")
    ap.add_argument("--checkpoint", default="", help="TF checkpoint prefix or directory (test.py:192-202: "
                    "<checkpoint_dir>/<experiment_name>); read without TensorFlow (matryodshka_amd/tf_checkpoint.py)")
    ap.add_argument("--step", type=int, default=0, help="global step recorded in step.txt (test.py:225-229)")
    ap.add_argument("--height", type=int, default=320)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--num_msi_planes", type=int, default=32)
    ap.add_argument("--min_depth", type=float, default=1.0)
    ap.add_argument("--max_depth", type=float, default=100.0)
    ap.add_argument("--ngf", type=int, default=64)
    ap.add_argument("--coord_net", action="store_true",
                    help="msi_coord_train_net (FLAGS.coord_net, test.py:52, default False = msi_train_net); the released "
                         "ODS models use it (scripts/test/ods-wotemp-elpips-coord-reg.sh).  With --checkpoint / --weights "
                         "the network is inferred from conv1_1/weights' input channels when the flag is absent")
    ap.add_argument("--which_color_pred", default="blend_psv",
                    help="blend_psv | blend_bg | blend_bg_psv | alpha_only (test.py:55-56)")
    ap.add_argument("--transform_inverse_reg", action="store_true",
                    help="also infer with a jittered sweep pose and write jitter_output_* (test.py:141-147, 160-165)")
    ap.add_argument("--rot_factor", type=float, default=0.0)
    ap.add_argument("--tr_factor", type=float, default=0.0)
    ap.add_argument("--random_seed", type=int, default=8964)
    ap.add_argument("--test_outputs", default="src_image_ref_image_tgt_image_psv_rgba_layers_blend_weights_alphas")
    ap.add_argument("--num_runs", type=int, default=-1)
    args = ap.parse_args(argv)

    from . import MSI, nets
    d = args.num_msi_planes
    coord = args.coord_net
    nout = {"blend_psv": 2 * d, "blend_bg": 2 * d + 3, "blend_bg_psv": 3 * d + 3, "alpha_only": d}[args.which_color_pred]
    if args.checkpoint:
        from . import tf_checkpoint
        weights, args.step = tf_checkpoint.network_weights(args.checkpoint)
    elif args.weights:
        weights = dict(np.load(args.weights))
    else:
        weights = nets.init_weights(6 * d, nout, args.ngf, coord)
    if (args.checkpoint or args.weights) and not coord:
        w0 = nets._lookup(weights, "conv1_1/weights")
        coord = w0.shape[2] == 6 * d + 1          # nets.py:260-270: CoordNet appends one input channel
    model = MSI(weights=weights, coord_net=coord)
    jitter = None
    if args.transform_inverse_reg:
        from . import poses
        jitter = poses.random_rotation(args.rot_factor, args.tr_factor, np.random.RandomState(args.random_seed))
    planes = model.inv_depths(args.min_depth, args.max_depth, d)
    exp_dir = os.path.join(args.output_root, args.experiment_name)
    os.makedirs(exp_dir, exist_ok=True)
    n = 0
    for scene, ids, baseline, tgt_pos in parse_camera_files(args.cameras_glob):
        if 0 <= args.num_runs <= n:
            break
        images = [load_image(os.path.join(args.image_dir, "%s_pos%s.jpeg" % (scene, i)), args.height, args.width)
                  for i in ids]
        dirname = "%s_%s%s%s" % (scene, ids[0], ids[1], ids[2])
        out_dir = os.path.join(exp_dir, dirname)
        print("Saving to %s" % out_dir)
        if n == 0:
            with open(os.path.join(exp_dir, "step.txt"), "w") as f:
                f.write("%d" % args.step)
        run_sample(model, images, baseline, tgt_pos, planes, d, args.ngf, args.test_outputs, out_dir, dirname,
                   which_color_pred=args.which_color_pred, jitter_pose=jitter)
        n += 1
    print("processed %d samples" % n)
    return n


if __name__ == "__main__":
    main()
