"""Oracle (TEST INFRASTRUCTURE ONLY, parity unpinned -- see oracle/__init__.py):
CPU restatement of matryodshka/msi.py's infer -> render path with the
reference's method names and argument order, on numpy arrays.

Reference followed: matryodshka/msi.py
  :40-52, 68-88, 119-147, 276-289   infer_msi (blend_psv; :166-275 blend_bg, blend_bg_psv, alpha_only)
  :384-429                          msi_render_equirect_depth / _view
  :431-452                          msi_render_equirect_view_single
  :1094-1130, 1157-1161             format_network_input / sweep_src
  :1163-1194                        preprocess / deprocess
  :1196-1217                        inv_depths
Hidden graph inputs of the reference (`ref_pose_inv:0`, global FLAGS) are
explicit keyword arguments here; defaults reproduce test.py's ODS path
(test.py:106-159).
"""
import numpy as np

from . import geometry as G
from . import nets

F = np.float32


def matmul4(a, b):
    """tf.matmul of [B,4,4] fp32 poses (msi.py:1120, :1125) with the four products summed k = 0..3 in fp32, one
    rounding per operation (the order Eigen uses is not observable from the reference [TF-knowledge]; a BLAS call's
    is not even fixed).  Identity operands -- the ODS test path -- are exact in any order."""
    a = np.asarray(a, dtype=F).reshape(-1, 4, 4)
    b = np.asarray(b, dtype=F).reshape(-1, 4, 4)
    n = max(a.shape[0], b.shape[0])
    a, b = np.broadcast_to(a, (n, 4, 4)), np.broadcast_to(b, (n, 4, 4))
    out = (a[:, :, 0:1] * b[:, 0:1, :]).astype(F)
    for k in range(1, 4):
        out = (out + (a[:, :, k:k + 1] * b[:, k:k + 1, :]).astype(F)).astype(F)
    return out


class MSI(object):
    def __init__(self, weights=None, coord_net=False, input_type='ODS', dtype='f32'):     # FLAGS.coord_net default, test.py:52
        self.weights = weights
        self.coord_net = coord_net
        # 'bf16': BASELINE configs[2] as the build defines it (the reference has no bf16 code): the sweep
        # volume is rounded to bf16 and the network uses bf16 operands (oracle/nets.py forward(bf16=True))
        self.dtype = dtype
        self.input_type = input_type     # FLAGS.input_type: 'ODS' | 'PP' (msi.py:1157-1161)

    # -- msi.py:1196-1217 ---------------------------------------------------
    def inv_depths(self, start_depth, end_depth, num_depths):
        inv_start_depth = 1.0 / start_depth
        inv_end_depth = 1.0 / end_depth
        depths = [start_depth, end_depth]
        for i in range(1, num_depths - 1):
            fraction = float(i) / float(num_depths - 1)
            inv_depth = inv_start_depth + (inv_end_depth - inv_start_depth) * fraction
            depths.append(1.0 / inv_depth)
        depths = sorted(depths)
        return depths[::-1]

    # -- msi.py:1163-1194; tf.image.convert_image_dtype [TF-knowledge] --------
    def preprocess_image(self, image):
        image = np.asarray(image)
        if image.dtype == np.uint8:
            image = image.astype(F) * F(1.0 / 255.0)
        else:
            image = image.astype(F)
        return image * F(2) - F(1)

    @staticmethod
    def _to_uint8(x):
        # convert_image_dtype(float -> uint8, saturate=False): cast(x * 255.5)
        # (truncation); out-of-range float->uint8 casts are undefined in TF, the
        # oracle clamps so comparisons are well defined (inputs are in range
        # whenever alpha, rgb are -- they are convex combinations).
        y = x.astype(F) * F(255.5)
        return np.clip(np.trunc(y), 0, 255).astype(np.uint8)

    def deprocess_image(self, image):
        image = (np.asarray(image, dtype=F) + F(1.0)) / F(2.0)
        return self._to_uint8(image)

    def deprocess_depth_image(self, image):
        return self._to_uint8(np.asarray(image, dtype=F))

    # -- msi.py:1094-1130 -----------------------------------------------------
    def format_network_input(self, ref_image, src_image, ref_pose, src_pose, planes,
                             intrinsics, ref_pose_inv=None, jitter_pose_inv=None):
        ref_image = np.asarray(ref_image, dtype=F)
        src_image = np.asarray(src_image, dtype=F)
        ref_pose = np.asarray(ref_pose, dtype=F)
        src_pose = np.asarray(src_pose, dtype=F)
        if ref_pose_inv is None:
            ref_pose_inv = np.linalg.inv(ref_pose.astype(np.float64)).astype(F)
        if jitter_pose_inv is not None:     # FLAGS.jitter, msi.py:1118-1120
            ref_pose_inv = matmul4(ref_pose_inv, jitter_pose_inv)
        net_input = []
        # The reference concatenates [ref_pose, src_pose] on the batch axis and
        # so only works for B=1 (msi.py:1109-1110); here batch element b uses
        # (ref_pose[b], src_pose[b]) -- "B independent B=1 evaluations".
        for i, (img, pose) in enumerate(((ref_image, ref_pose), (src_image, src_pose))):
            curr_pose = matmul4(pose, ref_pose_inv)
            order = 1 if (i % 2) == 0 else -1
            if self.input_type == 'ODS':
                net_input.append(G.ods_sphere_sweep(img, order, planes, curr_pose, intrinsics))
            else:   # sweep_src, msi.py:1157-1161 (ref_pose_inv is `interp_pose_inv:0` there, :1113)
                net_input.append(G.perspective_plane_sweep(img, planes, curr_pose, intrinsics))
        net_input = np.concatenate(net_input, axis=3)
        return nets.bf16_round(net_input) if self.dtype == 'bf16' else net_input

    # -- msi.py:40-289 (blend_psv) -------------------------------------------
    def infer_msi(self, raw_src_image, raw_ref_image, raw_hres_src_image, raw_hres_ref_image,
                  ref_pose, src_pose, intrinsics, which_color_pred, num_msi_planes, psv_planes,
                  extra_outputs='', ngf=64, ref_pose_inv=None, jitter_pose_inv=None):
        src_image = self.preprocess_image(raw_src_image)
        ref_image = self.preprocess_image(raw_ref_image)
        net_input = self.format_network_input(ref_image, src_image, ref_pose, src_pose,
                                              psv_planes, intrinsics, ref_pose_inv=ref_pose_inv,
                                              jitter_pose_inv=jitter_pose_inv)
        msi_pred = nets.forward(self.weights, net_input, coord_net=self.coord_net, bf16=self.dtype == 'bf16')
        pred = self.assemble(net_input, msi_pred, num_msi_planes, extra_outputs, which_color_pred)
        return pred, net_input

    def assemble(self, net_input, msi_pred, num_msi_planes, extra_outputs='', which_color_pred='blend_psv'):
        """layer_prediction: blend_psv msi.py:130-147, blend_bg :177-188, blend_bg_psv :223-242, alpha_only :258-268."""
        d = num_msi_planes
        b, h, w, _ = net_input.shape
        net_input = np.asarray(net_input, dtype=F)
        one = F(1)
        blend_weights = bg_blend_weights = None
        if which_color_pred == 'alpha_only':
            assert msi_pred.shape[-1] == d
            alphas = (msi_pred[..., :d] + F(1.)) / F(2.)
        else:
            blend_weights = (msi_pred[..., :d] + F(1.)) / F(2.)
            alphas = (msi_pred[..., d:2 * d] + F(1.)) / F(2.)
        if which_color_pred == 'blend_psv':
            assert msi_pred.shape[-1] == 2 * d
        elif which_color_pred == 'blend_bg':
            assert msi_pred.shape[-1] == 2 * d + 3
            pred_bg = msi_pred[..., -3:]                       # not rescaled (msi.py:177)
        elif which_color_pred == 'blend_bg_psv':
            assert msi_pred.shape[-1] == 3 * d + 3
            bg_blend_weights = (msi_pred[..., 2 * d:3 * d] + F(1.)) / F(2.)
            pred_bg = msi_pred[..., -3:]
        elif which_color_pred != 'alpha_only':
            raise ValueError(which_color_pred)
        rgba = np.empty((b, h, w, d, 4), dtype=F)
        for i in range(d):
            fg_rgb = net_input[..., i * 3:(1 + i) * 3]
            bg_rgb = net_input[..., (d + i) * 3:(d + 1 + i) * 3]
            if which_color_pred == 'alpha_only':
                rgb = fg_rgb
            else:
                wgt = blend_weights[..., i:i + 1]
                if which_color_pred == 'blend_bg':
                    rgb = wgt * fg_rgb + (one - wgt) * pred_bg
                else:
                    rgb = wgt * fg_rgb + (one - wgt) * bg_rgb
                    if which_color_pred == 'blend_bg_psv':
                        bg_w = bg_blend_weights[..., i:i + 1]
                        rgb = bg_w * rgb + (one - bg_w) * pred_bg
            rgba[..., i, :3] = rgb
            rgba[..., i, 3] = alphas[..., i]
        pred = {'rgba_layers': rgba}
        if 'blend_weights' in extra_outputs and 'blend' in which_color_pred:      # msi.py:280-284
            pred['blend_weights'] = blend_weights
            if bg_blend_weights is not None:
                pred['bg_blend_weights'] = bg_blend_weights
        if 'alpha' in extra_outputs:
            pred['alphas'] = alphas
        if 'psv' in extra_outputs:
            pred['psv'] = net_input
        return pred

    # -- msi.py:384-452 ---------------------------------------------------------
    def _project(self, rgba_layers, tgt_pose_rt, tgt_pos, planes):
        rgba_layers = np.asarray(rgba_layers, dtype=F)
        tgt_pose_rt = np.asarray(tgt_pose_rt, dtype=F)
        batch_size = tgt_pose_rt.shape[0]
        depths = np.tile(np.asarray(planes, dtype=F).reshape(-1, 1), (1, batch_size))
        layers = np.transpose(rgba_layers, (3, 0, 1, 2, 4))
        return G.projective_forward_sphere(layers, tgt_pose_rt, tgt_pos, depths)

    def msi_render_equirect_view(self, rgba_layers, tgt_pose_rt, tgt_pos, planes, intrinsics):
        proj = self._project(rgba_layers, tgt_pose_rt, tgt_pos, planes)
        return G.over_composite([proj[i] for i in range(len(planes))])

    def msi_render_equirect_depth(self, rgba_layers, tgt_pose_rt, tgt_pos, planes, intrinsics):
        proj = self._project(rgba_layers, tgt_pose_rt, tgt_pos, planes)
        return G.over_composite_depth([proj[i] for i in range(len(planes))])

    def msi_render_equirect_view_single(self, rgba_layers, tgt_pose_rt, tgt_pos, planes, intrinsics):
        return self._project(rgba_layers, tgt_pose_rt, tgt_pos, planes)

    # -- msi.py:502-525 -------------------------------------------------------------
    def msi_render_ods_view(self, rgba_layers, order, jitter_pose, tgt_pos, planes, intrinsics):
        rgba_layers = np.asarray(rgba_layers, dtype=F)
        intrinsics = np.asarray(intrinsics, dtype=F)
        batch_size = intrinsics.shape[0]
        depths = np.tile(np.asarray(planes, dtype=F).reshape(-1, 1), (1, batch_size))
        layers = np.transpose(rgba_layers, (3, 0, 1, 2, 4))
        jitter_pose = np.broadcast_to(np.asarray(jitter_pose, dtype=F).reshape(-1, 4, 4), (batch_size, 4, 4))
        proj = G.projective_forward_ods(layers, order, intrinsics, jitter_pose, depths)
        return G.over_composite([proj[i] for i in range(len(planes))])

    # -- msi.py:475-500 -------------------------------------------------------------
    def msi_render_perspective_view(self, rgba_layers, tgt_pose_rt, tgt_pos, planes, intrinsics,
                                    viewing_window=3, psp_height=270, psp_width=480):
        rgba_layers = np.asarray(rgba_layers, dtype=F)
        batch_size = np.asarray(tgt_pose_rt).reshape(-1, 4, 4).shape[0]
        depths = np.tile(np.asarray(planes, dtype=F).reshape(-1, 1), (1, batch_size))
        layers = np.transpose(rgba_layers, (3, 0, 1, 2, 4))
        proj = G.projective_forward_sphere_to_perspective(layers, tgt_pos, depths, viewing_window, psp_height, psp_width)
        return G.over_composite([proj[i] for i in range(len(planes))])


    # -- test.py:283-394: high-res re-render, plane by plane as the reference does ---------------
    @staticmethod
    def resize_bilinear_align_corners(x, out_h, out_w):
        """tf.image.resize(..., BILINEAR, align_corners=True) [TF-knowledge], x [B,H,W,C]."""
        x = np.asarray(x, dtype=F)
        _, h, w, _ = x.shape
        sy = F(h - 1) / F(out_h - 1) if out_h > 1 else F(0)
        sx = F(w - 1) / F(out_w - 1) if out_w > 1 else F(0)
        fy = (np.arange(out_h, dtype=F) * sy).astype(F)
        fx = (np.arange(out_w, dtype=F) * sx).astype(F)
        y0 = np.floor(fy).astype(int); y1 = np.minimum(np.ceil(fy).astype(int), h - 1)
        x0 = np.floor(fx).astype(int); x1 = np.minimum(np.ceil(fx).astype(int), w - 1)
        yl = (fy - y0.astype(F))[None, :, None, None]
        xl = (fx - x0.astype(F))[None, None, :, None]
        tl = x[:, y0][:, :, x0]; tr = x[:, y0][:, :, x1]
        bl = x[:, y1][:, :, x0]; br = x[:, y1][:, :, x1]
        top = tl + (tr - tl) * xl
        bot = bl + (br - bl) * xl
        return (top + (bot - top) * yl).astype(F)

    def render_hres(self, blend_weights, alphas, raw_hres_ref_image, raw_hres_src_image, ref_pose, src_pose,
                    tgt_pose_rt, tgt_pos, planes, intrinsics):
        hres_ref = self.preprocess_image(raw_hres_ref_image)
        hres_src = self.preprocess_image(raw_hres_src_image)
        b, hh, hw, _ = hres_ref.shape
        n = len(planes)
        out, depth = None, None
        for i in range(n):
            net_in = self.format_network_input(hres_ref, hres_src, ref_pose, src_pose, planes[i:i + 1], intrinsics)
            uw = self.resize_bilinear_align_corners(np.asarray(blend_weights)[..., i:i + 1], hh, hw)
            ua = self.resize_bilinear_align_corners(np.asarray(alphas)[..., i:i + 1], hh, hw)
            rgb = uw * net_in[..., 0:3] + (F(1) - uw) * net_in[..., 3:6]
            layer = np.concatenate([rgb, ua], axis=3).reshape(b, hh, hw, 1, 4)
            warped = self.msi_render_equirect_view_single(layer, tgt_pose_rt, tgt_pos, planes[i:i + 1], intrinsics)[0]
            wrgb, walpha = warped[..., :3], warped[..., 3:]
            a3 = np.tile(walpha, (1, 1, 1, 3))
            if i == 0:
                out, depth = wrgb, np.zeros_like(wrgb)
            else:
                out = out * (F(1.) - walpha) + wrgb * walpha
                depth = F(i / n) * a3 + depth * (F(1.0) - a3)
        return out.astype(F), depth.astype(F)


    # -- msi.py:527-548 -------------------------------------------------------------
    def mpi_render_view(self, rgba_layers, tgt_pose, planes, intrinsics, intrinsics_inv=None):
        rgba_layers = np.asarray(rgba_layers, dtype=F)
        tgt_pose = np.asarray(tgt_pose, dtype=F).reshape(-1, 4, 4)
        intrinsics = np.asarray(intrinsics, dtype=F)
        if intrinsics_inv is None:
            intrinsics_inv = np.linalg.inv(intrinsics.astype(np.float64)).astype(F)
        batch_size = tgt_pose.shape[0]
        depths = np.tile(np.asarray(planes, dtype=F).reshape(-1, 1), (1, batch_size))
        layers = np.transpose(rgba_layers, (3, 0, 1, 2, 4))
        proj = G.projective_forward_homography(layers, intrinsics, np.asarray(intrinsics_inv, dtype=F), tgt_pose, depths)
        return G.over_composite([proj[i] for i in range(len(planes))])
