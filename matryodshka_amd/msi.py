"""MI355X-native counterpart of the reference's `matryodshka.msi.MSI` class for the
infer -> render hot path, with the reference's method names and argument order
(so it drops into a test.py-style harness, test.py:127-159).

All arithmetic runs in libmsi_hip.so (hand-written HIP for gfx950) through the C
ABI of include/msi_hip.h; torch is used for device memory, streams and tiny 4x4
pose algebra only.  There is no CPU path: tensors must live on a HIP device.

Differences from the reference that are forced by leaving TF graph mode:
  * hidden graph inputs (`ref_pose_inv:0`, msi.py:1115) are keyword arguments with
    the test.py defaults (ref_pose_inv = inverse(ref_pose), no jitter);
  * global FLAGS become constructor arguments (coord_net);
  * batch semantics: the reference only works for B=1 (test.py:89, msi.py:1109);
    here B frames are B independent B=1 evaluations;
  * `rgba_layers` keeps the public [B,H,W,D,4] shape but is a permuted VIEW of the
    D-major [B,D,H,W,4] stack the kernels use (the layout msi.py:422 transposes to).
"""
import numpy as np
import torch

from . import _native as N
from . import nets


def _ptr(t):
    return 0 if t is None else t.data_ptr()


def _on_own_device(cls):
    """Every public method runs with the model's device current: the native library launches on the stream it is
    handed, plans read the CU count of the current device, and per-device kernel attributes are set on first use --
    a caller driving several GPUs from one thread must not have to remember torch.cuda.set_device."""
    import functools

    def wrap(fn):
        @functools.wraps(fn)
        def inner(self, *args, **kwargs):
            if torch.cuda.current_device() == self.device.index:
                return fn(self, *args, **kwargs)
            with torch.cuda.device(self.device):
                return fn(self, *args, **kwargs)
        return inner

    for name, fn in list(vars(cls).items()):
        if callable(fn) and not name.startswith("__") and not isinstance(fn, (staticmethod, classmethod)) and name != "inv_depths":
            setattr(cls, name, wrap(fn))
    return cls


@_on_own_device
class MSI(object):
    """Class definition for the MSI inference module (reference: msi.py:33-38)."""

    COLOR_SCHEMES = {'blend_psv': 0, 'blend_bg': 1, 'blend_bg_psv': 2, 'alpha_only': 3}   # MSI_COLOR_* of msi_hip.h

    def __init__(self, weights=None, coord_net=None, device=None, input_type='ODS', dtype='f32'):
        """coord_net: FLAGS.coord_net (test.py:52, default False = msi_train_net; the released ODS models are run with
        --coord_net = msi_coord_train_net, scripts/test/ods-wotemp-elpips-coord-reg.sh).  None (default): taken from the
        weights -- CoordNet's conv1_1/weights carry one extra input channel (nets.py:260-270), and the sweep volume's
        6 D channels are even -- or False without weights."""
        if not torch.cuda.is_available():
            raise RuntimeError("matryodshka_amd.MSI needs a HIP device (no CPU fallback)")
        self.device = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self._coord_explicit = coord_net is not None
        self.coord_net = bool(coord_net)
        if input_type not in ('ODS', 'PP'):
            raise ValueError("input_type must be 'ODS' or 'PP' (FLAGS.input_type, msi.py:1157-1161)")
        self.input_type = input_type
        # 'bf16' = BASELINE configs[2]: the sweep volume, the weights and the activations of the network are
        # bf16 (fp32 accumulate, fp32 LayerNorm statistics, fp32 prediction); geometry stays fp32
        if dtype not in ('f32', 'bf16'):
            raise ValueError("dtype must be 'f32' or 'bf16'")
        if dtype == 'bf16' and input_type != 'ODS':
            raise NotImplementedError("dtype='bf16' is built for the ODS path only")
        self.dtype = dtype
        self._weights = None
        self._blob_cache = {}     # (in_channels, num_outputs, ngf) -> np blob
        self._packed_cache = {}   # desc key -> device tensor
        self._ws_cache = {}       # desc key -> (native plan, device workspace)
        self.net_options = {}     # msi_net_plan_set_option key -> value, applied to plans created from now on (tests)
        self._trig_cache = {}     # (H, W) -> device tensor
        self._planes_cache = {}   # tuple(planes) -> device tensor
        self._render_status = torch.zeros(1, dtype=torch.int32, device=self.device)   # msi_render_*'s status_device word
        if weights is not None:
            self.load_weights(weights)

    # ------------------------------------------------------------------ plumbing
    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def _f32(self, x):
        if not torch.is_tensor(x):
            x = torch.as_tensor(np.asarray(x, dtype=np.float32))
        return x.to(device=self.device, dtype=torch.float32).contiguous()

    def _trig(self, height, width):
        key = (height, width)
        t = self._trig_cache.get(key)
        if t is None:
            host = np.empty(N.lib.msi_trig_table_floats(height, width), dtype=np.float32)
            N.check(N.lib.msi_build_trig_tables_host(height, width, host.ctypes.data), "msi_build_trig_tables_host")
            t = torch.from_numpy(host).to(self.device)
            self._trig_cache[key] = t
        return t

    def _planes(self, planes):
        if torch.is_tensor(planes):
            return planes.to(device=self.device, dtype=torch.float32).reshape(-1).contiguous()
        key = tuple(float(p) for p in planes)
        t = self._planes_cache.get(key)
        if t is None:
            t = torch.tensor(key, dtype=torch.float32).to(self.device)
            self._planes_cache[key] = t
        return t

    def load_weights(self, weights):
        """weights: dict TF-variable-name -> array (see nets.variable_shapes)."""
        try:
            cin_w = int(nets._lookup(weights, "conv1_1/weights").shape[2])
        except (KeyError, AttributeError, IndexError):
            cin_w = None
        if cin_w is not None:
            has_coord = cin_w % 2 == 1            # 6 D sweep channels (+ 1 for CoordNet's |sin(lat)| channel)
            if not self._coord_explicit:
                self.coord_net = has_coord
            elif has_coord != self.coord_net:
                raise ValueError("MSI(coord_net=%s) but conv1_1/weights has %d input channels: these are %s weights"
                                 % (self.coord_net, cin_w, "CoordNet (msi_coord_train_net)" if has_coord else "msi_train_net"))
        self._weights = weights
        self._blob_cache.clear()
        self._packed_cache.clear()

    def _net(self, batch, height, width, in_channels, num_outputs, ngf):
        if self._weights is None:
            raise RuntimeError("MSI: no network weights loaded (load_weights / weights=...)")
        key = (batch, height, width, in_channels, num_outputs, ngf, self.coord_net, self.dtype)
        desc = nets.make_desc(batch, height, width, in_channels, num_outputs, ngf, self.coord_net, self.dtype)
        pkey = key[1:]  # packing does not depend on the batch size
        packed = self._packed_cache.get(pkey)
        if packed is None:
            bkey = (in_channels, num_outputs, ngf)
            blob = self._blob_cache.get(bkey)
            if blob is None:
                blob = nets.flatten_params(self._weights, in_channels, num_outputs, ngf, self.coord_net)
                self._blob_cache[bkey] = blob
            packed = torch.from_numpy(nets.pack_params(desc, blob)).to(self.device)
            self._packed_cache[pkey] = packed
        key = key + tuple(sorted(self.net_options.items()))
        pw = self._ws_cache.get(key)
        if pw is None:
            plan = N.NetPlan(desc, self.net_options)      # layer table, tiling and work split resolved once
            ws = torch.empty(plan.workspace_bytes(), dtype=torch.uint8, device=self.device)
            pw = (plan, ws)
            self._ws_cache[key] = pw
        return desc, packed, pw[1]

    def network_status(self):
        """msi_net_plan_status of the LAST network forward of this model: raises MsiError (MSI_E_RANGE) when a LayerNorm
        statistic left the fixed-point window the kernels resolve (mis-scaled or non-finite input; see include/msi_hip.h).
        Synchronises the stream -- call it after a frame, not inside a timed loop.  Returns the status bits (0)."""
        if getattr(self, "_last_forward", None) is None:
            return 0
        plan, ws = self._last_forward
        bits = N.c_int32(0)
        N.check(N.lib.msi_net_plan_status(plan.handle, ws.data_ptr(), self._stream(), N.ctypes.byref(bits)), "msi_net_plan_status")
        return int(bits.value)

    def calibrate(self, net_input, num_outputs=None, ngf=64):
        """msi_net_plan_calibrate: centre every layer's LayerNorm fixed-point window on the raw output this network really produces for `net_input`
        ([B,H,W,Cin], the forward's layout; a representative frame).  The packer derives the windows from the weights alone; a checkpoint whose trained
        gamma / beta / weights break that estimate makes network_status() raise MSI_E_RANGE on every frame -- call this once (the harness does, on the
        first flagged sample) and the windows follow the measurement.  The packed blob is modified in place on the device: every plan of this model
        (any batch size) uses the new windows.  Blocking; returns the number of layers whose window moved.  All or nothing: when it raises (MSI_E_RANGE: a layer
        without finite, non-constant output on this frame) the windows are exactly what they were before the call."""
        b, h, w, cin = net_input.shape
        if num_outputs is None:
            num_outputs = cin // 3          # blend_psv: 6 D -> 2 D
        desc, packed, ws = self._net(b, h, w, cin, num_outputs, ngf)
        want = torch.bfloat16 if self.dtype == 'bf16' else torch.float32
        if net_input.dtype != want or not net_input.is_contiguous():
            net_input = net_input.to(want).contiguous()
        plan = self._plan(b, h, w, cin, num_outputs, ngf)
        changed = N.c_int32(0)
        N.check(N.lib.msi_net_plan_calibrate(plan.handle, packed.data_ptr(), net_input.data_ptr(), ws.data_ptr(), ws.numel(), self._stream(),
                                             N.ctypes.byref(changed)), "msi_net_plan_calibrate")
        return int(changed.value)

    def render_status(self):
        """Status word of the renders since the last call (include/msi_hip.h: MSI_RENDER_STATUS_*): raises ValueError when a
        ray origin -- handed over in DEVICE memory, where the host-side guard cannot look without a sync -- was not inside
        the innermost sphere (the kernel clamped the discriminant: finite pixels, not reference-defined).  Synchronises the
        stream -- call it after a frame, not inside a timed loop -- and resets the word.  Returns 0."""
        bits = int(self._render_status.item())
        if bits:
            self._render_status.zero_()
        if bits & N.RENDER_STATUS_ORIGIN_OUTSIDE:
            raise ValueError("a render's ray origin (pose @ tgt_pos) was not inside the innermost sphere: the reference takes "
                             "sqrt of a negative number there (spherical.py:316-318); the pixels of that call are not defined")
        return bits

    def _plan(self, batch, height, width, in_channels, num_outputs, ngf):
        self._net(batch, height, width, in_channels, num_outputs, ngf)
        key = (batch, height, width, in_channels, num_outputs, ngf, self.coord_net, self.dtype) \
            + tuple(sorted(self.net_options.items()))
        return self._ws_cache[key][0]

    # ------------------------------------------------------------------ msi.py:1196-1217
    def inv_depths(self, start_depth, end_depth, num_depths):
        """num_depths sphere radii uniform in inverse depth, both ends included exactly, far -> near
        (python floats, fp64; tests assert equality with the oracle's restatement of msi.py:1196-1217)."""
        near, far = float(start_depth), float(end_depth)
        n = int(num_depths)
        lo, hi = 1.0 / near, 1.0 / far
        radii = [near, far] + [1.0 / (lo + (hi - lo) * (float(k) / float(n - 1))) for k in range(1, n - 1)]
        return sorted(radii, reverse=True)

    # ------------------------------------------------------------------ msi.py:1163-1194
    def preprocess_image(self, image):
        """uint8 [0,255] or float [0,1] -> float [-1,1] (msi.py:1163-1171)."""
        if not torch.is_tensor(image):
            image = torch.as_tensor(np.asarray(image))
        image = image.to(self.device).contiguous()
        out = torch.empty(image.shape, dtype=torch.float32, device=self.device)
        if image.dtype == torch.uint8:
            N.check(N.lib.msi_preprocess_u8_f32(image.data_ptr(), out.data_ptr(), image.numel(), self._stream()),
                    "msi_preprocess_u8_f32")
        else:
            image = image.to(torch.float32)
            N.check(N.lib.msi_preprocess_f32(image.data_ptr(), out.data_ptr(), image.numel(), self._stream()),
                    "msi_preprocess_f32")
        return out

    def preprocess_image_pair(self, image0, image1):
        """preprocess_image of two uint8 images of one shape in a single launch (raw_src_image / raw_ref_image)."""
        if not (torch.is_tensor(image0) and torch.is_tensor(image1) and image0.dtype == torch.uint8 and
                image1.dtype == torch.uint8 and image0.shape == image1.shape):
            return self.preprocess_image(image0), self.preprocess_image(image1)
        image0, image1 = image0.to(self.device).contiguous(), image1.to(self.device).contiguous()
        out = torch.empty((2,) + tuple(image0.shape), dtype=torch.float32, device=self.device)
        N.check(N.lib.msi_preprocess_pair_u8_f32(image0.data_ptr(), image1.data_ptr(), out[0].data_ptr(), out[1].data_ptr(),
                                                 image0.numel(), self._stream()), "msi_preprocess_pair_u8_f32")
        return out[0], out[1]

    def deprocess_image_and_depth(self, rgb, depth):
        """deprocess_image(rgb), deprocess_depth_image(depth) (test.py:149-159) in a single launch."""
        rgb, depth = self._f32(rgb), self._f32(depth)
        if rgb.shape != depth.shape:
            return self.deprocess_image(rgb), self.deprocess_depth_image(depth)
        out = torch.empty((2,) + tuple(rgb.shape), dtype=torch.uint8, device=self.device)
        N.check(N.lib.msi_deprocess_pair_f32_u8(rgb.data_ptr(), depth.data_ptr(), out[0].data_ptr(), out[1].data_ptr(),
                                                rgb.numel(), self._stream()), "msi_deprocess_pair_f32_u8")
        return out[0], out[1]

    def _deprocess(self, image, is_depth):
        image = self._f32(image)
        out = torch.empty(image.shape, dtype=torch.uint8, device=self.device)
        N.check(N.lib.msi_deprocess_f32_u8(image.data_ptr(), out.data_ptr(), image.numel(), is_depth, self._stream()),
                "msi_deprocess_f32_u8")
        return out

    def deprocess_image(self, image):
        """float [-1,1] -> uint8 (msi.py:1173-1181)."""
        return self._deprocess(image, 0)

    def deprocess_depth_image(self, image):
        """float [0,1] -> uint8 without the (x+1)/2 (msi.py:1186-1194)."""
        return self._deprocess(image, 1)

    def _compose(self, lhs, rhs):
        """[B,4,4] @ [B,4,4] on the device (msi_compose_poses_f32)."""
        lhs, rhs = self._f32(lhs), self._f32(rhs)
        if lhs.dim() == 2:
            lhs = lhs[None]
        if rhs.dim() == 2:
            rhs = rhs[None]
        if rhs.shape[0] != lhs.shape[0]:
            rhs = rhs.expand(lhs.shape[0], 4, 4).contiguous()
        out = torch.empty((lhs.shape[0], 4, 4), dtype=torch.float32, device=self.device)
        N.check(N.lib.msi_compose_poses_f32(lhs.data_ptr(), rhs.data_ptr(), out.data_ptr(), lhs.shape[0], self._stream()),
                "msi_compose_poses_f32")
        return out

    # ------------------------------------------------------------------ msi.py:1094-1130
    def format_network_input(self, ref_image, src_image, ref_pose, src_pose, planes, intrinsics,
                             ref_pose_inv=None, jitter_pose_inv=None, dtype=None):
        """Format the network input into the double sphere-sweep volume.
        Returns net_input [B,H,W,2*3*len(planes)].
        jitter_pose_inv: the hidden graph input `jitter_pose_inv:0` of FLAGS.jitter (msi.py:1118-1120):
        ref_pose_inv <- ref_pose_inv @ jitter_pose_inv.  dtype: 'f32' forces an fp32 volume on a bf16 model."""
        ref_image = self._f32(ref_image)
        src_image = self._f32(src_image)
        b, h, w, c = ref_image.shape
        if c != 3 or src_image.shape != ref_image.shape:
            raise ValueError("format_network_input: images must be [B,H,W,3] and agree")
        # a missing ref_pose_inv is computed on the host (pass it explicitly to avoid the round trip)
        if ref_pose_inv is None:
            ref_pose_inv = torch.linalg.inv(torch.as_tensor(ref_pose, dtype=torch.float32).cpu().double()).float()   # test.py:111
        ref_pose, src_pose, ref_pose_inv = self._f32(ref_pose), self._f32(src_pose), self._f32(ref_pose_inv)
        if jitter_pose_inv is not None:
            ref_pose_inv = self._compose(ref_pose_inv, jitter_pose_inv)          # msi.py:1120
        depths = self._planes(planes)
        nd = depths.numel()
        intr = self._f32(intrinsics)
        trig = self._trig(h, w)
        bf16 = (dtype or self.dtype) == 'bf16'
        psv = torch.empty((b, h, w, 6 * nd), dtype=torch.bfloat16 if bf16 else torch.float32, device=self.device)
        # order = +1 for the reference image (i = 0), -1 for the source (i = 1), msi.py:1127;
        # curr_pose = pose @ ref_pose_inv for both sources in one launch (msi.py:1125)
        if ref_pose_inv.shape[0] != b:
            ref_pose_inv = ref_pose_inv.reshape(-1, 4, 4).expand(b, 4, 4).contiguous()
        cur = torch.empty((2, b, 4, 4), dtype=torch.float32, device=self.device)
        N.check(N.lib.msi_compose_pose_pair_f32(ref_pose.data_ptr(), src_pose.data_ptr(), ref_pose_inv.data_ptr(),
                                                cur[0].data_ptr(), cur[1].data_ptr(), b, self._stream()),
                "msi_compose_pose_pair_f32")
        if self.input_type == 'ODS':
            N.check(N.lib.msi_ods_sweep_volume(ref_image.data_ptr(), src_image.data_ptr(), cur[0].data_ptr(), cur[1].data_ptr(),
                                               intr.data_ptr(), depths.data_ptr(), trig.data_ptr(), b, h, w, nd,
                                               psv.data_ptr(), 1 if bf16 else 0, self._stream()), "msi_ods_sweep_volume")
        else:   # sweep_src for perspective inputs (msi.py:1157-1161); ref_pose_inv = interp_pose_inv (:1113)
            for i, img in enumerate((ref_image, src_image)):
                N.check(N.lib.msi_perspective_plane_sweep_f32(
                    img.data_ptr(), cur[i].data_ptr(), intr.data_ptr(), depths.data_ptr(),
                    b, h, w, nd, psv.data_ptr(), 6 * nd, i * 3 * nd, self._stream()),
                    "msi_perspective_plane_sweep_f32")
        return psv

    # ------------------------------------------------------------------ msi.py:40-289
    def infer_msi(self, raw_src_image, raw_ref_image, raw_hres_src_image, raw_hres_ref_image,
                  ref_pose, src_pose, intrinsics, which_color_pred, num_msi_planes, psv_planes,
                  extra_outputs='', ngf=64, ref_pose_inv=None, jitter_pose_inv=None):
        """Construct and run the MSI inference path.  Returns (pred dict, net_input).
        Note the reference's argument order: src before ref (msi.py:40-46).
        which_color_pred: blend_psv (2D outputs) | blend_bg (2D+3) | blend_bg_psv (3D+3) | alpha_only (D), msi.py:119-275."""
        if which_color_pred not in self.COLOR_SCHEMES:
            raise ValueError("which_color_pred=%r (blend_psv, blend_bg, blend_bg_psv, alpha_only)" % which_color_pred)
        if len(psv_planes) != num_msi_planes:
            # msi.py:138 indexes the src PSV with num_msi_planes
            raise ValueError("infer_msi assumes len(psv_planes) == num_msi_planes (msi.py:138)")
        src_image, ref_image = self.preprocess_image_pair(raw_src_image, raw_ref_image)
        net_input = self.format_network_input(ref_image, src_image, ref_pose, src_pose, psv_planes,
                                              intrinsics, ref_pose_inv=ref_pose_inv, jitter_pose_inv=jitter_pose_inv)
        pred = self.infer_layers(net_input, num_msi_planes, ngf, extra_outputs, which_color_pred)
        return pred, net_input

    def run_net(self, net_input, num_outputs, ngf=64):
        """msi_net(net_input, num_outputs) (msi.py:95-125): [B,H,W,Cin] -> [B,H,W,num_outputs]."""
        b, h, w, cin = net_input.shape
        desc, packed, ws = self._net(b, h, w, cin, num_outputs, ngf)
        want = torch.bfloat16 if self.dtype == 'bf16' else torch.float32
        if net_input.dtype != want or not net_input.is_contiguous():
            net_input = net_input.to(want).contiguous()
        pred = torch.empty((b, h, w, num_outputs), dtype=torch.float32, device=self.device)
        plan = self._plan(b, h, w, cin, num_outputs, ngf)
        N.check(N.lib.msi_net_plan_forward(plan.handle, packed.data_ptr(), net_input.data_ptr(), pred.data_ptr(),
                                           ws.data_ptr(), ws.numel(), self._stream()), "msi_net_plan_forward")
        self._last_forward = (plan, ws)
        return pred

    def infer_layers(self, net_input, num_msi_planes, ngf=64, extra_outputs='', which_color_pred='blend_psv',
                     event_after_convs=None):
        """msi_net + layer_prediction of infer_msi (msi.py:95-147): net_input -> pred dict.  For the reference's default
        colour scheme the 1x1 head, conv8_2's LayerNorm and the RGBA assembly run as ONE fused kernel
        (msi_net_plan_forward_rgba: the tanh prediction never goes to HBM; fp32: bit-identical to the two-step path,
        bf16: the same bf16 operands, fp32 summation order of the head differs);
        everything else takes run_net + assemble_layers.  event_after_convs: torch.cuda.Event (already recorded once, so
        that its handle exists) recorded between the convolutions and the fused tail."""
        b, h, w, cin = net_input.shape
        d = num_msi_planes
        fused = (which_color_pred == 'blend_psv' and
                 net_input.dtype == (torch.float32 if self.dtype == 'f32' else torch.bfloat16) and
                 cin == 6 * d and d % 4 == 0 and d <= 64 and ngf <= 64 and
                 (d <= 32 or d % (8 if self.dtype == 'f32' else 16) == 0) and    # D > 32: two layer groups of whole vectors
                 self.net_options.get(N.NET_OPT_HEAD_FUSE_LN, 1) and net_input.is_contiguous())
        if not fused:
            num_outputs = {'blend_psv': 2 * d, 'blend_bg': 2 * d + 3, 'blend_bg_psv': 3 * d + 3, 'alpha_only': d}[which_color_pred]
            msi_pred = self.run_net(net_input, num_outputs, ngf)
            if event_after_convs is not None:
                event_after_convs.record()
            return self.assemble_layers(net_input, msi_pred, d, extra_outputs, which_color_pred)
        desc, packed, ws = self._net(b, h, w, cin, 2 * d, ngf)
        plan = self._plan(b, h, w, cin, 2 * d, ngf)
        new = lambda: torch.empty((b, h, w, d), dtype=torch.float32, device=self.device)
        rgba = torch.empty((b, d, h, w, 4), dtype=torch.float32, device=self.device)
        bw = new() if 'blend_weights' in extra_outputs else None
        al = new() if 'alpha' in extra_outputs else None
        ev = 0 if event_after_convs is None else event_after_convs.cuda_event
        N.check(N.lib.msi_net_plan_forward_rgba(plan.handle, packed.data_ptr(), net_input.data_ptr(), rgba.data_ptr(),
                                                _ptr(bw), _ptr(al), 0, ws.data_ptr(), ws.numel(), self._stream(), ev),
                "msi_net_plan_forward_rgba")
        self._last_forward = (plan, ws)
        pred = {'rgba_layers': rgba.permute(0, 2, 3, 1, 4)}
        if bw is not None:
            pred['blend_weights'] = bw
        if al is not None:
            pred['alphas'] = al
        if 'psv' in extra_outputs:
            pred['psv'] = net_input
        return pred

    def assemble_layers(self, net_input, msi_pred, num_msi_planes, extra_outputs='', which_color_pred='blend_psv'):
        """layer_prediction of infer_msi (msi.py:130-147, 177-188, 223-242, 258-268; outputs :276-289)."""
        b, h, w, _ = net_input.shape
        d = num_msi_planes
        color = self.COLOR_SCHEMES[which_color_pred]
        new = lambda: torch.empty((b, h, w, d), dtype=torch.float32, device=self.device)
        rgba = torch.empty((b, d, h, w, 4), dtype=torch.float32, device=self.device)
        # msi.py:280-285: blend_weights only for the 'blend' schemes, bg_blend_weights where they exist (blend_bg_psv)
        bw = new() if 'blend_weights' in extra_outputs and 'blend' in which_color_pred else None
        bgw = new() if bw is not None and which_color_pred == 'blend_bg_psv' else None
        al = new() if 'alpha' in extra_outputs else None
        N.check(N.lib.msi_assemble_rgba_color_f32(
            net_input.data_ptr(), 1 if net_input.dtype == torch.bfloat16 else 0, msi_pred.data_ptr(), color,
            rgba.data_ptr(), _ptr(bw), _ptr(al), _ptr(bgw), b, h, w, d, self._stream()), "msi_assemble_rgba_color_f32")
        pred = {'rgba_layers': rgba.permute(0, 2, 3, 1, 4)}
        if bw is not None:
            pred['blend_weights'] = bw
        if bgw is not None:
            pred['bg_blend_weights'] = bgw
        if al is not None:
            pred['alphas'] = al
        if 'psv' in extra_outputs:
            pred['psv'] = net_input
        return pred

    # ------------------------------------------------------------------ msi.py:384-452
    def _native_layers(self, rgba_layers):
        rgba_layers = rgba_layers.to(device=self.device, dtype=torch.float32) if torch.is_tensor(rgba_layers) \
            else self._f32(rgba_layers)
        if rgba_layers.dim() != 5 or rgba_layers.shape[-1] != 4:
            raise ValueError("rgba_layers must be [B,H,W,D,4]")
        native = rgba_layers.permute(0, 3, 1, 2, 4)   # [B,D,H,W,4]
        return native if native.is_contiguous() else native.contiguous()

    @staticmethod
    def _domain_guard(tgt_pos, pose, planes, swap_xz):
        """Host-side only (never a device sync): the ray origin -- tgt_pos with its axes permuted as the ray model does
        (spherical.py:286-288 / :390-392) and taken through the FULL 4x4 pose, translation included (spherical.py:303-310)
        -- must lie inside the innermost sphere, or intersect_sphere takes the square root of a negative number
        (spherical.py:316-318) and the int cast of the NaN pixel coordinate is undefined.  Skipped when either input
        lives on the device: render_kernel clamps a negative discriminant to zero (finite, not reference-defined) and ORs
        MSI_RENDER_STATUS_ORIGIN_OUTSIDE into the model's status word -- render_status() reports it."""
        if any(torch.is_tensor(t) and t.is_cuda for t in (tgt_pos, pose, planes)):
            return
        tp = np.asarray(torch.as_tensor(tgt_pos), dtype=np.float64).reshape(-1, 3)
        c = np.stack([tp[:, 2], tp[:, 1], tp[:, 0]] if swap_xz else [tp[:, 0], tp[:, 1], -tp[:, 2]], axis=1)
        ps = np.asarray(torch.as_tensor(pose), dtype=np.float64).reshape(-1, 4, 4)
        if ps.shape[0] == 1 and c.shape[0] > 1:
            ps = np.repeat(ps, c.shape[0], axis=0)
        if ps.shape[0] != c.shape[0]:
            return      # the batch check that follows reports it
        origin = np.einsum('bij,bj->bi', ps[:, :3, :3], c) + ps[:, :3, 3]
        pl = np.asarray(planes.cpu() if torch.is_tensor(planes) else planes, dtype=np.float64)
        if np.any(np.linalg.norm(origin, axis=1) >= np.abs(pl).min()):
            raise ValueError("the target ray origin (pose @ tgt_pos) must lie inside the innermost sphere (radius %g)"
                             % np.abs(pl).min())

    def _render_args(self, rgba_layers, tgt_pose_rt, tgt_pos, planes):
        native = self._native_layers(rgba_layers)
        b, d, h, w, _ = native.shape
        self._domain_guard(tgt_pos, tgt_pose_rt, planes, swap_xz=True)
        pose = self._f32(tgt_pose_rt).reshape(-1, 4, 4)
        # batch size is taken from tgt_pose_rt in the reference (msi.py:419); broadcast a single pose
        if pose.shape[0] == 1 and b > 1:
            pose = pose.expand(b, 4, 4).contiguous()
        pos = self._f32(tgt_pos).reshape(-1, 3)
        if pose.shape[0] != b or pos.shape[0] != b:
            raise ValueError("tgt_pose_rt / tgt_pos batch must match rgba_layers")
        depths = self._planes(planes)
        if depths.numel() != d:
            raise ValueError("len(planes) != number of layers")
        return native, pose, pos, depths, self._trig(h, w), (b, d, h, w)

    def msi_render_equirect_view(self, rgba_layers, tgt_pose_rt, tgt_pos, planes, intrinsics):
        """Render a target view from an MSI representation -> [B,H,W,3] (msi.py:407-429)."""
        out, _ = self.msi_render_equirect_view_and_depth(rgba_layers, tgt_pose_rt, tgt_pos, planes,
                                                         intrinsics, want_rgb=True, want_depth=False)
        return out

    def msi_render_equirect_depth(self, rgba_layers, tgt_pose_rt, tgt_pos, planes, intrinsics):
        """Composite the normalised layer index instead of colour (msi.py:384-405)."""
        _, out = self.msi_render_equirect_view_and_depth(rgba_layers, tgt_pose_rt, tgt_pos, planes,
                                                         intrinsics, want_rgb=False, want_depth=True)
        return out

    def msi_render_equirect_view_and_depth(self, rgba_layers, tgt_pose_rt, tgt_pos, planes, intrinsics,
                                           want_rgb=True, want_depth=True):
        """Both outputs of test.py:149-159 from ONE warp (the reference recomputes it)."""
        native, pose, pos, depths, trig, (b, d, h, w) = self._render_args(rgba_layers, tgt_pose_rt, tgt_pos, planes)
        rgb = torch.empty((b, h, w, 3), dtype=torch.float32, device=self.device) if want_rgb else None
        dep = torch.empty((b, h, w, 3), dtype=torch.float32, device=self.device) if want_depth else None
        N.check(N.lib.msi_render_equirect_f32(native.data_ptr(), pose.data_ptr(), pos.data_ptr(),
                                              depths.data_ptr(), trig.data_ptr(), b, h, w, d,
                                              _ptr(rgb), _ptr(dep), self._render_status.data_ptr(), self._stream()),
                "msi_render_equirect_f32")
        return rgb, dep

    def msi_render_equirect_view_single(self, rgba_layers, tgt_pose_rt, tgt_pos, planes, intrinsics):
        """Warped, un-composited layers [D,B,H,W,4] (msi.py:431-452)."""
        native, pose, pos, depths, trig, (b, d, h, w) = self._render_args(rgba_layers, tgt_pose_rt, tgt_pos, planes)
        out = torch.empty((d, b, h, w, 4), dtype=torch.float32, device=self.device)
        N.check(N.lib.msi_project_layers_f32(native.data_ptr(), pose.data_ptr(), pos.data_ptr(),
                                             depths.data_ptr(), trig.data_ptr(), b, h, w, d,
                                             out.data_ptr(), self._render_status.data_ptr(), self._stream()), "msi_project_layers_f32")
        return out

    def msi_render_equirect_depth_single(self, rgba_layers, tgt_pose_rt, tgt_pos, planes, intrinsics):
        """msi.py:454-473: the same warped, un-composited layers as msi_render_equirect_view_single (the reference's
        two functions have identical bodies; the caller composites the index channel, test.py:374-382)."""
        return self.msi_render_equirect_view_single(rgba_layers, tgt_pose_rt, tgt_pos, planes, intrinsics)

    # ------------------------------------------------------------------ msi.py:502-525
    def msi_render_ods_view(self, rgba_layers, order, jitter_pose, tgt_pos, planes, intrinsics):
        """Render the left (order=+1) / right (order=-1) ODS eye from an MSI -> [B,H,W,3].
        `tgt_pos` is accepted and unused, as in the reference (spherical.intersect_ods ignores it)."""
        native = self._native_layers(rgba_layers)
        b, d, h, w, _ = native.shape
        pose = self._f32(jitter_pose).reshape(-1, 4, 4)
        if pose.shape[0] == 1 and b > 1:
            pose = pose.expand(b, 4, 4).contiguous()
        intr = self._f32(intrinsics).reshape(-1, 3, 3)
        if pose.shape[0] != b or intr.shape[0] != b:
            raise ValueError("jitter_pose / intrinsics batch must match rgba_layers")
        depths = self._planes(planes)
        if depths.numel() != d:
            raise ValueError("len(planes) != number of layers")
        out = torch.empty((b, h, w, 3), dtype=torch.float32, device=self.device)
        N.check(N.lib.msi_render_ods_f32(native.data_ptr(), pose.data_ptr(), intr.data_ptr(), depths.data_ptr(),
                                         self._trig(h, w).data_ptr(), b, h, w, d, int(order), out.data_ptr(),
                                         self._render_status.data_ptr(), self._stream()), "msi_render_ods_f32")
        return out

    # ------------------------------------------------------------------ msi.py:475-500
    @staticmethod
    def _crop_pose(viewing_window):
        """projector.py:78-86: from_euler([0, vw*pi/2, 0]) = Ry, zero translation; fp32 entries are
        the correctly rounded cos/sin of the fp32 angle."""
        ang = float(np.float32(viewing_window * np.pi / 2.0))
        c, s = np.float32(np.cos(ang)), np.float32(np.sin(ang))
        m = np.eye(4, dtype=np.float32)
        m[0, 0] = c; m[0, 2] = s; m[2, 0] = -s; m[2, 2] = c
        return m

    def msi_render_perspective_view(self, rgba_layers, tgt_pose_rt, tgt_pos, planes, intrinsics,
                                    viewing_window=3, psp_height=270, psp_width=480):
        """Perspective crop of the MSI -> [B,psp_height,psp_width,3].  As in the reference, the
        passed tgt_pose_rt only supplies the batch size: the crop rotation replaces it
        (projector.py:78-86)."""
        native = self._native_layers(rgba_layers)
        b, d, h, w, _ = native.shape
        crop = np.tile(self._crop_pose(viewing_window)[None], (b, 1, 1))
        self._domain_guard(tgt_pos, crop, planes, swap_xz=False)
        pose = self._f32(crop)
        pos = self._f32(tgt_pos).reshape(-1, 3)
        if pos.shape[0] != b:
            raise ValueError("tgt_pos batch must match rgba_layers")
        depths = self._planes(planes)
        if depths.numel() != d:
            raise ValueError("len(planes) != number of layers")
        out = torch.empty((b, psp_height, psp_width, 3), dtype=torch.float32, device=self.device)
        N.check(N.lib.msi_render_perspective_f32(native.data_ptr(), pose.data_ptr(), pos.data_ptr(), depths.data_ptr(),
                                                 b, h, w, d, psp_height, psp_width, out.data_ptr(),
                                                 self._render_status.data_ptr(), self._stream()),
                "msi_render_perspective_f32")
        return out

    # ------------------------------------------------------------------ test.py:283-394
    def msi_render_equirect_hres(self, blend_weights, alphas, raw_hres_ref_image, raw_hres_src_image,
                                 ref_pose, src_pose, tgt_pose_rt, tgt_pos, planes, intrinsics,
                                 ref_pose_inv=None):
        """High-res re-render of test.py:283-394 as ONE fused device pass: the reference loops over
        the planes on the host (one sess.run + numpy composite per plane, to fit its GPU memory); with
        288 GB of HBM the whole high-res sweep volume and layer stack stay resident.
        blend_weights / alphas: the low-res [B,H,W,D] outputs of infer_msi(extra_outputs=
        'blend_weights alphas') (test.py:264-271 saves them as .npy).  Returns (rgb, depth), both
        [B,Hh,Wh,3] float (rgb in [-1,1], depth = composited plane index / D as test.py:374-382)."""
        bw = self._f32(blend_weights)
        al = self._f32(alphas)
        b, h, w, d = bw.shape
        hres_ref = self.preprocess_image(raw_hres_ref_image)
        hres_src = self.preprocess_image(raw_hres_src_image)
        hh, hw = hres_ref.shape[1], hres_ref.shape[2]
        # the high-res volume feeds the fp32 assembly (not the network): fp32 also on a bf16 model
        psv = self.format_network_input(hres_ref, hres_src, ref_pose, src_pose, planes, intrinsics,
                                        ref_pose_inv=ref_pose_inv, dtype='f32')
        low = torch.cat([bw, al], dim=-1).contiguous()
        up = torch.empty((b, hh, hw, 2 * d), dtype=torch.float32, device=self.device)
        N.check(N.lib.msi_resize_bilinear_f32(low.data_ptr(), up.data_ptr(), b, h, w, 2 * d, hh, hw, self._stream()),
                "msi_resize_bilinear_f32")
        rgba = torch.empty((b, d, hh, hw, 4), dtype=torch.float32, device=self.device)
        N.check(N.lib.msi_assemble_rgba_scaled_f32(psv.data_ptr(), up.data_ptr(), rgba.data_ptr(), b, hh, hw, d,
                                                   self._stream()), "msi_assemble_rgba_scaled_f32")
        return self.msi_render_equirect_view_and_depth(rgba.permute(0, 2, 3, 1, 4), tgt_pose_rt, tgt_pos, planes,
                                                       intrinsics)

    # ------------------------------------------------------------------ msi.py:527-548
    def mpi_render_view(self, rgba_layers, tgt_pose, planes, intrinsics, intrinsics_inv=None):
        """Render a target perspective view from plane layers by per-plane inverse homographies
        (zero-padding bilinear) -> [B,H,W,3].  `intrinsics_inv` replaces the hidden graph input
        `intrinsics_inv:0` (homography.py:52); default inverse(intrinsics)."""
        native = self._native_layers(rgba_layers)
        b, d, h, w, _ = native.shape
        pose = self._f32(tgt_pose).reshape(-1, 4, 4)
        intr = self._f32(intrinsics).reshape(-1, 3, 3)
        if intrinsics_inv is None:
            intrinsics_inv = torch.linalg.inv(intr.cpu().double()).float()
        intr_inv = self._f32(intrinsics_inv).reshape(-1, 3, 3)
        if pose.shape[0] != b or intr.shape[0] != b or intr_inv.shape[0] != b:
            raise ValueError("tgt_pose / intrinsics batch must match rgba_layers")
        depths = self._planes(planes)
        if depths.numel() != d:
            raise ValueError("len(planes) != number of layers")
        out = torch.empty((b, h, w, 3), dtype=torch.float32, device=self.device)
        N.check(N.lib.msi_mpi_render_f32(native.data_ptr(), pose.data_ptr(), intr.data_ptr(), intr_inv.data_ptr(),
                                         depths.data_ptr(), b, h, w, d, out.data_ptr(), self._stream()),
                "msi_mpi_render_f32")
        return out
