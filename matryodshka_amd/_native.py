"""ctypes binding of libmsi_hip.so (the C ABI declared in include/msi_hip.h).

There is NO fallback: if the shared library is missing or does not export every
symbol of the header, importing this module raises.  The CPU oracle under
`oracle/` is test infrastructure and is never used from here.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char, c_char_p, c_float, c_int32, c_size_t, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmsi_hip.so")

MSI_OK = 0
MSI_NET_NUM_LAYERS = 18
RENDER_STATUS_ORIGIN_OUTSIDE = 1
MSI_ABI_VERSION = 8          # include/msi_hip.h: the version this binding's struct layouts and signatures are written for


class MsiError(RuntimeError):
    pass


class NetDesc(Structure):
    _fields_ = [("batch", c_int32), ("height", c_int32), ("width", c_int32),
                ("in_channels", c_int32), ("num_outputs", c_int32), ("ngf", c_int32),
                ("coord_net", c_int32), ("dtype", c_int32)]


MSI_DTYPE_F32, MSI_DTYPE_BF16 = 0, 1
# msi_net_plan_set_option keys (include/msi_hip.h)
NET_OPT_FIXUP_KERNEL, NET_OPT_TAILSPLIT, NET_OPT_BIGTILE, NET_OPT_HEAD_FUSE_LN, NET_OPT_NUM_CUS = 0, 1, 2, 3, 4
NET_OPT_F32_TILE, NET_OPT_F32_TILE_MASK, NET_OPT_APPLY_AHEAD, NET_OPT_HALO, NET_OPT_HALO_SKIP = 5, 6, 7, 8, 9
NET_OPT_UNIFORM_SPLIT, NET_OPT_BF16_STAGE_RAW, NET_OPT_SPLIT_OVERHEAD, NET_OPT_BF16_WAVES, NET_OPT_F32_SPLIT3 = 10, 11, 12, 13, 14
NET_OPT_F32_SPLIT_F16 = 15
NET_OPT_X3_TILE8 = 16
NET_OPT_X3_ROWPAR = 17
NET_STATUS_F16_SPLIT_RANGE = 8


class LayerInfo(Structure):
    _fields_ = [("name", c_char * 16), ("kind", c_int32), ("cin", c_int32), ("cout", c_int32),
                ("has_coord", c_int32), ("stride", c_int32), ("rate", c_int32),
                ("in_h", c_int32), ("in_w", c_int32), ("out_h", c_int32), ("out_w", c_int32),
                ("param_offset", c_uint64), ("param_floats", c_uint64),
                ("raw_offset", c_uint64), ("affine_offset", c_uint64), ("ln_scale_offset", c_uint64)]


_P = c_void_p
_I = c_int32

# name -> (restype, argtypes); must list every function declared in include/msi_hip.h
SIGNATURES = {
    "msi_version": (c_char_p, []),
    "msi_abi_version": (c_int32, []),
    "msi_last_error_string": (c_char_p, []),
    "msi_crc32c_host": (ctypes.c_uint32, [_P, c_size_t, ctypes.c_uint32]),
    "msi_probe_matrix_rate": (_I, [_I, ctypes.c_int64, _I, _P, _P, _P]),
    "msi_trig_table_floats": (c_size_t, [_I, _I]),
    "msi_build_trig_tables_host": (_I, [_I, _I, _P]),
    "msi_preprocess_u8_f32": (_I, [_P, _P, c_size_t, _P]),
    "msi_preprocess_pair_u8_f32": (_I, [_P, _P, _P, _P, c_size_t, _P]),
    "msi_deprocess_pair_f32_u8": (_I, [_P, _P, _P, _P, c_size_t, _P]),
    "msi_preprocess_f32": (_I, [_P, _P, c_size_t, _P]),
    "msi_deprocess_f32_u8": (_I, [_P, _P, c_size_t, _I, _P]),
    "msi_compose_poses_f32": (_I, [_P, _P, _P, _I, _P]),
    "msi_ods_sphere_sweep_f32": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _I, _I, _P]),
    "msi_ods_sphere_sweep_bf16": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _I, _I, _P]),
    "msi_ods_sweep_volume": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _I, _P]),
    "msi_compose_pose_pair_f32": (_I, [_P, _P, _P, _P, _P, _I, _P]),
    "msi_assemble_rgba_f32": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "msi_assemble_rgba_bf16psv_f32": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "msi_assemble_rgba_color_f32": (_I, [_P, _I, _P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "msi_resize_bilinear_f32": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "msi_assemble_rgba_scaled_f32": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "msi_render_equirect_f32": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P]),
    "msi_project_layers_f32": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P]),
    "msi_render_ods_f32": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P]),
    "msi_render_perspective_f32": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P]),
    "msi_perspective_plane_sweep_f32": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P, _I, _I, _P]),
    "msi_mpi_render_f32": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P]),
    "msi_net_layer_info": (_I, [POINTER(NetDesc), _I, POINTER(LayerInfo)]),
    "msi_net_param_floats": (c_size_t, [POINTER(NetDesc)]),
    "msi_net_packed_floats": (c_size_t, [POINTER(NetDesc)]),
    "msi_net_pack_weights_host": (_I, [POINTER(NetDesc), _P, _P]),
    "msi_net_workspace_bytes": (c_size_t, [POINTER(NetDesc)]),
    "msi_net_plan_create": (_I, [POINTER(NetDesc), POINTER(c_void_p)]),
    "msi_net_plan_destroy": (None, [_P]),
    "msi_net_plan_set_option": (_I, [_P, _I, _I]),
    "msi_net_plan_workspace_bytes": (c_size_t, [_P]),
    "msi_net_plan_layer_is_normalized": (_I, [_P, _I]),
    "msi_net_plan_layer_kernel": (_I, [_P, _I, _P, c_size_t, POINTER(c_int32), POINTER(c_int32)]),
    "msi_net_plan_status": (_I, [_P, _P, _P, POINTER(c_int32)]),
    "msi_net_plan_calibrate": (_I, [_P, _P, _P, _P, c_size_t, _P, POINTER(c_int32)]),
    "msi_net_plan_forward": (_I, [_P, _P, _P, _P, _P, c_size_t, _P]),
    "msi_net_plan_forward_rgba": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, c_size_t, _P, _P]),
    "msi_net_forward_f32": (_I, [POINTER(NetDesc), _P, _P, _P, _P, c_size_t, _P]),
    "msi_net_forward_bf16": (_I, [POINTER(NetDesc), _P, _P, _P, _P, c_size_t, _P]),
}


def _load():
    # torch bundles its own libamdhip64.so.7; load it FIRST so that libmsi_hip.so's
    # NEEDED libamdhip64.so.7 binds to the same runtime instance (two HIP runtimes in
    # one process do not share devices, streams or allocations).
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "matryodshka_amd: %s is missing -- build it with `python -m matryodshka_amd.build` "
            "(or __graft_entry__.build()).  There is no CPU fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise ImportError("matryodshka_amd: %s does not export %s" % (LIB_PATH, name))
        fn.restype = res
        fn.argtypes = args
    if lib.msi_abi_version() != MSI_ABI_VERSION:
        raise ImportError("matryodshka_amd: %s has ABI version %d, this binding is written for %d -- rebuild it "
                          "(python -m matryodshka_amd.build --force)" % (LIB_PATH, lib.msi_abi_version(), MSI_ABI_VERSION))
    return lib


lib = _load()


def last_error():
    return lib.msi_last_error_string().decode("utf-8", "replace")


class NetPlan(object):
    """Owner of a msi_net_plan (host memory of the native library)."""

    def __init__(self, desc, options=None):
        h = c_void_p()
        check(lib.msi_net_plan_create(desc, ctypes.byref(h)), "msi_net_plan_create")
        self.handle = h
        self._destroy = lib.msi_net_plan_destroy     # (module globals may be gone at interpreter shutdown)
        for key, value in (options or {}).items():
            self.set_option(key, value)

    def set_option(self, key, value):
        check(lib.msi_net_plan_set_option(self.handle, int(key), int(value)), "msi_net_plan_set_option")

    def workspace_bytes(self):
        return lib.msi_net_plan_workspace_bytes(self.handle)

    def layer_kernel(self, layer):
        """(kernel instantiation as rocprofv3 spells it, workgroups, tiles cut into K-ranges) of `layer`."""
        buf = ctypes.create_string_buffer(96)
        nb, ns = c_int32(0), c_int32(0)
        check(lib.msi_net_plan_layer_kernel(self.handle, int(layer), ctypes.cast(buf, c_void_p), 96, ctypes.byref(nb), ctypes.byref(ns)),
              "msi_net_plan_layer_kernel")
        return buf.value.decode(), int(nb.value), int(ns.value)

    def kernels(self):
        """[(layer name-less index, kernel, workgroups, split tiles)] for the 18 layers in graph order."""
        return [self.layer_kernel(i) for i in range(18)]

    def __del__(self):
        h, self.handle = getattr(self, "handle", None), None
        if h:
            self._destroy(h)


def check(rc, what):
    if rc != MSI_OK:
        raise MsiError("%s failed (code %d): %s" % (what, rc, last_error()))
