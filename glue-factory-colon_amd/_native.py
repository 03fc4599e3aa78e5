"""ctypes binding of libgfc_amd.so (the C ABI declared in include/gfc_amd.h).

The library is the product: if it is missing or a call fails, this module raises --
there is no PyTorch / CPU fallback for any arithmetic on the path.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int32, c_int64, c_longlong, c_size_t, c_void_p

import torch

_PKG = os.path.dirname(os.path.abspath(__file__))
# GFC_AMD_LIB: another build of the same library (same-box A/B of kernel variants: tools/ab_build.sh)
LIB_PATH = os.environ.get("GFC_AMD_LIB") or os.path.join(_PKG, "libgfc_amd.so")
GFC_LG_MAX_LAYERS = 16
GFC_LG_MAX_RAGGED_PAIRS = 128

STATUS = {0: "GFC_OK", 1: "GFC_ERR_INVALID", 2: "GFC_ERR_WORKSPACE", 3: "GFC_ERR_UNSUPPORTED", 4: "GFC_ERR_LAUNCH"}


class NativeError(RuntimeError):
    pass


class SpParams(Structure):
    _fields_ = [("w", c_void_p * 8), ("bias", c_void_p * 8), ("scale", c_void_p * 8), ("shift", c_void_p * 8),
                ("wh", c_void_p), ("bias_h", c_void_p), ("scale_h", c_void_p), ("shift_h", c_void_p),
                ("wp", c_void_p), ("bias_p", c_void_p), ("scale_p", c_void_p), ("shift_p", c_void_p),
                ("wd", c_void_p), ("bias_d", c_void_p), ("scale_d", c_void_p), ("shift_d", c_void_p),
                ("desc_dim", c_int), ("conv_mode", c_int),
                ("w_wino", c_void_p * 8), ("wh_wino", c_void_p), ("w_stem_wino43", c_void_p)]


class Trace(Structure):
    _fields_ = [("start", POINTER(c_void_p)), ("stop", POINTER(c_void_p)), ("capacity", c_int), ("count", c_int)]


_LG_ARRAYS = ["wqkv", "bqkv", "s_out_w", "s_out_b", "s_ffn0_w", "s_ffn0_b", "s_ln_g", "s_ln_b", "s_ffn3_w",
              "s_ffn3_b", "c_qkv_w", "c_qkv_b", "c_out_w", "c_out_b", "c_ffn0_w", "c_ffn0_b", "c_ln_g", "c_ln_b",
              "c_ffn3_w", "c_ffn3_b"]


class LgParams(Structure):
    _fields_ = ([("n_layers", c_int), ("input_dim", c_int), ("input_proj_w", c_void_p), ("input_proj_b", c_void_p),
                 ("posenc_wr", c_void_p), ("posenc_dim", c_int)]
                + [(n, c_void_p * GFC_LG_MAX_LAYERS) for n in _LG_ARRAYS]
                + [(n, c_void_p * GFC_LG_MAX_LAYERS) for n in ("final_proj_w", "final_proj_b", "matchability_w",
                                                               "matchability_b", "token_w", "token_b")])


_lib = None

# name -> (restype, argtypes); every symbol declared in include/gfc_amd.h
SIGNATURES = {
    "gfc_version": (c_char_p, []),
    "gfc_pack_conv3x3": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "gfc_conv3x3": (c_int, [c_void_p] * 6 + [c_int] * 7 + [c_void_p]),
    "gfc_sp_stem": (c_int, [c_void_p] * 10 + [c_int] * 3 + [c_void_p]),
    "gfc_linear": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p,
                           c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int,
                           c_void_p]),
    "gfc_batched_nt": (c_int, [c_void_p, c_int, c_longlong, c_void_p, c_int, c_longlong, c_void_p, c_int, c_longlong,
                               c_int, c_int, c_int, c_int, c_void_p]),
    "gfc_attention": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int,
                              c_int, c_int, c_float, c_void_p, c_size_t, c_void_p]),
    "gfc_attention_workspace_bytes": (c_size_t, [c_int] * 3),
    "gfc_linear_layernorm_gelu": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p,
                                          c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "gfc_ffn_fused": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                              c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "gfc_layernorm_gelu": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "gfc_sp_workspace_bytes": (c_size_t, [c_int] * 4),
    "gfc_sp_dense": (c_int, [POINTER(SpParams), c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                             c_size_t, POINTER(Trace), c_void_p]),
    "gfc_probe_mfma_peak": (c_int, [c_int, POINTER(c_float), POINTER(c_float), c_void_p]),
    "gfc_event_create": (c_int, [POINTER(c_void_p)]),
    "gfc_event_destroy": (c_int, [c_void_p]),
    "gfc_event_elapsed_ms": (c_int, [c_void_p, c_void_p, POINTER(c_float)]),
    "gfc_sp_detector_head": (c_int, [c_void_p, c_int] + [c_void_p] * 4 + [c_int] * 3 + [c_void_p, c_void_p]),
    "gfc_sp_nms": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "gfc_sp_select_workspace_bytes": (c_size_t, [c_int] * 3),
    "gfc_sp_select": (c_int, [c_void_p, c_int, c_int, c_int, c_float, c_int, c_int, c_void_p, c_void_p, c_void_p,
                              c_void_p, c_size_t, c_void_p]),
    "gfc_sp_nms_select_workspace_bytes": (c_size_t, [c_int] * 3),
    "gfc_sp_nms_select": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_float, c_int, c_int]
                          + [c_void_p] * 5 + [c_size_t, c_void_p]),
    "gfc_sp_sample": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p,
                              c_void_p, c_void_p]),
    "gfc_sp_pad_keypoints": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_int,
                                     c_float, ctypes.c_uint, c_void_p]),
    "gfc_l2norm_rows": (c_int, [c_void_p, c_longlong, c_int, c_void_p]),
    "gfc_lg_workspace_bytes": (c_size_t, [c_int] * 3),
    "gfc_lg_posenc": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int,
                              c_void_p, c_void_p, c_void_p]),
    "gfc_lg_log_assignment": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_size_t,
                                      c_void_p]),
    "gfc_lg_filter_matches": (c_int, [c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p,
                                      c_void_p, c_size_t, c_void_p]),
    "gfc_nn_workspace_bytes": (c_size_t, [c_int] * 3),
    "gfc_nn_match": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_float, c_int] + [c_void_p] * 7
                     + [c_size_t, c_void_p]),
    "gfc_eval_matches_homography": (c_int, [c_void_p] * 5 + [c_int] * 3 + [c_float] * 2 + [c_void_p] * 3),
    "gfc_pack_conv3x3_wino": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "gfc_conv3x3_wino": (c_int, [c_void_p] * 6 + [c_int] * 7 + [c_void_p]),
    "gfc_sp_stem_wino": (c_int, [c_void_p] * 10 + [c_int] * 3 + [c_void_p]),
    "gfc_pack_conv3x3_wino43": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "gfc_sp_stem_wino43": (c_int, [c_void_p] * 10 + [c_int] * 3 + [c_void_p]),
    "gfc_disk_select_workspace_bytes": (c_size_t, [c_int] * 3),
    "gfc_disk_nms_select": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_float, c_int, c_int, c_void_p, c_void_p,
                                    c_void_p, c_void_p, c_size_t, c_void_p]),
    "gfc_disk_gather_descriptors": (c_int, [c_void_p] + [c_int] * 4 + [c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "gfc_disk_gather_descriptors_nhwc": (c_int, [c_void_p] + [c_int] * 4 + [c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "gfc_disk_conv5x5_packed_floats": (c_size_t, [c_int] * 2),
    "gfc_disk_pack_conv5x5": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "gfc_disk_conv5x5": (c_int, [c_void_p] * 7 + [c_int] * 8 + [c_void_p]),
    "gfc_disk_instnorm_workspace_bytes": (c_size_t, [c_int] * 2),
    "gfc_disk_instnorm_stats": (c_int, [c_void_p] + [c_int] * 4 + [c_float, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "gfc_disk_avgpool2": (c_int, [c_void_p] + [c_int] * 5 + [c_void_p, c_void_p]),
    "gfc_disk_upsample2": (c_int, [c_void_p] + [c_int] * 4 + [c_void_p, c_int, c_void_p]),
    "gfc_disk_nchw3_to_nhwc4": (c_int, [c_void_p] + [c_int] * 3 + [c_void_p, c_void_p]),
    "gfc_sp_refine_keypoints": (c_int, [c_void_p] + [c_int] * 3 + [c_void_p] * 2 + [c_int] * 2 + [c_void_p]),
    "gfc_sp_mask_scores": (c_int, [c_void_p] + [c_int] * 3 + [c_void_p] + [c_int] * 2 + [c_void_p] * 2),
    "gfc_sp_filter_keypoints": (c_int, [c_void_p] * 3 + [c_int] * 2 + [c_void_p] + [c_int] * 2 + [c_void_p, c_float, c_void_p]),
    "gfc_eval_homography_dlt": (c_int, [c_void_p] * 6 + [c_int] * 3 + [c_void_p] * 3),
    "gfc_preprocess_resize": (c_int, [c_void_p] + [c_int] * 6 + [c_void_p] + [c_int] * 4 + [c_void_p]),
    "gfc_lg_layer_workspace_bytes": (c_size_t, [c_int]),
    "gfc_lg_layer": (c_int, [POINTER(LgParams), c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int,
                             c_int, c_void_p, c_size_t, c_void_p]),
    "gfc_lg_rowdot": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "gfc_lg_assign_workspace_bytes": (c_size_t, [c_int] * 3),
    "gfc_lg_assign": (c_int, [POINTER(LgParams), c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_float]
                      + [c_void_p] * 6 + [c_size_t, c_void_p]),
    "gfc_lg_packed_workspace_bytes": (c_size_t, [c_int] * 3),
    "gfc_lg_forward_packed": (c_int, [POINTER(LgParams)] + [c_void_p] * 5 + [c_int] * 3 + [c_float] + [c_void_p] * 7
                              + [c_size_t, POINTER(Trace), c_void_p]),
    "gfc_lg_forward": (c_int, [POINTER(LgParams)] + [c_void_p] * 8 + [c_int] * 3 + [c_float] + [c_void_p] * 8
                       + [c_size_t, c_void_p]),
    "gfc_lg_ragged_workspace_bytes": (c_size_t, [c_int, POINTER(c_int32), POINTER(c_int32)]),
    "gfc_lg_forward_ragged": (c_int, [POINTER(LgParams)] + [c_void_p] * 5 + [c_int, POINTER(c_int32), POINTER(c_int32),
                                                                            c_float] + [c_void_p] * 7
                              + [c_size_t, POINTER(Trace), c_void_p]),
}


def lib():
    """Load libgfc_amd.so once; raise loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeError(
                f"{LIB_PATH} not found: build it with `python glue-factory-colon_amd/csrc/build.py` "
                "(or __graft_entry__.build()).  There is no CPU fallback.")
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(status: int, what: str):
    if status != 0:
        raise NativeError(f"{what} failed: {STATUS.get(status, status)}")


def ptr(t):
    """Device pointer of a tensor (None -> NULL).  The tensor must be contiguous fp32/int32/int64."""
    if t is None:
        return None
    assert t.is_contiguous(), "native calls need contiguous tensors"
    return c_void_p(t.data_ptr())


def stream_ptr(device=None):
    return c_void_p(torch.cuda.current_stream(device).cuda_stream)


def require_cuda(t: torch.Tensor, name: str):
    if t.device.type != "cuda":
        raise NativeError(f"{name} is on {t.device}: the MI355X path needs tensors on a 'cuda' (ROCm) device; "
                          "there is no CPU implementation in this package")


class Workspace:
    """Grow-only device scratch buffer (torch-allocated, passed to the library as void*)."""

    def __init__(self):
        self.buf = None

    def get(self, nbytes: int, device):
        if self.buf is None or self.buf.numel() < nbytes or self.buf.device != device:
            self.buf = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
        return self.buf


class KernelTrace:
    """Host-side owner of a gfc_trace: `capacity` hipEvent pairs recorded by the library around
    the launches of the dominant kernel (see include/gfc_amd.h)."""

    def __init__(self, capacity: int):
        l = lib()
        self.capacity = capacity
        self.starts = (c_void_p * capacity)()
        self.stops = (c_void_p * capacity)()
        for arr in (self.starts, self.stops):
            for i in range(capacity):
                ev = c_void_p()
                check(l.gfc_event_create(ctypes.byref(ev)), "gfc_event_create")
                arr[i] = ev
        self.c = Trace(ctypes.cast(self.starts, POINTER(c_void_p)), ctypes.cast(self.stops, POINTER(c_void_p)),
                       capacity, 0)

    def reset(self):
        self.c.count = 0

    def durations_ms(self):
        """Call after a device synchronisation."""
        l = lib()
        out = []
        for i in range(self.c.count):
            ms = c_float()
            check(l.gfc_event_elapsed_ms(self.starts[i], self.stops[i], ctypes.byref(ms)), "gfc_event_elapsed_ms")
            out.append(ms.value)
        return out

    def close(self):
        l = lib()
        for arr in (self.starts, self.stops):
            for i in range(self.capacity):
                if arr[i]:
                    l.gfc_event_destroy(arr[i])
                    arr[i] = None
