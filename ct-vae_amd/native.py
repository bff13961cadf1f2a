"""ctypes binding of ``libctvae_hip.so`` (C ABI: ``include/ctvae_hip.h``).

The product path has NO CPU or eager-PyTorch fallback: if the library is missing, cannot be loaded or
a launch fails, a ``RuntimeError`` is raised (SURVEY.md §8b "Errors").  Tensors are passed as raw device
pointers; the current torch stream is passed as the ``hipStream_t``.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CTVAE_LIB_OVERRIDE") or os.path.join(_HERE, "lib", "libctvae_hip.so")   # override: A/B of two builds in one gpurun call

_c = ctypes
_fp, _vp, _i, _l, _f, _sz = _c.c_void_p, _c.c_void_p, _c.c_int, _c.c_long, _c.c_float, _c.c_size_t

# name -> argtypes (all return int unless listed in _RESTYPES)
SIGNATURES = {
    "ctvae_linear_pixmajor_supported": [_i, _i, _i, _i, _sz],
    "ctvae_linear_pixmajor_forward": [_fp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _fp, _sz, _vp],
    "ctvae_conv_forward": [_i, _fp, _fp, _fp, _fp, _fp] + [_i] * 10 + [_fp, _fp, _i, _fp, _fp, _fp, _sz, _vp],
    "ctvae_wino_filters_batch": [_i, _vp, _vp, _vp, _vp, _vp, _vp],
    "ctvae_conv_bn_act_forward": [_i, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _f, _f, _i, _i, _fp, _fp, _fp, _fp, _fp, _fp] + [_i] * 9
                                 + [_fp, _fp, _i, _fp, _sz, _vp],
    "ctvae_conv_dgrad": [_i, _fp, _fp, _fp, _fp, _i, _fp] + [_i] * 9 + [_fp, _fp, _sz, _vp],
    "ctvae_conv_wgrad": [_i, _fp, _fp, _fp, _fp] + [_i] * 10 + [_fp, _fp, _i, _fp, _fp, _i, _fp, _fp, _fp, _i, _fp, _sz, _vp],
    "ctvae_conv_backward": [_i, _fp, _fp, _fp, _fp, _fp, _fp] + [_i] * 10 + [_fp, _i, _fp, _fp, _fp, _fp, _fp, _fp, _i, _fp, _i,
                            _fp, _fp, _fp, _i, _fp, _fp, _i, _fp, _fp, _i, _fp, _fp, _sz, _vp],
    "ctvae_conv_backward_lazy": [_i, _fp, _fp, _fp, _fp, _fp, _fp] + [_i] * 10 + [_fp, _fp, _i, _i, _fp, _sz, _vp],
    "ctvae_bn_backward_fused": [_fp, _i] + [_i] * 10 + [_fp, _fp, _fp, _fp, _fp, _i, _fp, _fp, _fp, _i, _vp],
    "ctvae_bn_forward": [_fp, _i, _i, _fp, _fp, _fp, _fp, _f, _f, _i, _i, _fp, _fp, _fp, _fp, _fp, _sz, _vp],
    "ctvae_bn_backward": [_fp, _fp, _fp, _i, _i, _fp, _fp, _fp, _i, _fp, _fp, _fp, _i, _fp, _i, _fp, _fp, _fp, _sz, _vp],
    "ctvae_conv_dgrad_bn": [_i, _fp, _fp, _fp, _fp, _i, _fp] + [_i] * 9 + [_fp, _fp, _fp, _fp, _fp, _i, _fp, _i, _fp, _sz, _vp],
    "ctvae_permute": [_fp, _fp, _i, _i, _i, _i, _vp],
    "ctvae_crop_resize_u8": [_fp, _fp, _fp, _i, _i, _i, _i, _i, _i, _vp],
    "ctvae_gat_score": [_i, _fp, _fp, _fp, _fp, _fp, _fp, _i, _i, _i, _i, _f, _vp],
    "ctvae_gat_score_backward": [_fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _i, _i, _i, _i, _f, _vp],
    "ctvae_gat_layer_forward": [_fp, _fp, _i, _fp, _fp, _fp, _fp, _fp, _fp, _i, _fp, _i, _i, _i, _f, _i, _vp],
    "ctvae_gat_layer_backward": [_fp, _fp, _i, _fp, _fp, _fp, _fp, _fp, _fp, _i, _fp, _fp, _fp, _fp, _fp, _fp, _i, _fp, _fp, _fp,
                                 _fp, _i, _i, _i, _i, _f, _i, _vp],
    "ctvae_ct_reg_forward": [_fp, _fp, _fp, _fp, _f, _f, _f, _i, _i, _vp],
    "ctvae_ct_reg_backward": [_fp, _fp, _fp, _fp, _fp, _f, _f, _f, _fp, _fp, _i, _i, _vp],
    "ctvae_ct_blend_softmax_forward": [_fp, _fp, _fp, _l, _i, _i, _vp],
    "ctvae_ct_blend_softmax_backward": [_fp, _fp, _fp, _fp, _fp, _fp, _l, _i, _i, _vp],
    "ctvae_ct_latent_ce_forward": [_fp, _fp, _fp, _l, _i, _vp],
    "ctvae_ct_latent_ce_backward": [_fp, _fp, _fp, _fp, _l, _i, _vp],
    "ctvae_ct_mask_forward": [_fp, _fp, _fp, _fp, _f, _fp, _fp, _fp, _i, _i, _i, _i, _fp, _fp, _fp, _fp, _vp],
    "ctvae_ct_mask_backward": [_fp, _fp, _fp, _fp, _f, _fp, _fp, _fp, _fp, _i, _i, _i, _i, _fp, _fp, _vp],
    "ctvae_ct_blend_forward": [_fp, _fp, _fp, _fp, _l, _vp],
    "ctvae_ct_blend_backward": [_fp, _fp, _fp, _fp, _fp, _fp, _fp, _l, _vp],
    "ctvae_ct_posenc_forward": [_fp, _fp, _fp, _f, _fp, _l, _i, _vp],
    "ctvae_ct_posenc_backward": [_fp, _fp, _f, _fp, _l, _vp],
    "ctvae_one_hot": [_fp, _l, _i, _fp, _vp],
    "ctvae_group_rowsum": [_fp, _l, _i, _i, _i, _i, _fp, _i, _fp, _i, _vp],
    "ctvae_ct_sample_forward": [_fp, _fp, _fp, _fp, _fp, _l, _vp],
    "ctvae_ct_sample_backward": [_fp, _fp, _fp, _fp, _fp, _fp, _l, _vp],
    "ctvae_glinear_forward": [_fp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _fp, _i, _i, _vp],
    "ctvae_glinear_dgrad": [_fp, _i, _i, _i, _vp, _vp, _vp, _vp, _fp, _i, _i, _i, _vp],
    "ctvae_glinear_wgrad": [_fp, _i, _i, _fp, _i, _i, _i, _fp, _i, _i, _fp, _i, _fp, _i, _fp, _sz, _vp],
    "ctvae_pair_mlp_forward": [_fp, _fp, _i, _fp, _fp, _fp, _i, _i, _i, _f, _i, _fp, _vp],
    "ctvae_pair_mlp_backward": [_fp, _fp, _i, _fp, _fp, _fp, _fp, _fp, _i, _fp, _fp, _i, _i, _i, _f, _i, _fp, _vp],
    "ctvae_act_forward": [_fp, _fp, _l, _i, _vp],
    "ctvae_act_backward": [_fp, _fp, _fp, _l, _i, _vp],
    "ctvae_gauss_latent_forward": [_fp, _fp, _fp, _fp, _fp, _i, _i, _fp, _i, _fp, _vp],
    "ctvae_gauss_latent_backward": [_fp, _fp, _fp, _fp, _fp, _fp, _fp, _i, _i, _i, _vp],
    "ctvae_conv_forward_lazy": [_i, _fp, _fp, _fp] + [_i] * 9 + [_fp, _sz, _vp],
    "ctvae_splitk_permute": [_fp, _i, _fp, _i, _i, _i, _vp],
    "ctvae_reparam_forward": [_fp, _l, _fp, _l, _fp, _fp, _i, _i, _vp],
    "ctvae_reparam_backward": [_fp, _fp, _l, _fp, _fp, _fp, _i, _i, _vp],
    "ctvae_loss_forward": [_fp, _fp, _l, _fp, _l, _fp, _l, _i, _i, _f, _fp, _fp, _fp, _sz, _vp],
    "ctvae_loss_forward_grad": [_fp, _fp, _l, _fp, _l, _fp, _l, _i, _i, _f, _fp, _fp, _fp, _fp, _fp, _i, _fp, _sz, _vp],
    "ctvae_mse_backward": [_fp, _fp, _fp, _fp, _l, _i, _vp],
    "ctvae_logcosh_loss_forward": [_fp, _fp, _l, _f, _fp, _l, _fp, _l, _i, _i, _f, _fp, _fp, _sz, _vp],
    "ctvae_logcosh_backward": [_fp, _fp, _fp, _fp, _l, _f, _i, _vp],
    "ctvae_loss_backward": [_fp, _fp, _fp, _fp, _l, _f, _fp, _l, _fp, _l, _fp, _fp, _i, _i, _f, _i, _vp],
    "ctvae_kl_backward": [_fp, _l, _fp, _l, _fp, _fp, _fp, _i, _i, _f, _vp],
    "ctvae_vq_inds": [_fp, _fp, _fp, _i, _i, _i, _i, _i, _vp],
    "ctvae_vq_lookup": [_fp, _fp, _fp, _fp, _fp, _f, _i, _i, _i, _i, _i, _fp, _sz, _vp],
    "ctvae_vq_backward": [_fp, _fp, _fp, _fp, _fp, _fp, _fp, _i, _f, _i, _i, _i, _i, _i, _fp, _sz, _vp],
    "ctvae_gumbel_st_forward": [_fp, _fp, _fp, _fp, _l, _vp],
    "ctvae_gumbel_st_backward": [_fp, _fp, _fp, _fp, _l, _vp],
    "ctvae_gumbel_softmax_forward": [_fp, _fp, _fp, _l, _i, _f, _f, _vp],
    "ctvae_gumbel_softmax_backward": [_fp, _fp, _fp, _l, _i, _f, _vp],
    "ctvae_cat_kl_forward": [_fp, _l, _i, _i, _f, _f, _fp, _fp, _sz, _vp],
    "ctvae_cat_kl_backward": [_fp, _fp, _fp, _l, _i, _i, _f, _f, _vp],
    "ctvae_iw_loss_forward": [_fp, _fp, _l, _i, _i, _fp, _fp, _i, _i, _f, _fp, _fp, _fp, _fp, _vp],
    "ctvae_iw_loss_backward": [_fp, _fp, _l, _i, _i, _fp, _fp, _i, _f, _fp, _fp, _fp, _fp, _fp, _vp],
    "ctvae_l2l1_loss_forward": [_fp, _fp, _l, _fp, _fp, _fp, _sz, _vp],
    "ctvae_l2l1_backward": [_fp, _fp, _fp, _fp, _l, _i, _vp],
    "ctvae_ladder_merge_forward": [_fp, _fp, _fp, _fp, _fp, _i, _i, _fp, _fp, _vp],
    "ctvae_ladder_merge_backward": [_fp, _fp, _fp, _fp, _fp, _fp, _fp, _i, _i, _fp, _fp, _fp, _fp, _vp],
    "ctvae_gamma_reparam_forward": [_fp, _fp, _fp, _f, _fp, _l, _vp],
    "ctvae_gamma_reparam_backward": [_fp, _fp, _fp, _fp, _f, _fp, _fp, _l, _vp],
    "ctvae_gamma_kl_forward": [_fp, _fp, _i, _i, _f, _f, _fp, _fp, _sz, _vp],
    "ctvae_gamma_kl_backward": [_fp, _fp, _fp, _i, _i, _f, _f, _fp, _fp, _vp],
    "ctvae_sigmoid_forward": [_fp, _fp, _l, _vp],
    "ctvae_sigmoid_backward": [_fp, _fp, _fp, _l, _vp],
    "ctvae_tc_forward": [_fp, _fp, _fp, _fp, _i, _i, _fp, _fp, _fp, _fp, _sz, _vp],
    "ctvae_tc_backward": [_fp, _fp, _fp, _fp, _fp, _fp, _fp, _i, _i, _fp, _fp, _fp, _vp],
    "ctvae_vamp_kl_forward": [_fp, _fp, _fp, _fp, _fp, _i, _i, _i, _fp, _fp, _fp, _sz, _vp],
    "ctvae_vamp_kl_backward": [_fp, _fp, _fp, _fp, _fp, _fp, _fp, _i, _i, _i, _fp, _fp, _fp, _fp, _fp, _vp],
    "ctvae_swd_forward": [_fp, _fp, _fp, _i, _i, _i, _f, _f, _fp, _fp, _fp, _sz, _vp],
    "ctvae_mmd_forward": [_fp, _fp, _i, _i, _i, _f, _f, _f, _f, _f, _fp, _fp, _fp, _sz, _vp],
    "ctvae_dip_forward": [_fp, _l, _fp, _l, _i, _i, _f, _f, _fp, _vp],
    "ctvae_dip_backward": [_fp, _fp, _fp, _fp, _i, _i, _vp],
    "ctvae_adam_step": [_fp, _fp, _fp, _fp, _fp, _l, _f, _vp],
    "ctvae_mssim_forward": [_fp, _fp, _fp, _fp, _fp, _fp, _fp, _i, _i, _i, _i, _vp],
    "ctvae_mssim_backward": [_fp, _fp, _fp, _fp, _fp, _fp, _i, _i, _i, _i, _vp],
    "ctvae_defer_begin": [_fp, _sz],
    "ctvae_defer_flush": [_vp],
}
_RESTYPES = {
    "ctvae_version": _c.c_char_p,
    "ctvae_arch": _c.c_char_p,
    "ctvae_error_string": _c.c_char_p,
    "ctvae_workspace_bytes": _c.c_size_t,
    "ctvae_prof_enable": None,
    "ctvae_prof_calibrate": None,
    "ctvae_prof_report": _c.c_size_t,
    "ctvae_conv_dgrad_bn_rows": _c.c_int,
    "ctvae_conv_backward_bn_rows": _c.c_int,
    "ctvae_conv_backward_lazy_slices": _c.c_int,
    "ctvae_conv_forward_lazy_slices": _c.c_int,
    "ctvae_conv_bn_act_apply_is_separate": _c.c_int,
    "ctvae_winograd_enable": _c.c_int,
    "ctvae_conv_wino_filter_floats": _c.c_size_t,
    "ctvae_dip_state_floats": _c.c_size_t,
    "ctvae_adam_state_floats": _c.c_size_t,
    "ctvae_mssim_part_floats": _c.c_size_t,
    "ctvae_glinear_wgrad_ws_bytes": _c.c_size_t,
    "ctvae_conv_input_transform_supported": _c.c_int,
    "ctvae_conv_wgrad_bn_apply_supported": _c.c_int,
}
EXPORTS = sorted(list(SIGNATURES) + list(_RESTYPES))

_lib = None


def load():
    """Load the shared library (once).  Raises RuntimeError when it is absent or does not match the header."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -m ctvae_amd.build` (hipcc, gfx950). "
            "There is no CPU/PyTorch fallback for the ctvae hot path.")
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise RuntimeError(f"cannot load {LIB_PATH}: {e}") from e
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name, None)
        if fn is None:
            raise RuntimeError(f"{LIB_PATH} does not export {name} (stale build?)")
        fn.argtypes = argtypes
        fn.restype = _c.c_int
    for name, res in _RESTYPES.items():
        fn = getattr(lib, name, None)
        if fn is None:
            raise RuntimeError(f"{LIB_PATH} does not export {name} (stale build?)")
        fn.restype = res
        fn.argtypes = {"ctvae_error_string": [_c.c_int], "ctvae_prof_enable": [_c.c_int],
                       "ctvae_prof_calibrate": [_c.c_void_p, _c.c_int],
                       "ctvae_prof_report": [_c.c_char_p, _c.c_size_t],
                       "ctvae_conv_dgrad_bn_rows": [_c.c_int] * 10 + [_c.c_size_t],
                       "ctvae_conv_backward_bn_rows": [_c.c_int] * 10 + [_c.c_size_t],
                       "ctvae_conv_backward_lazy_slices": [_c.c_int] * 11 + [_c.c_size_t],
                       "ctvae_conv_forward_lazy_slices": [_c.c_int] * 10 + [_c.c_size_t],
                       "ctvae_conv_bn_act_apply_is_separate": [_c.c_int] * 10 + [_c.c_size_t],
                       "ctvae_winograd_enable": [_c.c_int],
                       "ctvae_dip_state_floats": [_c.c_int, _c.c_int],
                       "ctvae_glinear_wgrad_ws_bytes": [_c.c_int, _c.c_int, _c.c_int],
                       "ctvae_conv_wino_filter_floats": [_c.c_int] * 10 + [_c.c_size_t],
                       "ctvae_conv_input_transform_supported": [_c.c_int] * 10,
                       "ctvae_conv_wgrad_bn_apply_supported": [_c.c_int] * 10}.get(name, [])
    if lib.ctvae_arch() != b"gfx950":
        raise RuntimeError("libctvae_hip.so was not built for gfx950")
    _lib = lib
    return lib


def check(code: int, what: str):
    if code != 0:
        msg = load().ctvae_error_string(int(code)).decode()
        raise RuntimeError(f"{what} failed: {msg} (code {code})")


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


_workspaces = {}


def workspace(device) -> torch.Tensor:
    """Per-(device, stream) scratch buffer.  Kernels on one stream are ordered, so one buffer per stream is
    race-free; a second stream gets its own."""
    key = (device.index if device.index is not None else torch.cuda.current_device(), torch.cuda.current_stream().cuda_stream)
    ws = _workspaces.get(key)
    if ws is None:
        n = load().ctvae_workspace_bytes()
        ws = torch.empty(n // 4, dtype=torch.float32, device=device)
        _workspaces[key] = ws
    return ws


def call(name: str, *args):
    """Launch entry point `name` on the current stream (the trailing stream argument is appended here)."""
    lib = load()
    check(getattr(lib, name)(*args, stream_ptr()), name)


_wino_on = None


def winograd_enable(on: bool) -> bool:
    """Select Winograd (default) or the direct tap-GEMM kernels for the 3x3 stride-1 layers; returns the old setting."""
    global _wino_on
    prev = bool(load().ctvae_winograd_enable(1 if on else 0))
    _wino_on = bool(on)
    return prev


def winograd_enabled() -> bool:
    global _wino_on
    if _wino_on is None:
        lib = load()
        _wino_on = bool(lib.ctvae_winograd_enable(1))
        lib.ctvae_winograd_enable(1 if _wino_on else 0)
    return _wino_on


def prof_enable(on, detailed: bool = False):
    """on=False: off; on=True: per kernel symbol; detailed=True: names also carry the problem shape."""
    load().ctvae_prof_enable((2 if detailed else 1) if on else 0)


def prof_calibrate(n: int = 64):
    """Log n empty event pairs on the current stream (key "(empty event pair)" of the next prof_report())."""
    load().ctvae_prof_calibrate(stream_ptr(), n)


def prof_report() -> dict:
    """{kernel name: dict(count, ms, flops, bytes)} since the last call (synchronises the recorded events)."""
    lib = load()
    buf = ctypes.create_string_buffer(1 << 16)
    lib.ctvae_prof_report(buf, len(buf))
    out = {}
    for line in buf.value.decode().splitlines():
        name, cnt, ms, fl, by = line.split("\t")
        out[name] = dict(count=int(cnt), ms=float(ms), flops=float(fl), bytes=float(by))
    return out
