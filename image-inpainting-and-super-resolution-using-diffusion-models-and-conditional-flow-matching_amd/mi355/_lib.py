"""ctypes binding of csrc/libmi355_sampler.so (C ABI: include/mi355_sampler.h).

There is NO fallback: if the shared library is missing or cannot be loaded, every product entry
point raises `MI355BackendError`.  Nothing in this package computes on the CPU or through PyTorch ops.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(_HERE), "csrc")
# `make -C csrc` builds csrc/libmi355_sampler.so and installs a copy as <repo>/lib/libmi355_sampler.so: the copy is what gets loaded (a short path,
# a real file: the mapping in /proc/<pid>/maps then names a file of the repository's lib/ directory).  MI355_SAMPLER_LIB selects an experiment variant.
LIB_PATH = os.environ.get("MI355_SAMPLER_LIB") or os.path.join(os.path.dirname(os.path.dirname(_HERE)), "lib", "libmi355_sampler.so")

MI355_F16 = 3   # fp16 storage / MFMAs: the reference's own reduced-precision mode (use_fp16)
MI355_F32, MI355_BF16, MI355_BF16X2 = 0, 1, 2   # BF16X2: bf16 storage / MFMAs, conv + qkv weights as hi + lo bf16 halves (weight rounding removed)
DDPM_PRIOR, DDPM_AMORTIZED, DDPM_REPLACEMENT, DDIM = 0, 1, 2, 3


class MI355BackendError(RuntimeError):
    pass


class DebugConfigC(C.Structure):
    """mi355_debug_config (include/mi355_sampler.h): diagnostic switches, copied into a handle at creation / passed to the test ops."""
    _fields_ = [("conv_ws", C.c_int32), ("conv_small", C.c_int32), ("conv_min_wgs", C.c_int32), ("conv_stagger", C.c_int32),
                ("conv_ablate", C.c_int32), ("conv_spin_limit", C.c_int32), ("conv_time_reps", C.c_int32), ("gn_apply_max_hw", C.c_int32),
                ("gn_fuse", C.c_int32), ("l2_warm", C.c_int32), ("attn_fused", C.c_int32), ("gn_epilogue", C.c_int32), ("conv_pp", C.c_int32), ("conv_edge", C.c_int32), ("sampler_graph", C.c_int32), ("reserved", C.c_int32 * 1)]


# experiment scripts (tools/*.sh) set these; the LIBRARY reads no environment variable - the Python binding turns them into the struct
_DEBUG_ENV = {"MI355_CONV_WS": "conv_ws", "MI355_CONV_SMALL": "conv_small", "MI355_CONV_MINWG": "conv_min_wgs", "MI355_CONV_STAGGER": "conv_stagger",
              "MI355_CONV_ABLATE": "conv_ablate", "MI355_CONV_SPIN": "conv_spin_limit", "MI355_CONV_TIME": "conv_time_reps",
              "MI355_GN_APPLY_MAXHW": "gn_apply_max_hw", "MI355_GN_FUSE": "gn_fuse", "MI355_L2_WARM": "l2_warm", "MI355_ATTN_FUSE": "attn_fused",
              "MI355_GN_EPILOGUE": "gn_epilogue", "MI355_CONV_PP": "conv_pp", "MI355_CONV_EDGE": "conv_edge", "MI355_SAMPLER_GRAPH": "sampler_graph"}


def debug_config(**overrides) -> "DebugConfigC":
    """Defaults from the library, then MI355_* environment variables (experiments), then keyword overrides (tests)."""
    d = DebugConfigC()
    lib().mi355_debug_defaults(C.byref(d))
    for env, field in _DEBUG_ENV.items():
        if os.environ.get(env, "") != "":
            setattr(d, field, int(os.environ[env]))
    for k, v in overrides.items():
        if k not in dict((f[0], 1) for f in DebugConfigC._fields_):
            raise KeyError(k)
        setattr(d, k, int(v))
    return d


class UNetConfigC(C.Structure):
    _fields_ = [
        ("image_size", C.c_int32), ("in_channels", C.c_int32), ("model_channels", C.c_int32),
        ("out_channels", C.c_int32), ("num_res_blocks", C.c_int32), ("n_attention_ds", C.c_int32),
        ("attention_ds", C.c_int32 * 8), ("n_channel_mult", C.c_int32), ("channel_mult", C.c_int32 * 8),
        ("conv_resample", C.c_int32), ("num_heads", C.c_int32), ("num_head_channels", C.c_int32),
        ("num_heads_upsample", C.c_int32), ("use_scale_shift_norm", C.c_int32), ("resblock_updown", C.c_int32),
        ("use_new_attention_order", C.c_int32), ("dtype", C.c_int32), ("differentiable", C.c_int32),
        ("debug", C.POINTER(DebugConfigC)),
    ]


class ConvExtrasC(C.Structure):
    """mi355_conv_extras (include/mi355_sampler.h): the fused forms of the small-level conv, for mi355_conv2d_ex."""
    _fields_ = [("skip_x0", C.c_void_p), ("skip_x1", C.c_void_p), ("skip_c0", C.c_int32), ("skip_c1", C.c_int32),
                ("skip_w_host", C.c_void_p), ("skip_bias_host", C.c_void_p),
                ("act_out", C.c_void_p * 2), ("act_gamma", C.c_void_p * 2), ("act_beta", C.c_void_p * 2),
                ("act_ctotal", C.c_int32 * 2), ("act_coff", C.c_int32 * 2), ("act_silu", C.c_int32 * 2),
                ("act_film", C.c_void_p), ("act_done", C.c_int32), ("skip_done", C.c_int32)]


class UNetStatsC(C.Structure):
    _fields_ = [("launches", C.c_int64), ("conv_flops", C.c_double), ("attn_flops", C.c_double),
                ("act_bytes", C.c_double), ("weight_bytes", C.c_double)]


class OpProfileC(C.Structure):
    _fields_ = [("kind", C.c_int32), ("ks", C.c_int32), ("cin", C.c_int32), ("cout", C.c_int32), ("h", C.c_int32),
                ("w", C.c_int32), ("tile_m", C.c_int32), ("tile_n", C.c_int32), ("ms", C.c_float), ("flops", C.c_double),
                ("bytes", C.c_double)]


_FP = C.POINTER(C.c_float)


class DDPMTablesC(C.Structure):
    _fields_ = [("Ns", C.c_int32)] + [(n, _FP) for n in (
        "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod", "posterior_mean_coef1", "posterior_mean_coef2",
        "posterior_log_variance_clipped", "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod",
        "recip_sqrt_m1_alphas_cumprod", "alphas_cumprod_prev")]


class DDPMOptionsC(C.Structure):
    _fields_ = [("mode", C.c_int32), ("n_corrector", C.c_int32), ("delta", C.c_float), ("tmin", C.c_float),
                ("tmax", C.c_float), ("start_fraction", C.c_float), ("noise_condition", C.c_int32),
                ("pad_value", C.c_float), ("none_value", C.c_float), ("cond_is_none", C.c_int32), ("seed", C.c_uint64)]


_VP, _I, _I64, _F, _U64 = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_uint64

# name -> (restype, argtypes); this table is also what tests/test_abi.py checks against the header
SIGNATURES = {
    "mi355_version": (_I, []),
    "mi355_last_error": (C.c_char_p, []),
    "mi355_debug_defaults": (None, [C.POINTER(DebugConfigC)]),
    "mi355_unet_status": (_I, [_VP, _I]),
    "mi355_unet_param_count": (_I, [C.POINTER(UNetConfigC)]),
    "mi355_unet_param_info": (_I, [C.POINTER(UNetConfigC), _I, C.c_char_p, _I, C.POINTER(C.c_int64), C.POINTER(C.c_int)]),
    "mi355_unet_weight_bytes": (_I64, [C.POINTER(UNetConfigC)]),
    "mi355_unet_create": (_I, [C.POINTER(UNetConfigC), C.POINTER(_VP), _I, _VP, _I64, _VP, C.POINTER(_VP)]),
    "mi355_unet_destroy": (None, [_VP]),
    "mi355_unet_workspace_bytes": (_I64, [_VP, _I]),
    "mi355_unet_forward": (_I, [_VP, _VP, _I, _VP, _I, _VP, _VP, _I, _VP, _I64, _VP]),
    "mi355_unet_forward_t": (_I, [_VP, _VP, _I, _VP, _I, _F, _VP, _I, _VP, _I64, _VP]),
    "mi355_unet_vjp": (_I, [_VP, _VP, _VP, _I, _I, _VP, _I64, _VP]),
    "mi355_unet_plan_op": (_I, [_VP, _I, C.POINTER(C.c_int32)]),
    "mi355_unet_read_tensor": (_I, [_VP, _I, _I, _VP, _I, _VP, _I64, _VP]),
    "mi355_unet_get_stats": (_I, [_VP, _I, C.POINTER(UNetStatsC)]),
    "mi355_unet_profile": (_I, [_VP, _VP, _I, _VP, _I, _VP, _VP, _I, _VP, _I64, _VP, C.POINTER(OpProfileC), _I]),
    "mi355_cfm_euler_sample": (_I, [_VP, _VP, _I, _VP, _I, _I, _FP, _I, _VP, _VP, _I, _VP, _I64, _VP]),
    "mi355_ddpm_sample": (_I, [_VP, _VP, _I, _VP, C.POINTER(DDPMTablesC), C.POINTER(DDPMOptionsC), _VP, _I64, _I, _VP, _I64, _VP]),
    "mi355_timestep_embedding": (_I, [_VP, _I, _I, _F, _VP, _VP]),
    "mi355_groupnorm": (_I, [_VP, _VP, _VP, _VP, _I, _I, _I, _I, _F, _I, _VP]),
    "mi355_euler_step": (_I, [_VP, _VP, _F, _I64, _VP]),
    "mi355_ddpm_step": (_I, [_VP, _VP, _VP, _F, _F, _F, _F, _F, _I, _U64, _U64, _I64, _VP]),
    "mi355_corrector_step": (_I, [_VP, _VP, _VP, _F, _F, _F, _F, _F, _I, _U64, _U64, _I64, _VP]),
    "mi355_ddim_step": (_I, [_VP, _VP, _F, _F, _F, _I64, _VP]),
    "mi355_replace_mask": (_I, [_VP, _VP, _VP, _F, _I, _F, _F, _I, _U64, _U64, _I64, _VP]),
    "mi355_guidance_seed": (_I, [_VP, _VP, _VP, _F, _F, _I, _F, _I64, _VP, _VP, _I64, _VP]),
    "mi355_guidance_update": (_I, [_VP, _VP, _VP, _F, _I, _VP, _I64, _VP]),
    "mi355_clip": (_I, [_VP, _F, _F, _I64, _VP]),
    "mi355_ema_update": (_I, [_VP, _VP, _F, _F, _I64, _VP]),
    "mi355_mse_per_sample": (_I, [_VP, _VP, _VP, C.c_int, _I64, _VP]),
    "mi355_lincomb_per_sample": (_I, [_VP, _VP, _VP, _VP, _VP, _I, _I64, _VP]),
    "mi355_resize_bilinear": (_I, [_VP, _VP, _I64, _I, _I, _I, _I, _VP]),
    "mi355_paint_patch": (_I, [_VP, _VP, _VP, _I, _F, _I, _VP, _I, _I, _I, _I, _VP]),
    "mi355_quantize_u8": (_I, [_VP, _VP, _I64, _VP]),
    "mi355_to_unit_range": (_I, [_VP, _VP, _I64, _VP]),
    "mi355_randn": (_I, [_VP, _U64, _U64, _I64, _VP]),
    "mi355_rk_combine": (_I, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _FP, _I, _I64, _VP]),
    "mi355_rk_sqnorm": (_I, [_VP, _VP, _VP, _VP, _F, _F, _I64, _VP, _VP]),
    "mi355_rk_interp": (_I, [_VP, _VP, _VP, _VP, _VP, _VP, _F, _F, _I64, _VP]),
    "mi355_op_workspace_bytes": (_I64, [_I, _I, _I]),
    "mi355_box_probe_workspace_bytes": (_I64, []),
    "mi355_box_probe": (_I, [_I, _VP, _I64, _VP, _FP, _FP, _FP]),
    "mi355_box_probe_hbm_workspace_bytes": (_I64, []),
    "mi355_box_probe_hbm": (_I, [_I, _VP, _I64, _VP, _FP, _FP]),
    "mi355_conv2d": (_I, [_VP, _VP, _I, _FP, _FP, _VP, _I, _I, _I, _I, _I, _I, _I, _I, _VP, _VP, _I, _VP, _VP, _I, _I, C.POINTER(DebugConfigC), _VP,
                          _I64, _VP]),
    "mi355_conv2d_ex": (_I, [_VP, _VP, _I, _FP, _FP, _VP, _I, _I, _I, _I, _I, _I, _I, _I, _VP, _VP, _I, _VP, _VP, _I, _I, C.POINTER(DebugConfigC), _VP,
                             _I64, _VP, C.POINTER(ConvExtrasC)]),
    "mi355_qkv_attention": (_I, [_VP, _VP, _I, _I, _I, _I, _I, _I, _VP, _I64, _VP]),
}

_lock = threading.Lock()
_lib = None


def build(force: bool = False) -> str:
    """Compile the HIP extension in-tree for gfx950 (cross-compiles without a GPU)."""
    if force:
        subprocess.run(["make", "-C", CSRC, "clean"], check=True, capture_output=True)
    r = subprocess.run(["make", "-C", CSRC, "-j8"], capture_output=True, text=True)
    if r.returncode != 0:
        raise MI355BackendError("building libmi355_sampler.so failed:\n" + r.stdout[-4000:] + r.stderr[-4000:])
    return LIB_PATH


def lib():
    """Load the shared library (once) and type every entry point.  Raises if it is not built."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise MI355BackendError(
                f"{LIB_PATH} is missing: build it with `make -C {CSRC}` (or `python -c 'import __graft_entry__ as g; g.build()'`). "
                "This package has no CPU or PyTorch fallback.")
        # PyTorch-ROCm ships its own libamdhip64; the process must use ONE HIP runtime for the device pointers and streams torch
        # hands over to mean anything, so torch's copy has to be resident before this library's dependency is resolved
        # (loading the system runtime first and torch afterwards ends in "no ROCm-capable device is detected").
        import torch  # noqa: F401

        try:
            L = C.CDLL(LIB_PATH)
        except OSError as e:  # pragma: no cover
            raise MI355BackendError(f"cannot load {LIB_PATH}: {e}") from e
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
        return L


def check(rc: int, what: str = ""):
    if rc is not None and rc < 0:
        msg = lib().mi355_last_error().decode("utf-8", "replace")
        raise MI355BackendError(f"{what or 'libmi355_sampler'} failed ({rc}): {msg}")
    return rc


def make_config(*, image_size, in_channels, model_channels, out_channels, num_res_blocks, attention_ds, channel_mult,
                conv_resample=True, num_heads=1, num_head_channels=-1, num_heads_upsample=-1, use_scale_shift_norm=False,
                resblock_updown=False, use_new_attention_order=False, dtype=MI355_BF16, differentiable=False, debug=None) -> UNetConfigC:
    c = UNetConfigC()
    c._debug_keepalive = debug if debug is not None else debug_config()   # the struct must outlive the pointer (read at creation)
    c.debug = C.pointer(c._debug_keepalive)
    c.image_size, c.in_channels, c.model_channels, c.out_channels = image_size, in_channels, model_channels, out_channels
    c.num_res_blocks = num_res_blocks
    attention_ds = list(attention_ds)
    channel_mult = list(channel_mult)
    if len(attention_ds) > 8 or len(channel_mult) > 8:
        raise ValueError("at most 8 attention resolutions / channel multipliers")
    for m in channel_mult:
        if int(m) != m:
            raise ValueError("only integer channel multipliers are supported by the HIP plan")
    c.n_attention_ds = len(attention_ds)
    for i, v in enumerate(attention_ds):
        c.attention_ds[i] = int(v)
    c.n_channel_mult = len(channel_mult)
    for i, v in enumerate(channel_mult):
        c.channel_mult[i] = int(v)
    c.conv_resample, c.num_heads, c.num_head_channels = int(conv_resample), num_heads, num_head_channels
    c.num_heads_upsample = num_heads_upsample
    c.use_scale_shift_norm, c.resblock_updown = int(use_scale_shift_norm), int(resblock_updown)
    c.use_new_attention_order, c.dtype = int(use_new_attention_order), dtype
    c.differentiable = int(bool(differentiable))
    return c
