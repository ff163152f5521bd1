"""ctypes binding of libsdk_hip.so (the C ABI declared in include/sdk_hip.h).

There is no CPU fallback: if the shared library has not been built, or no gfx950 device is
present, every entry point raises.  torch is imported first on purpose - it brings the HIP
runtime (libamdhip64.so.7) into the process, and libsdk_hip.so binds to that same runtime, so
torch tensors' device pointers and torch's streams are valid inside the library.
Exception: SDK_NO_TORCH=1 (the torch-free path, lite.py): torch is NOT imported - the library then binds to the system HIP
runtime and device memory comes from sdk_device_malloc; such a process must never import torch afterwards.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

NO_TORCH = os.environ.get("SDK_NO_TORCH") == "1"      # lite.py sets this attribute (not the environment) when it is what loads the library

_HERE = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("SDK_HIP_LIB", _HERE / "libsdk_hip.so"))


class SdkError(RuntimeError):
    """A libsdk_hip.so call returned non-zero (message from sdk_last_error())."""


class DeviceInfo(C.Structure):
    _fields_ = [("device", C.c_int), ("compute_units", C.c_int), ("clock_khz", C.c_int),
                ("wavefront_size", C.c_int), ("hbm_bytes", C.c_uint64), ("name", C.c_char * 128),
                ("arch", C.c_char * 64)]


class ConvGemmArgs(C.Structure):
    _fields_ = [("A", C.c_void_p), ("lda", C.c_int64), ("W", C.c_void_p),
                ("C", C.c_void_p), ("ldc", C.c_int64), ("C32", C.c_void_p), ("ldc32", C.c_int64),
                ("bias", C.c_void_p), ("scale", C.c_void_p), ("shift", C.c_void_p),
                ("ubias", C.c_void_p), ("ldub", C.c_int64),
                ("X2", C.c_void_p), ("ldx2", C.c_int64), ("S", C.c_void_p), ("lds", C.c_int64),
                ("M", C.c_int), ("N", C.c_int), ("Cin", C.c_int), ("taps", C.c_int), ("dil", C.c_int),
                ("T", C.c_int), ("flags", C.c_uint32), ("stats_mode", C.c_int32), ("stats_part", C.c_void_p),
                ("A2", C.c_void_p), ("lda2", C.c_int64), ("tap_pack", C.c_int32), ("reserved", C.c_int32)]


class EcapaDesc(C.Structure):
    _fields_ = [("n_mels_padded", C.c_int32), ("channels", C.c_int32), ("sub_channels", C.c_int32),
                ("scale", C.c_int32), ("se_channels", C.c_int32), ("attn_channels", C.c_int32),
                ("mfa_channels", C.c_int32), ("embed_dim", C.c_int32), ("n_blocks", C.c_int32),
                ("kernel0", C.c_int32), ("dilation", C.c_int32 * 4), ("precision", C.c_int32), ("blk0_tap_pack", C.c_int32),
                ("off", C.c_int64 * 256)]


class ConvGemmHpArgs(C.Structure):
    """sdk_conv_gemm_hp_args (precise mode: fp16 hi+lo planes)."""
    _fields_ = [("A", C.c_void_p), ("lda", C.c_int64), ("a_lo", C.c_int64), ("W", C.c_void_p),
                ("C", C.c_void_p), ("ldc", C.c_int64), ("c_lo", C.c_int64), ("C32", C.c_void_p), ("ldc32", C.c_int64),
                ("bias", C.c_void_p), ("scale", C.c_void_p), ("shift", C.c_void_p), ("ubias", C.c_void_p), ("ldub", C.c_int64),
                ("X2", C.c_void_p), ("ldx2", C.c_int64), ("x2_lo", C.c_int64), ("S", C.c_void_p), ("lds", C.c_int64), ("s_lo", C.c_int64),
                ("M", C.c_int), ("N", C.c_int), ("Cin", C.c_int), ("taps", C.c_int), ("dil", C.c_int), ("T", C.c_int), ("flags", C.c_uint32)]


class ProfileReport(C.Structure):
    _fields_ = [("launches", C.c_int32 * 24), ("ms", C.c_double * 24), ("flops", C.c_double * 24), ("bytes", C.c_double * 24)]


KERNEL_FAMILIES = ["conv_gemm", "se_gate", "asp_stats", "rows_fc", "asp_pool", "fbank_tile", "fbank_norm", "l2norm",
                   "affinity_coarse", "affinity_rescore", "affinity_rescan", "copy", "affinity_matvec", "conv_gemm256", "asp_fused", "res2net_chain", "resample",
                   "conv_gemm_hp"]

ABI_VERSION = 4
GEMM_RELU = 1
GEMM_TANH = 2
GEMM_A_KBLOCKED = 4      # A read / C written as [cols / 64][M][64] (include/sdk_hip.h)
GEMM_C_KBLOCKED = 8
GEMM_F16 = 16             # fp16 operands and outputs (single-plane fp16 contract)

_vp, _i, _i64, _sz = C.c_void_p, C.c_int, C.c_int64, C.c_size_t

# name -> (restype, argtypes); every symbol include/sdk_hip.h declares must be listed here
SIGNATURES = {
    "sdk_abi_version": (_i, []),
    "sdk_init": (_i, [_i, C.POINTER(_vp)]),
    "sdk_shutdown": (_i, [_vp]),
    "sdk_last_error": (C.c_char_p, []),
    "sdk_get_device_info": (_i, [_vp, C.POINTER(DeviceInfo)]),
    "sdk_set_option": (_i, [_vp, C.c_char_p, _i]),
    "sdk_debug_set_ptr": (_i, [_vp, C.c_char_p, _vp]),
    "sdk_profile_begin": (_i, [_vp]),
    "sdk_profile_end": (_i, [_vp, C.POINTER(ProfileReport)]),
    "sdk_fbank_tables_bytes": (_sz, []),
    "sdk_fbank_tables_fill": (_i, [_vp, _sz]),
    "sdk_fbank_workspace_bytes": (_sz, [_i, _i]),
    "sdk_fbank": (_i, [_vp, _vp, _i, _i, _vp, _vp, _i, _vp, _sz, _vp]),
    "sdk_fbank_fmt": (_i, [_vp, _vp, _i, _i, _vp, _vp, _i, _vp, _sz, _i, _vp]),
    "sdk_fbank_windows_fmt": (_i, [_vp, _vp, _i64, _vp, _i, _i, _vp, _vp, _i, _vp, _sz, _i, _vp]),
    "sdk_fbank_windows": (_i, [_vp, _vp, _i64, _vp, _i, _i, _vp, _vp, _i, _vp, _sz, _vp]),
    "sdk_ingest_create": (_i, [_vp, _i64, _i, _i, C.POINTER(_vp)]),
    "sdk_ingest_destroy": (_i, [_vp]),
    "sdk_ingest_acquire": (_i, [_vp, C.POINTER(_i), C.POINTER(_vp), C.POINTER(_vp)]),
    "sdk_ingest_acquire_sized": (_i, [_vp, _i64, _i, C.POINTER(_i), C.POINTER(_vp), C.POINTER(_vp)]),
    "sdk_ingest_slot_info": (_i, [_vp, _i, C.POINTER(_i64), C.POINTER(_i)]),
    "sdk_ingest_commit": (_i, [_vp, _i, _i64, _i, _i, _vp, C.POINTER(_vp), C.POINTER(_vp)]),
    "sdk_ingest_submit": (_i, [_vp, _vp, _i64, _vp, _i, _i, _vp, C.POINTER(_i), C.POINTER(_vp), C.POINTER(_vp)]),
    "sdk_ingest_release": (_i, [_vp, _i, _vp]),
    "sdk_ingest_copy_ms": (_i, [_vp, _i, C.POINTER(C.c_float), C.POINTER(C.c_double)]),
    "sdk_conv_gemm": (_i, [_vp, C.POINTER(ConvGemmArgs), _vp]),
    "sdk_conv_gemm_hp": (_i, [_vp, C.POINTER(ConvGemmHpArgs), _vp]),
    "sdk_set_gemm_variant": (_i, [_i]),
    "sdk_conv_gemm_stats_bytes": (_sz, [_i, _i, _i]),
    "sdk_conv_gemm_stats_fusable": (_i, [_i, _i, _i]),
    "sdk_colstats_finish": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "sdk_res2net_chain_max_frames": (_i, []),
    "sdk_res2net_chain": (_i, [_vp, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "sdk_se_workspace_bytes": (_sz, [_i, _i, _i]),
    "sdk_se_gate_residual": (_i, [_vp, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "sdk_asp_stats": (_i, [_vp, _vp, _i64, _i, _i, _i, _vp, _vp]),
    "sdk_asp_stats_fmt": (_i, [_vp, _vp, _i64, _i, _i, _i, _vp, _i, _vp]),
    "sdk_rows_fc": (_i, [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _i, _vp]),
    "sdk_asp_pool": (_i, [_vp, _vp, _i64, _vp, _i64, _i, _i, _i, _vp, _vp]),
    "sdk_asp_fused_max_frames": (_i, []),
    "sdk_asp_fused": (_i, [_vp, _vp, _i64, _vp, _vp, _vp, _i64, _i, _i, _i, _i, _vp, _vp]),
    "sdk_asp_kblocked_ok": (_i, [_vp, _i, _i]),
    "sdk_asp_fused_kblocked": (_i, [_vp, _vp, _i64, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "sdk_ecapa_workspace_bytes": (_sz, [C.POINTER(EcapaDesc), _i, _i]),
    "sdk_ecapa_forward": (_i, [_vp, _vp, C.POINTER(EcapaDesc), _vp, _i, _i, _i, _vp, _sz, _vp, _vp]),
    "sdk_ecapa_calib_floats": (_sz, [C.POINTER(EcapaDesc), _i]),
    "sdk_ecapa_forward_calib": (_i, [_vp, _vp, C.POINTER(EcapaDesc), _vp, _i, _i, _i, _vp, _sz, _vp, _vp, _vp]),
    "sdk_xvector_workspace_bytes": (_sz, [_vp, _i, _i]),
    "sdk_xvector_forward": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _sz, _vp, _vp]),
    "sdk_resample_out_len": (_i64, [_i64, _i, _i]),
    "sdk_resample_s16": (_i, [_vp, _vp, _i64, _i, _vp, _i, _i, _i, _vp, _i64, _vp]),
    "sdk_l2norm": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp]),
    "sdk_affinity_workspace_bytes": (_sz, [_i, _i]),
    "sdk_affinity_plan": (_i, [_i, _i, _i, _vp, _vp]),
    "sdk_affinity_plan_range": (_i, [_i, _i, _i, _i, _vp, _vp, _vp]),
    "sdk_affinity_block_plan": (_i, [_i, _i, _i, _i, _vp]),
    "sdk_affinity_block_plan_wave": (_i, [_i, _i, _i, _i, _i, _vp]),
    "sdk_affinity_matvec_workspace_bytes": (_sz, [_i]),
    "sdk_affinity_matvec_plan": (_i, [_i, _i, _i, _vp, _vp]),
    "sdk_affinity_matvec": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _sz, _vp]),
    "sdk_allgather": (_i, [_vp, _vp, _vp, _sz, _vp, _vp]),
    "sdk_allgather_direct": (_i, [_vp, _vp, _vp, _sz, _vp, _vp]),
    "sdk_laplacian_topk_workspace_bytes": (_sz, [_i, _i]),
    "sdk_laplacian_topk": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp, _i, _vp]),
    "sdk_rows_gram_workspace_bytes": (_sz, [_i, _i]),
    "sdk_rows_gram": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _sz, _vp]),
    "sdk_rows_apply": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp]),
    "sdk_rows_unit": (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    "sdk_chol_inverse": (_i, [_vp, _vp, _i, _vp, _vp, _vp]),
    "sdk_kmeans_mindist": (_i, [_vp, _vp, _i, _i, _vp, _vp, _i, _vp]),
    "sdk_kmeans_assign": (_i, [_vp, _vp, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "sdk_affinity_topk": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "sdk_device_malloc": (_i, [_vp, _sz, C.POINTER(C.c_void_p)]),
    "sdk_device_free": (_i, [_vp, _vp]),
    "sdk_memcpy": (_i, [_vp, _vp, _vp, _sz, _i, _vp]),
    "sdk_stream_synchronize": (_i, [_vp, _vp]),
}

_lib = None
_ctx = {}


def load_library() -> C.CDLL:
    """dlopen libsdk_hip.so and type every entry point.  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise SdkError(
            f"{LIB_PATH} not found: the HIP extension has not been built. Run "
            f"`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C {_HERE / 'csrc'}`). "
            "This package has no CPU fallback.")
    if not NO_TORCH:
        import torch  # noqa: F401  (must precede loading libsdk_hip.so, see module docstring)
    lib = C.CDLL(str(LIB_PATH))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    if lib.sdk_abi_version() != ABI_VERSION:
        raise SdkError(f"libsdk_hip.so ABI {lib.sdk_abi_version()} != {ABI_VERSION} expected by this host package")
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load_library().sdk_last_error()
        raise SdkError(f"{what or 'libsdk_hip'} failed (rc={rc}): {msg.decode() if msg else '?'}")


def get_ctx(device: int = 0) -> C.c_void_p:
    """One sdk_ctx per (process, device); raises SdkError when no MI355X is visible."""
    if device not in _ctx:
        lib = load_library()
        h = C.c_void_p()
        check(lib.sdk_init(int(device), C.byref(h)), "sdk_init")
        _ctx[device] = h
    return _ctx[device]


def device_info(device: int = 0) -> dict:
    lib = load_library()
    info = DeviceInfo()
    check(lib.sdk_get_device_info(get_ctx(device), C.byref(info)), "sdk_get_device_info")
    return {"device": info.device, "compute_units": info.compute_units, "clock_khz": info.clock_khz,
            "wavefront_size": info.wavefront_size, "hbm_bytes": int(info.hbm_bytes),
            "name": info.name.decode(), "arch": info.arch.decode()}
