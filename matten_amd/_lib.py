"""ctypes binding of libtfc (the C ABI declared in include/tfc.h).

There is no fallback: if the shared library is missing the import of any op fails loudly.
Build it with `python -c "import __graft_entry__ as g; g.build()"` (hipcc, gfx950).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TFC_LIB_PATH") or os.path.join(_HERE, "libtfc.so")  # (TFC_LIB_PATH: kernel experiments, tools/build_variant.sh)


class TfcError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libtfc error {code}: {msg}")
        self.code = code


class Path(C.Structure):
    _fields_ = [
        ("l1", C.c_int32),
        ("l2", C.c_int32),
        ("l3", C.c_int32),
        ("mul", C.c_int32),
        ("x_off", C.c_int32),
        ("y_off", C.c_int32),
        ("out_off", C.c_int32),
        ("w_off", C.c_int32),
        ("coeff", C.c_float),
    ]


class IMap(C.Structure):
    _fields_ = [("d1", C.c_int32), ("d2", C.c_int32), ("s0", C.c_int64), ("s1", C.c_int64), ("s2", C.c_int64)]


class GemmDesc(C.Structure):
    _fields_ = [
        ("m", C.c_int64),
        ("n", C.c_int64),
        ("k", C.c_int64),
        ("a1", C.c_void_p),
        ("a2", C.c_void_p),
        ("b", C.c_void_p),
        ("c", C.c_void_p),
        ("a1_r", IMap),
        ("a1_k", IMap),
        ("a2_r", IMap),
        ("a2_k", IMap),
        ("b_k", IMap),
        ("b_j", IMap),
        ("c_r", IMap),
        ("c_j", IMap),
        ("alpha", C.c_float),
        ("accumulate", C.c_int32),
        ("epilogue", C.c_int32),
        ("aux", C.c_void_p),
        ("act_cst", C.c_float),
        ("workspace", C.c_void_p),
        ("workspace_floats", C.c_int64),
        ("tag", C.c_char_p),
        ("a_rep", C.c_int32),
        ("a1_rep_stride", C.c_int64),
        ("a2_rep_stride", C.c_int64),
    ]


class IrBlock(C.Structure):
    _fields_ = [("off", C.c_int32), ("mul", C.c_int32), ("dim", C.c_int32)]


class LinearIns(C.Structure):
    _fields_ = [
        ("in_off", C.c_int32),
        ("mul_in", C.c_int32),
        ("out_off", C.c_int32),
        ("mul_out", C.c_int32),
        ("dim", C.c_int32),
        ("w_off", C.c_int32),
        ("path_weight", C.c_float),
    ]


class FctpIns(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("x_off", "mul_u", "a_off", "mul_v", "out_off", "mul_w", "dim", "w_off")] + [("alpha", C.c_float)]


class GateSeg(C.Structure):
    _fields_ = [
        ("in_off", C.c_int32),
        ("out_off", C.c_int32),
        ("count", C.c_int32),
        ("dim", C.c_int32),
        ("gate_off", C.c_int32),
        ("act", C.c_int32),
        ("cst", C.c_float),
    ]


EPI_NONE, EPI_SSP, EPI_SSP_GRAD = 0, 1, 2
ACT = {"identity": 0, "ssp": 1, "silu": 2, "tanh": 3, "abs": 4, "sigmoid": 5}

_P = C.c_void_p
_I64 = C.c_int64
_I32 = C.c_int32
_F = C.c_float

# name -> (restype, argtypes); every symbol include/tfc.h declares
SIGNATURES = {
    "tfc_version": (C.c_int, []),
    "tfc_last_error": (C.c_char_p, []),
    "tfc_real_cg": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]),
    "tfc_csr_workspace_bytes": (_I64, [_I64, _I64]),
    "tfc_csr_build": (C.c_int, [_P, _P, _I64, _I64, _P, _P, _P, _P, _P, _P]),
    "tfc_csr_build_noalloc": (C.c_int, [_P, _P, _I64, _I64, _P, _P, _P, _P, _P]),
    "tfc_segment_sum": (C.c_int, [_P, _P, _P, _I64, _I32, _P, _P]),
    "tfc_gather_rows": (C.c_int, [_P, _P, _I64, _I64, _I32, _P, _P]),
    "tfc_nl_count": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _I64, _I64, C.c_double, _P, _P]),
    "tfc_nl_fill": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _I64, _I64, C.c_double, _P, _I64, _P, _P, _P]),
    "tfc_nl_ordering": (C.c_int, [_P, _P, _I64, _I64, _P, _P, _P, _P, _P, _P]),
    "tfc_conv_plan_create": (C.c_int, [C.POINTER(Path), _I32, _I32, _I32, _I32, _I32, C.POINTER(_P)]),
    "tfc_conv_plan_destroy": (C.c_int, [_P]),
    "tfc_conv_workspace_floats": (_I64, [_P, _I64, C.c_int]),
    "tfc_conv_fwd": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _I64, _I64, _F, _P, _P, _P]),
    "tfc_conv_bwd": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _I64, _I64, _F, _P, _P, _P, _P, _P]),
    "tfc_conv_m_signature": (_I64, [_P]),
    "tfc_conv_m_floats": (_I64, [_P, _I64]),
    "tfc_conv_dy_floats": (_I64, [_P, _I64]),
    "tfc_conv_m_build": (C.c_int, [_P, _P, _I64, _P, _P]),
    "tfc_conv_fwd_m": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _I64, _I64, _F, _P, _P]),
    "tfc_conv_fwd_fused_m": (C.c_int, [_P, _P, _P, _P, _P, _I32, _F, _P, _P, _P, _I64, _I64, _F, _P, _P]),
    "tfc_conv_bwd_m": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _I64, _I64, _F, _P, _P, _P, _P, _P]),
    "tfc_conv_plan_set_out_blocks": (C.c_int, [_P, C.POINTER(IrBlock), _I32]),
    "tfc_conv_fwd_m_t": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _I64, _I64, _F, _P, _P, _P]),
    "tfc_conv_bwd_m_t": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _I64, _I64, _F, _P, _P, _P, _P, _P, _P]),
    "tfc_edge_encode_fwd": (C.c_int, [_P, _P, _P, _P, _P, _I64, _I64, _I64, _I32, _I32, _P, _I32, _F, _F, _P, _P, _P, _P, _P]),
    "tfc_edge_bwd_blocks": (_I64, [_I64]),
    "tfc_edge_encode_bwd": (C.c_int, [_P, _P, _I64, _I32, _I32, _P, _I32, _F, _F, _P, _P, _P, _P, _P, _P, _P]),
    "tfc_edge_encode_bwd2": (C.c_int, [_P, _I64, _I32, _I32, _P, _I32, _F, _F, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "tfc_gate_bwd2": (C.c_int, [C.POINTER(GateSeg), _I32, _P, _P, _P, _I64, _I32, _I32, _P, _P, _P]),
    "tfc_edge_vec_bwd": (C.c_int, [_P, _P, _P, _P, _P, _I64, _P, _P]),
    "tfc_edge_cell_bwd": (C.c_int, [_P, _P, _P, _P, _I64, _P, _P]),
    "tfc_gemm": (C.c_int, [C.POINTER(GemmDesc), _P]),
    "tfc_gemm_group": (C.c_int, [C.POINTER(GemmDesc), _I32, _P]),
    "tfc_gemm_workspace_floats": (_I64, [_I64, _I64, _I64]),
    "tfc_radial_mlp_fwd": (C.c_int, [_P, _I64, _I32, C.POINTER(_I32), C.POINTER(_P), _F, C.POINTER(_P), _P, _P]),
    "tfc_radial_mlp_fwd_workspace_floats": (_I64, [_I32, C.POINTER(_I32)]),
    "tfc_radial_mlp_fwd_ws": (C.c_int, [_P, _I64, _I32, C.POINTER(_I32), C.POINTER(_P), _F, C.POINTER(_P), _P, _P, _I64, _P]),
    "tfc_radial_mlp_bwd_workspace_floats": (_I64, [_I64, _I32, C.POINTER(_I32)]),
    "tfc_radial_mlp_bwd": (C.c_int, [_P, _I64, _I32, C.POINTER(_I32), C.POINTER(_P), _F, C.POINTER(_P), _P, C.POINTER(_P), _P,
                                     C.POINTER(_P), _P, _I64, _P]),
    "tfc_linear_workspace_floats": (_I64, [C.POINTER(LinearIns), _I32, _I64]),
    "tfc_linear_fwd": (C.c_int, [C.POINTER(LinearIns), _I32, _P, _P, _I64, _I32, _I32, _I64, _P, _P, _P]),
    "tfc_linear_fwd_t": (C.c_int, [C.POINTER(LinearIns), _I32, _P, _P, _I64, _I32, _I32, _I64, _P, _I32, _P, _P]),
    "tfc_linear_bwd": (C.c_int, [C.POINTER(LinearIns), _I32, _P, _P, _P, _I64, _I32, _I32, _I64, _P, _I32, _P, _P, _P, _I64, _P]),
    "tfc_fctp_workspace_floats": (_I64, [C.POINTER(FctpIns), _I32, _I64]),
    "tfc_fctp_fwd": (C.c_int, [C.POINTER(FctpIns), _I32, _P, _P, _P, _I64, _I32, _I32, _I32, _I64, _I32, _P, _P, _I64, _P]),
    "tfc_fctp_bwd": (C.c_int, [C.POINTER(FctpIns), _I32, _P, _P, _P, _P, _I64, _I32, _I32, _I32, _I64, _P, _I32, _P, _P, _P, _P, _P,
                               _P, _I64, _P]),
    "tfc_ir_transpose": (C.c_int, [_P, _I64, _I32, C.POINTER(IrBlock), _I32, _P, _P]),
    "tfc_gate_fwd": (C.c_int, [C.POINTER(GateSeg), _I32, _P, _I64, _I32, _I32, _P, _P]),
    "tfc_norm_act_fwd": (C.c_int, [C.POINTER(GateSeg), _I32, _P, _I64, _I32, _F, _I32, _P, _P]),
    "tfc_norm_act_bwd": (C.c_int, [C.POINTER(GateSeg), _I32, _P, _P, _I64, _I32, _F, _I32, _P, _P]),
    "tfc_norm_act_bwd2": (C.c_int, [C.POINTER(GateSeg), _I32, _P, _P, _P, _I64, _I32, _F, _I32, _P, _P]),
    "tfc_profile_enable": (C.c_int, [C.c_int]),
    "tfc_profile_reset": (C.c_int, []),
    "tfc_profile_report": (C.c_int, [C.c_char_p, _I64]),
    "tfc_gate_bwd": (C.c_int, [C.POINTER(GateSeg), _I32, _P, _P, _I64, _I32, _I32, _P, _P]),
    "tfc_fctp_class_rows": (C.c_int, [_P, _I32, _I32, _I32, _P, _P, _I64, _P, _P]),
    "tfc_class_table": (C.c_int, [_P, _I64, _I32, _I32, _P, _P, _P, _P, _P, _P]),
    "tfc_fctp_rows_numel": (C.c_int64, [_P, _I32, _I32]),
    "tfc_fctp_rows_w": (C.c_int, [_P, _I32, _P, _I32, _I32, _P, _I64, _P, _P]),
    "tfc_fctp_rows_dw": (C.c_int, [_P, _I32, _P, _I32, _I32, _P, _I64, _P, _P]),
    "tfc_gate_fwd_t": (C.c_int, [C.POINTER(GateSeg), _I32, _P, _I64, _I32, _I32, _P, C.POINTER(IrBlock), _I32, _P, _P]),
    "tfc_gate_bwd_t": (C.c_int, [C.POINTER(GateSeg), _I32, _P, _P, _I64, _I32, _I32, _P, C.POINTER(IrBlock), _I32, _P, _P]),
}

_lib = None


def lib():
    """The loaded library (loaded once).  Raises if libtfc.so has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: the HIP extension has not been built "
                "(run `python -c 'import __graft_entry__ as g; g.build()'`). "
                "matten_amd has no CPU or PyTorch fallback for its kernels."
            )
        h = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(h, name)
            fn.restype = res
            fn.argtypes = args
        _lib = h
    return _lib


def check(code: int):
    if code != 0:
        raise TfcError(code, lib().tfc_last_error().decode("utf-8", "replace"))


def real_cg(l1: int, l2: int, l3: int):
    """Host-only: the library's own Clebsch-Gordan table as a nested list (fp64)."""
    n = (2 * l1 + 1) * (2 * l2 + 1) * (2 * l3 + 1)
    buf = (C.c_double * n)()
    check(lib().tfc_real_cg(l1, l2, l3, buf))
    return list(buf)


def profile_report():
    """{kernel name: (launches, total ms)} from the library's event timers (synchronises)."""
    buf = C.create_string_buffer(1 << 16)
    n = lib().tfc_profile_report(buf, len(buf))
    if n < 0:
        check(n)
    out = {}
    for line in buf.value.decode().splitlines():
        name, cnt, ms = line.split()
        out[name] = (int(cnt), float(ms))
    return out
