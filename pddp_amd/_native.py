"""ctypes binding of the C ABI in include/pddp_hip.h (libpddp_hip.so).

There is NO fallback: if the HIP library is missing or a call fails, this
module raises.  PyTorch is only used by callers for device memory and streams;
no torch type crosses this boundary (pointers and sizes only).
"""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# (PDDP_HIP_LIB: another build of the same library - the instrumented ones of
# tools/, e.g. -DPDDP_ELEM_MARKS)
LIB_PATH = os.environ.get("PDDP_HIP_LIB") or os.path.join(
    _HERE, "lib", "libpddp_hip.so")
CSRC = os.path.join(_HERE, "csrc")

MAX_AUG, MAX_ACTION, MAX_PARAMS = 8, 4, 8

c_int, c_double, c_void_p = ctypes.c_int, ctypes.c_double, ctypes.c_void_p


class PddpProblem(ctypes.Structure):
    """include/pddp_problem.h `pddp_problem`."""
    _fields_ = [
        ("model", c_int), ("encoding", c_int), ("state_size", c_int),
        ("action_size", c_int), ("encoded_size", c_int), ("aug_size", c_int),
        ("params", c_double * MAX_PARAMS),
        ("Q", c_double * (MAX_AUG * MAX_AUG)),
        ("Q_term", c_double * (MAX_AUG * MAX_AUG)),
        ("R", c_double * (MAX_ACTION * MAX_ACTION)),
        ("x_goal", c_double * MAX_AUG),
        ("u_goal", c_double * MAX_ACTION),
    ]


class RecordLayout(ctypes.Structure):
    """include/pddp_hip.h `pddp_record_layout`."""
    _fields_ = [(k, c_int) for k in (
        "n", "m", "o_Fz", "o_Lzz", "o_Fu", "o_Luz", "o_Lz", "o_Luu", "o_Lu",
        "o_U", "stride", "gain_stride")]


class NativeError(RuntimeError):
    pass


def build(force=False):
    """Compiles libpddp_hip.so for gfx950 with hipcc (no GPU needed)."""
    if force and os.path.exists(LIB_PATH):
        os.remove(LIB_PATH)
    subprocess.check_call(["make", "-s", "-j4", "-C", CSRC])
    return LIB_PATH


_P = c_void_p
_SIGS = {
    "pddp_hip_abi_version": [],
    "pddp_hip_device_count": [],
    "pddp_record_layout_of": [c_int, c_int, _P],
    "pddp_riccati_backward": [c_int] * 4 + [_P] * 4 + [c_int] + [_P] * 4,
    "pddp_riccati_backward_variant": [c_int] * 4 + [_P] * 4 + [c_int] +
                                     [_P] * 4 + [c_int],
    "pddp_riccati_backward_timed": [c_int] * 4 + [_P] * 4 + [c_int] +
                                   [_P] * 4 + [c_int, _P, _P],
    "pddp_boxqp_m1": [c_int] + [_P] * 9,
    "pddp_boxqp": [c_int, c_int] + [_P] * 10,
    "pddp_pack_records": [c_int] * 4 + [_P] * 10,
    "pddp_sum_stage_costs": [c_int, c_int, _P, _P, _P],
    "pddp_nominal_rollout": [_P, c_int, c_int] + [_P] * 7,
    "pddp_derivs": [_P, c_int, c_int] + [_P] * 10,
    "pddp_line_search": [_P, c_int, c_int, c_int] + [_P] * 12,
    "pddp_search_accept": [_P, c_int, c_int, c_int] + [_P] * 11 +
                          [c_double, c_double, c_int] +
                          [_P] * 11,
    "pddp_accept": [c_int] * 5 + [_P] * 5 + [c_double, c_double, c_int] +
                   [_P] * 12,
    "pddp_pack_best": [c_int, c_int, c_int, _P, _P, _P, ctypes.c_longlong, _P,
                       _P],
    "pddp_sweep_nominal": [_P, c_int, c_int] + [_P] * 5 + [c_int] +
                          [_P] * 7,
    "pddp_sweep_nominal_kernel": [c_int],
    "pddp_boxqp_m1_lean_f32": [c_int] + [_P] * 11,
    "pddp_round_nominal_f32": [_P, c_int, c_int, c_int] + [_P] * 5 + [c_int] +
                              [_P] * 9 + [c_double, c_double, c_int] +
                              [_P] * 7 + [c_int, _P, _P],
    "pddp_search_candidates": [c_int],
    "pddp_search_form": [c_int],
    "pddp_bnn_mlp_precision": [c_int],
    "pddp_bnn_mlp_deal": [c_int],
    "pddp_bnn_mlp_f32": [c_int] * 5 + [_P] * 11,
    "pddp_bnn_mlp_rows_f32": [c_int] * 5 + [_P] * 12,
    "pddp_bnn_mlp_f64": [c_int] * 5 + [_P] * 11,
    "pddp_bnn_mlp_rows_f64": [c_int] * 5 + [_P] * 12,
    "pddp_bnn_moment_step_f64": [_P, _P],
    "pddp_bnn_mlp_jvp_rows_f64": [c_int] * 7 + [_P] * 12,
    "pddp_bnn_jvp_features_f64": [_P, _P],
    "pddp_bnn_jvp_moments_f64": [_P, _P],
    "pddp_qr_cost_derivs_f64": [_P, _P],
    "pddp_bnn_moment_step_f32": [_P, _P],
    "pddp_bnn_mlp_jvp_f32": [c_int] * 6 + [_P] * 11,
    "pddp_bnn_mlp_jvp_live_f32": [c_int] * 7 + [_P] * 11,
    "pddp_bnn_mlp_jvp_rows_f32": [c_int] * 7 + [_P] * 12,
    "pddp_bnn_jvp_group": [c_int, c_int],
    "pddp_bnn_jvp_features_f32": [_P, _P],
    "pddp_bnn_jvp_moments_f32": [_P, _P],
    "pddp_qr_cost_derivs_f32": [_P, _P],
    "pddp_gp_step": [_P, c_int, _P, _P, _P, _P, _P, _P],
    "pddp_gp_step_masked": [_P, c_int, _P, _P, _P, _P, _P, _P, c_int, _P],
    "pddp_gp_step_lds_bytes": [c_int] * 6,
    "pddp_gp_rollout": [_P, _P, _P],
    "pddp_event_create": [_P],
    "pddp_event_record": [_P, _P],
    "pddp_event_elapsed_ms": [_P, _P, _P],
    "pddp_event_destroy": [_P],
    "pddp_attach_events": [_P, _P],
}
_TYPED = ("pddp_riccati_backward", "pddp_riccati_backward_variant",
          "pddp_riccati_backward_timed",
          "pddp_boxqp_m1", "pddp_boxqp", "pddp_pack_records", "pddp_sum_stage_costs",
          "pddp_nominal_rollout",
          "pddp_derivs",
          "pddp_line_search", "pddp_search_accept", "pddp_accept",
          "pddp_pack_best", "pddp_sweep_nominal", "pddp_gp_step",
          "pddp_gp_step_masked",
          "pddp_gp_rollout")

_lib = None


def exported_symbols():
    """Every symbol include/pddp_hip.h declares."""
    names = []
    for k in _SIGS:
        if k in _TYPED:
            names += [k + "_f32", k + "_f64"]
        else:
            names.append(k)
    names.append("pddp_hip_arch")
    return names


def lib():
    """Loads the HIP library; raises NativeError when it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeError(
                "libpddp_hip.so is not built (%s). Run "
                "`python -c 'import __graft_entry__ as g; g.build()'` - there "
                "is no CPU fallback." % LIB_PATH)
        l = ctypes.CDLL(LIB_PATH)
        for name, sig in _SIGS.items():
            for full in ([name + "_f32", name + "_f64"]
                         if name in _TYPED else [name]):
                fn = getattr(l, full)
                fn.argtypes = sig
                fn.restype = c_int
        l.pddp_hip_arch.restype = ctypes.c_char_p
        l.pddp_gp_step_lds_bytes.restype = ctypes.c_longlong
        _lib = l
    return _lib


def suffix(dtype):
    if dtype == torch.float32:
        return "f32"
    if dtype == torch.float64:
        return "f64"
    raise NativeError("unsupported dtype %s (f32 / f64 only)" % dtype)


def ptr(t):
    """Device pointer of a tensor (None -> NULL). Must be contiguous."""
    if t is None:
        return None
    if not t.is_contiguous():
        raise NativeError("non-contiguous tensor passed to the C ABI")
    return t.data_ptr()


def stream_handle(device=None):
    return torch.cuda.current_stream(device).cuda_stream


def check(rc, what):
    if rc != 0:
        raise NativeError("%s failed with code %d" % (what, rc))


def call(name, dtype, *args):
    fn = getattr(lib(), "%s_%s" % (name, suffix(dtype)))
    check(fn(*args), name)


class BnnStep(ctypes.Structure):
    """pddp_bnn_step of include/pddp_hip.h."""
    _fields_ = (
        [(k, ctypes.c_int32) for k in ("B", "A", "P", "D", "m", "N", "t",
                                       "n_ang")] +
        [("ang", ctypes.c_int32 * 2), ("n_non", ctypes.c_int32),
         ("non", ctypes.c_int32 * 8), ("in_dim", ctypes.c_int32),
         ("out_dim", ctypes.c_int32)] +
        [(k, ctypes.c_void_p) for k in (
            "Z", "U", "gains", "alphas", "u_min", "u_max", "active",
            "bwd_status", "Q", "Q_term", "R", "x_goal", "u_goal", "X_mean",
            "X_std_inv", "dX_mean", "dX_std", "net_out", "Xp", "F", "Zc", "Uc",
            "J", "Jc", "eps_out", "slot")])


class BnnJvp(ctypes.Structure):
    """pddp_bnn_jvp of include/pddp_hip.h."""
    _fields_ = (
        [(k, ctypes.c_int32) for k in ("B", "P", "D", "m", "N", "t", "n_ang")] +
        [("ang", ctypes.c_int32 * 2), ("n_non", ctypes.c_int32),
         ("non", ctypes.c_int32 * 8), ("in_dim", ctypes.c_int32),
         ("out_dim", ctypes.c_int32)] +
        [(k, ctypes.c_void_p) for k in (
            "Z", "U", "u_min", "u_max", "X_mean", "X_std_inv", "dX_mean",
            "dX_std", "net_out", "Xp", "Xp_next", "eps", "F", "Z_next", "F_z",
            "F_u", "eps_out")] +
        [("independent_noise", ctypes.c_int32), ("slot", ctypes.c_void_p)])


class GpModel(ctypes.Structure):
    """pddp_gp_model of include/pddp_hip.h."""
    _fields_ = (
        [(k, ctypes.c_int32) for k in ("state_size", "action_size", "M",
                                       "encoding", "n_ang", "n_non")] +
        [("ang", ctypes.c_int32 * 4), ("non", ctypes.c_int32 * 8)] +
        [(k, ctypes.c_void_p) for k in ("Xt", "Xt_pairs", "beta", "beta_pairs",
                                        "Kinv", "inv_ell2", "sf2", "sn2")])


class GpRollout(ctypes.Structure):
    """pddp_gp_rollout of include/pddp_hip.h."""
    _fields_ = (
        [(k, ctypes.c_int32) for k in ("B", "N", "A")] +
        [(k, ctypes.c_void_p) for k in (
            "Z", "U", "gains", "alphas", "u_min", "u_max", "active",
            "bwd_status", "Zc", "Uc", "Jc", "Q", "Q_term", "R", "x_goal",
            "u_goal")])


class QrCost(ctypes.Structure):
    """pddp_qr_cost of include/pddp_hip.h."""
    _fields_ = (
        [(k, ctypes.c_int32) for k in ("B", "N", "D", "m", "n_ang")] +
        [("ang", ctypes.c_int32 * 2), ("n_non", ctypes.c_int32),
         ("non", ctypes.c_int32 * 8)] +
        [(k, ctypes.c_void_p) for k in (
            "Z", "U", "u_min", "u_max", "Q", "Q_term", "R", "x_goal", "u_goal",
            "L", "L_z", "L_u", "L_zz", "L_uz", "L_uu")])


E_UNSUPPORTED = -2  # PDDP_E_UNSUPPORTED of include/pddp_hip.h


def call_rc(name, dtype, *args):
    """Like call(), but hands PDDP_E_UNSUPPORTED back to the caller (entry
    points that document it as 'make the separate calls instead')."""
    fn = getattr(lib(), "%s_%s" % (name, suffix(dtype)))
    rc = fn(*args)
    if rc != E_UNSUPPORTED:
        check(rc, name)
    return rc


def record_layout(n, m):
    out = RecordLayout()
    check(lib().pddp_record_layout_of(n, m, ctypes.addressof(out)),
          "pddp_record_layout_of")
    return out


def require_gpu(t):
    if not t.is_cuda:
        raise NativeError(
            "pddp_amd runs its hot path on the MI355X only; got a %s tensor "
            "(no CPU fallback)" % t.device)
