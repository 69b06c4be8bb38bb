"""ctypes loader for the CPU oracle (oracle/liblfbm5d_oracle.so) and the compiled-reference leaf
pins (oracle/_ref/libref_leaf.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (lfbm5d_amd) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

YUV, YCBCR, OPP, RGB, ID, DCT, SADCT, BIOR, HADAMARD, HAAR = range(10)
ROWMAJOR, COLMAJOR = 11, 12
TAU = {"id": ID, "dct": DCT, "sadct": SADCT, "bior": BIOR, "hw": HADAMARD, "haar": HAAR}
CS = {"yuv": YUV, "ycbcr": YCBCR, "opp": OPP, "rgb": RGB}


class Params(C.Structure):
    _fields_ = [("sigma", C.c_float), ("lambda_", C.c_float), ("N", C.c_uint), ("nSim", C.c_uint),
                ("nDisp", C.c_uint), ("k", C.c_uint), ("p", C.c_uint), ("useSD", C.c_uint),
                ("tau_2D", C.c_uint), ("tau_4D", C.c_uint), ("tau_5D", C.c_uint),
                ("color_space", C.c_uint)]


class Stats(C.Structure):
    _fields_ = [("groups", C.c_ulonglong), ("sadct_groups", C.c_ulonglong),
                ("stack_patches", C.c_ulonglong), ("windows", C.c_ulonglong),
                ("passes", C.c_ulonglong), ("bm_seconds", C.c_double), ("total_seconds", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


def build(force=False):
    so = os.path.join(_HERE, "liblfbm5d_oracle.so")
    src = os.path.join(_HERE, "lfbm5d_oracle.cpp")
    stale = (not os.path.exists(so)) or os.path.getmtime(so) < os.path.getmtime(src)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "liblfbm5d_oracle.so"], stdout=subprocess.DEVNULL)
    if os.path.exists("/root/reference/src/lib_transforms.cpp") and (
            force or not os.path.exists(os.path.join(_HERE, "_ref", "libref_leaf.so"))):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)
    return so


_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
_lib = None
_ref = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    L = C.CDLL(build())
    L.orc_haar_forward.argtypes = [_f32p, C.c_uint]
    L.orc_haar_inverse.argtypes = [_f32p, C.c_uint]
    L.orc_hadamard.argtypes = [_f32p, C.c_uint]
    L.orc_bior_forward.argtypes = [_f32p, C.c_uint, _f32p, C.c_uint]
    L.orc_bior_inverse.argtypes = [_f32p, C.c_uint]
    L.orc_redft10.argtypes = [_f32p, _f32p, C.c_uint]
    L.orc_redft01.argtypes = [_f32p, _f32p, C.c_uint]
    L.orc_dct2d_forward.argtypes = [_f32p, C.c_uint, _f32p, C.c_uint]
    L.orc_dct2d_inverse.argtypes = [_f32p, C.c_uint]
    L.orc_dct4d_forward.argtypes = [_f32p, C.c_uint, C.c_uint]
    L.orc_dct4d_inverse.argtypes = [_f32p, C.c_uint, C.c_uint]
    L.orc_sadct_forward.argtypes = [_f32p, _u32p, C.c_uint, C.c_uint, _u32p]
    L.orc_sadct_inverse.argtypes = [_f32p, _u32p, C.c_uint, C.c_uint]
    L.orc_kaiser_window.argtypes = [_f32p, C.c_uint]
    L.orc_bm_self.argtypes = [_f32p, C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.c_uint,
                              C.c_float, _u32p, C.c_uint, _u32p, _u32p]
    L.orc_bm_stereo.argtypes = [_f32p, _f32p, C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.c_float,
                                _u32p, _u8p]
    L.orc_pass.argtypes = [C.c_int, C.POINTER(Params), C.c_uint, C.c_uint, C.c_uint, C.c_uint,
                           C.c_uint, _f32p, C.c_void_p, _f32p, _f32p, _u32p, _u32p, C.c_uint,
                           C.c_uint, C.c_int, C.c_int, C.POINTER(Stats)]
    L.orc_run_step1.argtypes = [C.POINTER(Params), _f32p, _u32p, _f32p, C.c_uint, C.c_uint, C.c_uint,
                                C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.c_int, C.POINTER(Stats)]
    L.orc_run_step2.argtypes = [C.POINTER(Params), _f32p, _u32p, _f32p, _f32p, C.c_uint, C.c_uint,
                                C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.c_int,
                                C.POINTER(Stats)]
    L.orc_bm3d_step.argtypes = [C.c_int, C.c_float, C.c_float, _f32p, C.c_void_p, _f32p] + [C.c_uint] * 10 + [C.POINTER(Stats)]
    L.orc_run_bm3d_lf.argtypes = [C.c_float, _f32p, _u32p, _f32p, _f32p] + [C.c_uint] * 16 + [C.c_float, C.c_uint, C.POINTER(Stats)]
    L.orc_mt_seed.argtypes = [C.c_ulong]
    L.orc_mt_int32.restype = C.c_ulong
    L.orc_mt_res53.restype = C.c_double
    L.orc_add_noise.argtypes = [_f32p, _f32p, C.c_ulonglong, C.c_float]
    L.orc_symetrize.argtypes = [_f32p, _f32p, C.c_uint, C.c_uint, C.c_uint, C.c_uint]
    L.orc_unsymetrize.argtypes = [_f32p, _f32p, C.c_uint, C.c_uint, C.c_uint, C.c_uint]
    L.orc_color_transform.argtypes = [_f32p, C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.c_int]
    L.orc_sigma_table.argtypes = [C.c_float, C.c_uint, C.c_uint, _f32p]
    L.orc_last_weights.argtypes = [_f32p, C.c_uint]
    L.orc_last_weights.restype = C.c_uint
    L.orc_ind_initialize.argtypes = [C.c_uint, C.c_uint, C.c_uint, C.c_void_p]
    L.orc_ind_initialize.restype = C.c_uint
    L.orc_search_window.argtypes = [C.c_int, C.c_uint, C.c_uint, C.POINTER(C.c_int),
                                    C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.orc_denoised_percent.argtypes = [_f32p, _u32p, C.c_uint, C.c_uint, C.c_uint, C.c_uint,
                                       C.c_uint, C.c_uint]
    L.orc_denoised_percent.restype = C.c_float
    L.orc_psnr.argtypes = [_f32p, _f32p, C.c_ulonglong, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.orc_ht_filter_slab.argtypes = [_f32p, C.c_uint, C.c_uint, C.c_uint, _f32p, C.c_float, _f32p, C.c_void_p, C.c_uint]
    L.orc_wiener_filter_slab.argtypes = [_f32p, _f32p, C.c_uint, C.c_uint, C.c_uint, _f32p, _f32p, C.c_void_p, C.c_uint]
    L.orc_set_threads.argtypes = [C.c_int]
    L.orc_set_time_limit.argtypes = [C.c_double]
    L.orc_set_tiles.argtypes = [C.c_int]
    L.orc_last_windows.argtypes = [C.c_void_p, C.c_uint]
    L.orc_last_windows.restype = C.c_int
    L.orc_get_threads.restype = C.c_int
    _lib = L
    return L


def ref_lib():
    """Compiled reference leaf routines (None when neither /root/reference nor a prebuilt copy exists)."""
    global _ref
    if _ref is not None:
        return _ref
    build()
    path = os.path.join(_HERE, "_ref", "libref_leaf.so")
    if not os.path.exists(path):
        return None
    R = C.CDLL(path)
    R.ref_haar_forward.argtypes = [_f32p, C.c_uint]
    R.ref_haar_inverse.argtypes = [_f32p, C.c_uint]
    R.ref_hadamard.argtypes = [_f32p, C.c_uint]
    R.ref_bior_forward.argtypes = [_f32p, C.c_uint, C.c_uint, _f32p, C.c_uint]
    R.ref_bior_inverse.argtypes = [_f32p, C.c_uint]
    R.ref_mt_seed.argtypes = [C.c_ulong]
    R.ref_mt_res53.restype = C.c_double
    _ref = R
    return R


def make_params(sigma, lam, N, nSim, nDisp, k, p, tau2, tau4, tau5, useSD=0, cs="opp"):
    t = lambda v: TAU[v] if isinstance(v, str) else v
    return Params(sigma, lam, N, nSim, nDisp, k, p, useSD, t(tau2), t(tau4), t(tau5),
                  CS[cs] if isinstance(cs, str) else cs)


def add_noise_lf(clean, sigma, seed=1):
    """clean: float32 [A][C*H*W]-shaped array (any shape, st-major).  One MT19937 stream seeded once,
    SAIs in st order, planar pixel order."""
    L = lib()
    clean = np.ascontiguousarray(clean, dtype=np.float32)
    out = np.empty_like(clean)
    L.orc_mt_seed(seed)
    L.orc_add_noise(clean.reshape(-1), out.reshape(-1), clean.size, sigma)
    return out


def psnr(a, b):
    L = lib()
    p, r = C.c_float(), C.c_float()
    a = np.ascontiguousarray(a, np.float32).reshape(-1)
    b = np.ascontiguousarray(b, np.float32).reshape(-1)
    L.orc_psnr(a, b, a.size, C.byref(p), C.byref(r))
    return p.value


def psnr_lf(a, b):
    """Mean over SAIs of the per-SAI PSNR (compute_psnr_LF, utilities_LF.cpp:639-692)."""
    return float(np.mean([psnr(a[i], b[i]) for i in range(a.shape[0])]))


def last_windows():
    """Processed SAI of every window the last run_step* call visited (the reference's data-driven choice)."""
    L = lib()
    n = L.orc_last_windows(None, 0)
    out = np.zeros(max(n, 1), np.uint32)
    L.orc_last_windows(out.ctypes.data, out.size)
    return out[:n].copy()


def run_step1(P, noisy, mask, ang_major, aw, ah, an, W, H, Cc, max_windows=0):
    L = lib()
    noisy = np.ascontiguousarray(noisy, np.float32)
    basic = np.zeros_like(noisy)
    st = Stats()
    rc = L.orc_run_step1(C.byref(P), noisy.reshape(-1), np.ascontiguousarray(mask, np.uint32),
                         basic.reshape(-1), ang_major, aw, ah, an, W, H, Cc, max_windows, C.byref(st))
    if rc:
        raise RuntimeError("orc_run_step1 failed")
    return noisy, basic, st


def run_step2(P, noisy, basic, mask, ang_major, aw, ah, an, W, H, Cc, max_windows=0):
    L = lib()
    noisy = np.ascontiguousarray(noisy, np.float32)
    basic = np.ascontiguousarray(basic, np.float32)
    den = np.zeros_like(noisy)
    st = Stats()
    rc = L.orc_run_step2(C.byref(P), noisy.reshape(-1), np.ascontiguousarray(mask, np.uint32),
                         basic.reshape(-1), den.reshape(-1), ang_major, aw, ah, an, W, H, Cc,
                         max_windows, C.byref(st))
    if rc:
        raise RuntimeError("orc_run_step2 failed")
    return noisy, basic, den, st


def bm3d_step(step, sigma, lam, noisy_sym, basic_sym, Wb, Hb, Cc, nHW, k, N, p, tau2, useSD=0, cs="opp"):
    """One BM3D step (bm3d.cpp:315-690) on a mirror-padded image [C][Hb][Wb]; returns numerator / denominator."""
    L = lib()
    noisy_sym = np.ascontiguousarray(noisy_sym, np.float32)
    out = np.zeros_like(noisy_sym)
    st = Stats()
    b = None
    if basic_sym is not None:
        basic_sym = np.ascontiguousarray(basic_sym, np.float32)
        b = basic_sym.ctypes.data_as(C.c_void_p)
    rc = L.orc_bm3d_step(step, sigma, lam, noisy_sym.reshape(-1), b, out.reshape(-1), Wb, Hb, Cc, nHW, k, N, p, useSD,
                         CS[cs] if isinstance(cs, str) else cs, TAU[tau2] if isinstance(tau2, str) else tau2, C.byref(st))
    if rc:
        raise RuntimeError("orc_bm3d_step failed")
    return out, st


def run_bm3d_lf(sigma, lam, noisy, mask, W, H, Cc, hard, wien, cs="opp"):
    """run_bm3d_LF (bm3d_LF.cpp:75-125).  hard / wien = (N, n, k, p, tau_2D, useSD).  Returns the mutated noisy LF,
    the basic and the denoised light fields and the stats."""
    L = lib()
    noisy = np.ascontiguousarray(noisy, np.float32).copy()
    basic = np.zeros_like(noisy)
    den = np.zeros_like(noisy)
    st = Stats()
    t = lambda v: TAU[v] if isinstance(v, str) else v
    rc = L.orc_run_bm3d_lf(sigma, noisy.reshape(-1), np.ascontiguousarray(mask, np.uint32), basic.reshape(-1), den.reshape(-1),
                           noisy.shape[0], W, H, Cc, hard[1], wien[1], hard[2], wien[2], hard[0], wien[0], hard[3], wien[3],
                           hard[5], wien[5], t(hard[4]), t(wien[4]), lam, CS[cs] if isinstance(cs, str) else cs, C.byref(st))
    if rc:
        raise RuntimeError("orc_run_bm3d_lf failed")
    return noisy, basic, den, st


def last_weights(n_groups, Cc):
    """Aggregation weights of the groups of the last orc_pass, [group][channel]."""
    out = np.zeros(n_groups * Cc, np.float32)
    n = lib().orc_last_weights(out, out.size)
    assert n >= out.size
    return out.reshape(n_groups, Cc)
