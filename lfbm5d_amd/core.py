"""ctypes binding of liblfbm5d_hip.so (include/lfbm5d.h) and the reference-named wrappers.

Device buffers are torch CUDA tensors (torch is plumbing for HBM allocations and, in bench.py,
torch.distributed for the rendezvous); the C-ABI itself only sees raw pointers.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

# enum ints of the reference (src/bm5d.cpp:36-48)
YUV, YCBCR, OPP, RGB, ID, DCT, SADCT, BIOR, HADAMARD, HAAR = range(10)
ROWMAJOR, COLMAJOR = 11, 12
TAU = {"id": ID, "dct": DCT, "sadct": SADCT, "bior": BIOR, "hw": HADAMARD, "haar": HAAR}
COLOR_SPACE = {"yuv": YUV, "ycbcr": YCBCR, "opp": OPP, "rgb": RGB}
UNIQUE_ID_BYTES = 128


class LfBm5dError(RuntimeError):
    pass


class Params(C.Structure):
    """lfbm5d_params: the parameter tail of run_bm5d_*_step (bm5d.h:11-62)."""
    _fields_ = [("sigma", C.c_float), ("lambda_", C.c_float), ("N", C.c_uint), ("nSim", C.c_uint),
                ("nDisp", C.c_uint), ("k", C.c_uint), ("p", C.c_uint), ("useSD", C.c_uint),
                ("tau_2D", C.c_uint), ("tau_4D", C.c_uint), ("tau_5D", C.c_uint),
                ("color_space", C.c_uint)]


class Bm3dParams(C.Structure):
    """lfbm5d_bm3d_params: one step's parameters of run_bm3d (src/bm3d.h:11-34)."""
    _fields_ = [("sigma", C.c_float), ("lambda3D", C.c_float), ("N", C.c_uint), ("nHW", C.c_uint), ("k", C.c_uint),
                ("p", C.c_uint), ("useSD", C.c_uint), ("tau_2D", C.c_uint), ("color_space", C.c_uint)]


class Stats(C.Structure):
    _fields_ = [("windows", C.c_ulonglong), ("passes", C.c_ulonglong), ("groups", C.c_ulonglong),
                ("stack_patches", C.c_ulonglong), ("sadct_groups", C.c_ulonglong),
                ("algorithmic_bytes", C.c_double), ("ms_bm", C.c_double), ("ms_group", C.c_double),
                ("ms_aggregate", C.c_double), ("ms_other", C.c_double), ("ms_comm", C.c_double),
                ("launches_group", C.c_ulonglong), ("launches_aggregate", C.c_ulonglong),
                ("lane_windows", C.c_ulonglong), ("messages", C.c_ulonglong)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


def library_path():
    return os.path.join(_HERE, "liblfbm5d_hip.so")


def build_library(force=False):
    """Compile the HIP extension in-tree (hipcc --offload-arch=gfx950; works without a GPU)."""
    src = os.path.join(_HERE, "csrc")
    args = ["make", "-C", src]
    if force:
        subprocess.check_call(args + ["clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(args, stdout=subprocess.DEVNULL)
    return library_path()


_lib = None


def lib():
    """Load the HIP extension.  Fails loudly if it has not been built: there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    # LFBM5D_HIP_LIB: an alternative build of the same library (kernel A/B experiments, tools/build_variant.sh)
    path = os.environ.get("LFBM5D_HIP_LIB") or library_path()
    if not os.path.exists(path):
        raise LfBm5dError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950).  lfbm5d_amd has no CPU fallback.")
    L = C.CDLL(path)
    vp, up, fp = C.c_void_p, C.POINTER(C.c_uint), C.c_void_p
    L.lfbm5d_create.argtypes = [C.POINTER(vp), C.c_int]
    L.lfbm5d_destroy.argtypes = [vp]
    L.lfbm5d_last_error.argtypes = [vp]
    L.lfbm5d_last_error.restype = C.c_char_p
    L.lfbm5d_reset_stats.argtypes = [vp]
    L.lfbm5d_get_stats.argtypes = [vp, C.POINTER(Stats)]
    L.lfbm5d_stream.argtypes = [vp]
    L.lfbm5d_stream.restype = vp
    if hasattr(L, "lfbm5d_set_option"):      # (absent from older builds loaded through LFBM5D_HIP_LIB for A/B runs: they read the environment themselves)
        L.lfbm5d_set_option.argtypes = [vp, C.c_char_p, C.c_char_p]
        L.lfbm5d_get_option.argtypes = [vp, C.c_char_p, C.c_char_p, C.c_ulonglong]
    L.lfbm5d_comm_unique_id.argtypes = [vp]
    L.lfbm5d_comm_init.argtypes = [vp, vp, C.c_int, C.c_int]
    if hasattr(L, "lfbm5d_comm_init_ipc"):   # (absent from older builds loaded through LFBM5D_HIP_LIB for A/B runs)
        L.lfbm5d_comm_init_ipc.argtypes = [vp, C.c_int, C.c_int, C.c_char_p, C.c_double]
    L.lfbm5d_set_shard.argtypes = [vp, C.c_int, C.c_int]
    L.lfbm5d_set_tiles.argtypes = [vp, C.c_int]
    L.lfbm5d_comm_ranks.argtypes = [vp]
    L.lfbm5d_comm_ranks.restype = C.c_int
    L.lfbm5d_comm_selftest.argtypes = [vp, C.c_uint]
    L.lfbm5d_auto_bands.argtypes = [C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.c_int]
    L.lfbm5d_auto_bands.restype = C.c_int
    L.lfbm5d_shard_rows.argtypes = [C.c_uint, C.c_int, C.c_int, up, up]
    L.lfbm5d_shard_rows.restype = None
    L.lfbm5d_plan_windows.argtypes = [C.c_uint, C.c_uint, C.c_uint, C.c_uint, up, up, C.c_uint]
    L.lfbm5d_last_windows.argtypes = [vp, up, C.c_uint]
    L.lfbm5d_plan_graph.argtypes = [C.c_uint, C.c_uint, C.c_uint, C.c_uint, up, C.c_int, C.c_int, up, up, up, C.c_uint]
    L.lfbm5d_plan_messages.argtypes = [C.c_uint, C.c_uint, C.c_uint, C.c_uint, up, C.c_int, up, C.c_uint]
    L.lfbm5d_plan_job.argtypes = [C.c_uint, C.c_uint, C.c_uint, up, C.c_int, up, up, C.c_int, C.c_int, up, C.c_uint, up, C.c_uint, up]
    L.lfbm5d_denoise_device.argtypes = [vp, C.POINTER(Params), C.POINTER(Params), fp, up, fp, fp] + [C.c_uint] * 8
    L.lfbm5d_denoise_host.argtypes = [vp, C.POINTER(Params), C.POINTER(Params), fp, up, fp, fp] + [C.c_uint] * 8
    tail = [C.c_uint] * 7
    L.lfbm5d_step1_device.argtypes = [vp, C.POINTER(Params), fp, up, fp] + tail
    L.lfbm5d_step2_device.argtypes = [vp, C.POINTER(Params), fp, up, fp, fp] + tail
    L.lfbm5d_step1_host.argtypes = [vp, C.POINTER(Params), fp, up, fp] + tail
    L.lfbm5d_step2_host.argtypes = [vp, C.POINTER(Params), fp, up, fp, fp] + tail
    # host seam with one pointer per SAI (what the reference's vector<vector<float>> is): arrays of float*
    if hasattr(L, "lfbm5d_denoise_host_sai"):
        L.lfbm5d_step1_host_sai.argtypes = [vp, C.POINTER(Params), fp, up, fp] + tail
        L.lfbm5d_step2_host_sai.argtypes = [vp, C.POINTER(Params), fp, up, fp, fp] + tail
        L.lfbm5d_denoise_host_sai.argtypes = [vp, C.POINTER(Params), C.POINTER(Params), fp, up, fp, fp] + [C.c_uint] * 8
    L.lfbm5d_pass_device.argtypes = [vp, C.c_int, C.POINTER(Params), C.c_uint, C.c_uint, C.c_uint,
                                     C.c_uint, C.c_uint, fp, fp, fp, fp, up, up, C.c_uint, C.c_uint]
    L.lfbm5d_last_bm.argtypes = [vp, up, vp, vp, vp, vp, vp]
    bp = C.POINTER(Bm3dParams)
    L.lfbm5d_bm3d_step_device.argtypes = [vp, C.c_int, bp, C.c_uint, C.c_uint, C.c_uint, fp, fp, fp]
    L.lfbm5d_bm3d_lf_device.argtypes = [vp, bp, bp, fp, up, fp, fp, C.c_uint, C.c_uint, C.c_uint, C.c_uint]
    L.lfbm5d_bm3d_lf_host.argtypes = [vp, bp, bp, fp, up, fp, fp, C.c_uint, C.c_uint, C.c_uint, C.c_uint]
    L.lfbm5d_last_tables.argtypes = [vp, vp, C.c_size_t]
    L.lfbm5d_last_tables.restype = C.c_size_t
    L.lfbm5d_last_weights.argtypes = [vp, vp, C.c_size_t]
    L.lfbm5d_last_weights.restype = C.c_size_t
    L.lfbm5d_last_scores.argtypes = [vp, vp, C.c_size_t]
    L.lfbm5d_last_scores.restype = C.c_size_t
    L.lfbm5d_last_scan_version.argtypes = [vp]
    L.lfbm5d_last_scan_version.restype = C.c_int
    L.lfbm5d_malloc.argtypes = [C.POINTER(vp), C.c_size_t]
    L.lfbm5d_free.argtypes = [vp]
    L.lfbm5d_memcpy_h2d.argtypes = [vp, vp, C.c_size_t]
    L.lfbm5d_memcpy_d2h.argtypes = [vp, vp, C.c_size_t]
    L.lfbm5d_device_count.restype = C.c_int
    _lib = L
    return L


def plan_windows(awidth, aheight, an=1, ang_major=None, mask=None):
    """Window schedule of a step (processed SAI of every window, in order); host only."""
    ang_major = ROWMAJOR if ang_major is None else ang_major
    m = np.ones(awidth * aheight, np.uint32) if mask is None else _u32(mask)
    out = np.zeros(awidth * aheight, np.uint32)
    n = lib().lfbm5d_plan_windows(awidth, aheight, an, ang_major, m.ctypes.data_as(C.POINTER(C.c_uint)),
                                  out.ctypes.data_as(C.POINTER(C.c_uint)), out.size)
    if n < 0:
        raise LfBm5dError("lfbm5d_plan_windows: bad arguments")
    return out[:n].copy()


def plan_graph(awidth, aheight, world, lanes=1, an=1, ang_major=None, mask=None):
    """Graph form of a step (host only): (rank, lane, start slot) of every planned window, as uint32 arrays."""
    ang_major = ROWMAJOR if ang_major is None else ang_major
    m = np.ones(awidth * aheight, np.uint32) if mask is None else _u32(mask)
    mp = m.ctypes.data_as(C.POINTER(C.c_uint))
    n = lib().lfbm5d_plan_graph(awidth, aheight, an, ang_major, mp, world, lanes, None, None, None, 0)
    if n < 0:
        raise LfBm5dError("lfbm5d_plan_graph: bad arguments")
    r, l, t = (np.zeros(max(n, 1), np.uint32) for _ in range(3))
    p = lambda a: a.ctypes.data_as(C.POINTER(C.c_uint))
    lib().lfbm5d_plan_graph(awidth, aheight, an, ang_major, mp, world, lanes, p(r), p(l), p(t), n)
    return r[:n], l[:n], t[:n]


def plan_messages(awidth, aheight, world, an=1, ang_major=None, mask=None):
    """Messages of the graph form in issue order: rows of (producer window, consumer window, SAI, channel)."""
    ang_major = ROWMAJOR if ang_major is None else ang_major
    m = np.ones(awidth * aheight, np.uint32) if mask is None else _u32(mask)
    mp = m.ctypes.data_as(C.POINTER(C.c_uint))
    n = lib().lfbm5d_plan_messages(awidth, aheight, an, ang_major, mp, world, None, 0)
    if n < 0:
        raise LfBm5dError("lfbm5d_plan_messages: bad arguments")
    out = np.zeros((max(n, 1), 4), np.uint32)
    lib().lfbm5d_plan_messages(awidth, aheight, an, ang_major, mp, world, out.ctypes.data_as(C.POINTER(C.c_uint)), n)
    return out[:n]


def plan_job(awidth, aheight, world, lanes=1, an=(1, 1), cost=None, ang_major=None, mask=None):
    """Graph of a job (len(an) = 1: one step; 2: both steps as lfbm5d_denoise_* runs them), host only.  Returns
    (nodes, msgs, info): nodes[i] = (step slot, window, processed SAI, graph rank, lane, start, issue position, chain) as a
    uint32 array [n, 8]; msgs[j] = (kind, producer node, consumer node | 0xffffffff, receiving rank, SAI, channel) [m, 6];
    info = dict(windows, messages, makespan, centre_ok)."""
    ang_major = ROWMAJOR if ang_major is None else ang_major
    m = np.ones(awidth * aheight, np.uint32) if mask is None else _u32(mask)
    p = lambda a: a.ctypes.data_as(C.POINTER(C.c_uint))
    an_a = _u32(list(an))
    cost_a = _u32(list(cost)) if cost is not None else None
    cnt = np.zeros(4, np.uint32)
    n = lib().lfbm5d_plan_job(awidth, aheight, ang_major, p(m), len(an_a), p(an_a), p(cost_a) if cost_a is not None else None,
                              world, lanes, None, 0, None, 0, p(cnt))
    if n < 0:
        raise LfBm5dError("lfbm5d_plan_job: bad arguments")
    nodes = np.zeros((max(int(cnt[0]), 1), 8), np.uint32)
    msgs = np.zeros((max(int(cnt[1]), 1), 6), np.uint32)
    lib().lfbm5d_plan_job(awidth, aheight, ang_major, p(m), len(an_a), p(an_a), p(cost_a) if cost_a is not None else None,
                          world, lanes, p(nodes), int(cnt[0]), p(msgs), int(cnt[1]), p(cnt))
    return nodes[:int(cnt[0])], msgs[:int(cnt[1])], {"windows": int(cnt[0]), "messages": int(cnt[1]), "makespan": int(cnt[2]),
                                                     "centre_ok": bool(cnt[3])}


def auto_bands(awidth, aheight, height, halo, world):
    """Spatial bands S for a two-step job on `world` ranks (lfbm5d_auto_bands, include/lfbm5d.h; tools/scale_model.py): the graph takes the
    ranks it can keep busy (a power of two within 0.8 ceil(a / 3)), bands take the rest, as long as a band stays twice as tall as its
    halo.  1 = the graph alone (bit-identical to one GPU)."""
    return int(lib().lfbm5d_auto_bands(int(awidth), int(aheight), int(height), int(halo), int(world)))


def shard_rows(n_rows, rank, world):
    b, e = C.c_uint(), C.c_uint()
    lib().lfbm5d_shard_rows(n_rows, rank, world, C.byref(b), C.byref(e))
    return b.value, e.value


def make_params(sigma, lam, N, nSim, nDisp, k, p, tau_2D, tau_4D, tau_5D, useSD=0, color_space=OPP):
    t = lambda v: TAU[v] if isinstance(v, str) else int(v)
    cs = COLOR_SPACE[color_space] if isinstance(color_space, str) else int(color_space)
    return Params(float(sigma), float(lam), int(N), int(nSim), int(nDisp), int(k), int(p),
                  int(bool(useSD)), t(tau_2D), t(tau_4D), t(tau_5D), cs)


def make_bm3d_params(sigma, lam, N, nHW, k, p, tau_2D, useSD=0, color_space=OPP):
    cs = COLOR_SPACE[color_space] if isinstance(color_space, str) else int(color_space)
    return Bm3dParams(float(sigma), float(lam), int(N), int(nHW), int(k), int(p), int(bool(useSD)),
                      TAU[tau_2D] if isinstance(tau_2D, str) else int(tau_2D), cs)


def _u32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.uint32))


def _dev_ptr(t):
    """Raw device pointer of a CUDA tensor.  The library launches on the context's own stream and knows nothing
    about torch's: work torch has queued on the tensor's device (a zero-fill, a copy) must have finished before the
    library reads the buffer, so the current torch stream is drained here (include/lfbm5d.h: "buffers are ready on
    entry, results complete on return")."""
    import torch
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise LfBm5dError("device entry points need contiguous float32 CUDA tensors")
    torch.cuda.current_stream(t.device).synchronize()
    return C.c_void_p(t.data_ptr())


def _sai_ptrs(arrays, mask):
    """float*[asize] over a list of per-SAI float32 arrays (entries of empty SAIs may be None)."""
    out = (C.c_void_p * len(arrays))()
    for i, a in enumerate(arrays):
        if a is None or not mask[i]:
            continue
        if not (isinstance(a, np.ndarray) and a.dtype == np.float32 and a.flags.c_contiguous):
            raise LfBm5dError("per-SAI host buffers must be contiguous float32 numpy arrays")
        out[i] = a.ctypes.data
    return out


# The library's run-time options (lfbm5d_amd/csrc/lfbm5d_options.h) and the environment variables they came from: the library
# reads the environment ONCE, at lfbm5d_create; afterwards options change through lfbm5d_set_option.  This Python mirror keeps
# the old convenience -- os.environ["LFBM5D_LANES"] = "3" between two calls on one Context takes effect -- by handing changed
# variables to lfbm5d_set_option before every library call (Context._h).
OPTION_ENV = ("LFBM5D_LANES", "LFBM5D_EMULATE_WORLD", "LFBM5D_MAX_WINDOWS", "LFBM5D_FUSED", "LFBM5D_STEP_SHARDING",
              "LFBM5D_DATA_DRIVEN_SCHEDULE", "LFBM5D_HOST_BLOCKING", "LFBM5D_BAND_MB", "LFBM5D_BM3D_LANES", "LFBM5D_SCAN_LDS_CAP",
              "LFBM5D_FORCE_REDO", "LFBM5D_SPATIAL_BANDS", "LFBM5D_BAND_HALO", "LFBM5D_SCAN_V1", "LFBM5D_SCAN_ANY", "LFBM5D_SCAN_FULL_TABLES", "LFBM5D_DCT8W_V2",
              "LFBM5D_GROUP_GENERIC", "LFBM5D_NO_SA_KERNELS", "LFBM5D_NO_SLAB_KERNEL", "LFBM5D_WIDE_NOSPLIT", "LFBM5D_AGG_64BIT",
              "LFBM5D_AGG_SCALAR_SCAN", "LFBM5D_SUBSET_LIST_HOST", "LFBM5D_SUBSET_SCAN_V1", "LFBM5D_FILT_GROUP_MAJOR")


class Context:
    """lfbm5d_ctx: one per process / GPU."""

    def __init__(self, device=0):
        self._L = lib()
        h = C.c_void_p()
        if self._L.lfbm5d_create(C.byref(h), int(device)) != 0:
            raise LfBm5dError(self._L.lfbm5d_last_error(None).decode())
        self._handle = h
        self._env_seen = {k: os.environ.get(k) for k in OPTION_ENV}   # what lfbm5d_create has just read
        self.device = int(device)

    @property
    def _h(self):
        """The context handle, with the options brought up to date with the environment (see OPTION_ENV)."""
        h = self._handle
        if h and hasattr(self._L, "lfbm5d_set_option"):
            for k in OPTION_ENV:
                v = os.environ.get(k)
                if v != self._env_seen[k]:
                    self._env_seen[k] = v
                    # a variable that is present but empty counted as "set" for the presence flags
                    self._L.lfbm5d_set_option(h, k.encode(), None if v is None else (v or "1").encode())
        return h

    def set_option(self, key, value):
        """lfbm5d_set_option: `key` as in lfbm5d_options.h ("lanes", "step_sharding", ...; the old variable names work too); None
        resets it to its default."""
        if self._L.lfbm5d_set_option(self._handle, key.encode(), None if value is None else str(value).encode()) != 0:
            raise LfBm5dError(self._L.lfbm5d_last_error(self._handle).decode())

    def get_option(self, key):
        buf = C.create_string_buffer(64)
        if self._L.lfbm5d_get_option(self._handle, key.encode(), buf, 64) != 0:
            raise LfBm5dError(self._L.lfbm5d_last_error(self._handle).decode())
        return buf.value.decode()

    def close(self):
        if getattr(self, "_handle", None):
            self._L.lfbm5d_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != 0:
            raise LfBm5dError(self._L.lfbm5d_last_error(self._h).decode())

    # ---- multi-GPU ----
    @staticmethod
    def unique_id():
        buf = (C.c_ubyte * UNIQUE_ID_BYTES)()
        if lib().lfbm5d_comm_unique_id(buf) != 0:
            raise LfBm5dError("ncclGetUniqueId failed")
        return bytes(buf)

    def comm_init(self, unique_id, rank, world):
        buf = (C.c_ubyte * UNIQUE_ID_BYTES).from_buffer_copy(unique_id)
        self._ck(self._L.lfbm5d_comm_init(self._h, buf, rank, world))

    def comm_init_ipc(self, rank, world, rendezvous_dir, timeout_s=30.0):
        """Tests only: this process is rank `rank` of `world` processes sharing ONE GPU; the window graph's messages travel through
        IPC-mapped device buffers instead of RCCL (include/lfbm5d.h)."""
        self._ck(self._L.lfbm5d_comm_init_ipc(self._h, rank, world, os.fsencode(rendezvous_dir), float(timeout_s)))

    def comm_selftest(self, n=1 << 20):
        """All-reduce n floats through RCCL on the context's stream and check the sums."""
        self._ck(self._L.lfbm5d_comm_selftest(self._h, n))

    def comm_ranks(self):
        """Ranks of the RCCL communicator as RCCL counts them (0: none)."""
        return int(self._L.lfbm5d_comm_ranks(self._h))

    def set_shard(self, rank, world):
        self._ck(self._L.lfbm5d_set_shard(self._h, rank, world))

    def set_tiles(self, nb_tiles):
        """The reference's tile mode (run_bm5d_* with nb_threads > 1) for whole steps; 0 / 1 switches it off."""
        self._ck(self._L.lfbm5d_set_tiles(self._h, int(nb_tiles)))

    # ---- stats ----
    def reset_stats(self):
        self._L.lfbm5d_reset_stats(self._h)

    def stats(self):
        s = Stats()
        self._L.lfbm5d_get_stats(self._h, C.byref(s))
        return s

    def stream(self):
        return self._L.lfbm5d_stream(self._h)

    # ---- outer seam ----
    def step1(self, P, noisy, mask, basic, ang_major, awidth, aheight, an, W, H, Cc):
        """run_bm5d_1st_step on device tensors (torch CUDA) or host arrays (numpy float32)."""
        m = _u32(mask)
        mp = m.ctypes.data_as(C.POINTER(C.c_uint))
        tail = (ang_major, awidth, aheight, an, W, H, Cc)
        if isinstance(noisy, (list, tuple)):   # one float32 array per SAI, like the reference's vector<vector<float>>
            self._ck(self._L.lfbm5d_step1_host_sai(self._h, C.byref(P), _sai_ptrs(noisy, m), mp, _sai_ptrs(basic, m), *tail))
        elif isinstance(noisy, np.ndarray):
            self._ck(self._L.lfbm5d_step1_host(self._h, C.byref(P), noisy.ctypes.data_as(C.c_void_p), mp,
                                               basic.ctypes.data_as(C.c_void_p), *tail))
        else:
            self._ck(self._L.lfbm5d_step1_device(self._h, C.byref(P), _dev_ptr(noisy), mp, _dev_ptr(basic), *tail))

    def step2(self, P, noisy, mask, basic, denoised, ang_major, awidth, aheight, an, W, H, Cc):
        m = _u32(mask)
        mp = m.ctypes.data_as(C.POINTER(C.c_uint))
        tail = (ang_major, awidth, aheight, an, W, H, Cc)
        if isinstance(noisy, (list, tuple)):
            self._ck(self._L.lfbm5d_step2_host_sai(self._h, C.byref(P), _sai_ptrs(noisy, m), mp, _sai_ptrs(basic, m), _sai_ptrs(denoised, m), *tail))
        elif isinstance(noisy, np.ndarray):
            self._ck(self._L.lfbm5d_step2_host(self._h, C.byref(P), noisy.ctypes.data_as(C.c_void_p), mp,
                                               basic.ctypes.data_as(C.c_void_p),
                                               denoised.ctypes.data_as(C.c_void_p), *tail))
        else:
            self._ck(self._L.lfbm5d_step2_device(self._h, C.byref(P), _dev_ptr(noisy), mp, _dev_ptr(basic),
                                                 _dev_ptr(denoised), *tail))

    def denoise(self, P1, P2, noisy, mask, basic, denoised, ang_major, awidth, aheight, an1, an2, W, H, Cc):
        """run_bm5d_1st_step + run_bm5d_2nd_step as one job (lfbm5d_denoise_*): bit-identical to step1() followed by step2()."""
        m = _u32(mask)
        mp = m.ctypes.data_as(C.POINTER(C.c_uint))
        tail = (ang_major, awidth, aheight, an1, an2, W, H, Cc)
        if isinstance(noisy, (list, tuple)):
            self._ck(self._L.lfbm5d_denoise_host_sai(self._h, C.byref(P1), C.byref(P2), _sai_ptrs(noisy, m), mp, _sai_ptrs(basic, m),
                                                     _sai_ptrs(denoised, m), *tail))
        elif isinstance(noisy, np.ndarray):
            self._ck(self._L.lfbm5d_denoise_host(self._h, C.byref(P1), C.byref(P2), noisy.ctypes.data_as(C.c_void_p), mp,
                                                 basic.ctypes.data_as(C.c_void_p), denoised.ctypes.data_as(C.c_void_p), *tail))
        else:
            self._ck(self._L.lfbm5d_denoise_device(self._h, C.byref(P1), C.byref(P2), _dev_ptr(noisy), mp, _dev_ptr(basic),
                                                   _dev_ptr(denoised), *tail))

    # ---- inner seam ----
    def core_pass(self, step, P, aw, ah, Wb, Hb, Cc, noisy, basic, num, den, mask, procSAI, cst, pst):
        """bm5d_1st_step / bm5d_2nd_step on a padded window held in CUDA tensors."""
        m, pr = _u32(mask), _u32(procSAI)
        self._ck(self._L.lfbm5d_pass_device(
            self._h, step, C.byref(P), aw, ah, Wb, Hb, Cc, _dev_ptr(noisy),
            _dev_ptr(basic) if basic is not None else None, _dev_ptr(num), _dev_ptr(den),
            m.ctypes.data_as(C.POINTER(C.c_uint)), pr.ctypes.data_as(C.POINTER(C.c_uint)), cst, pst))

    # ---- per-SAI BM3D (LFBM3Ddenoising) ----
    def bm3d_step(self, step, P, Wb, Hb, Cc, noisy, basic, out):
        """bm3d_1st_step / bm3d_2nd_step on a mirror-padded image held in CUDA tensors; out = num / den."""
        self._ck(self._L.lfbm5d_bm3d_step_device(self._h, step, C.byref(P), Wb, Hb, Cc, _dev_ptr(noisy),
                                                 _dev_ptr(basic) if basic is not None else None, _dev_ptr(out)))

    def bm3d_lf(self, hard, wien, noisy, mask, basic, denoised, W, H, Cc):
        """run_bm3d_LF on device tensors (torch CUDA) or host arrays (numpy float32), [asize][C*H*W]."""
        m = _u32(mask)
        mp = m.ctypes.data_as(C.POINTER(C.c_uint))
        if isinstance(noisy, np.ndarray):
            self._ck(self._L.lfbm5d_bm3d_lf_host(self._h, C.byref(hard), C.byref(wien), noisy.ctypes.data_as(C.c_void_p), mp,
                                                 basic.ctypes.data_as(C.c_void_p), denoised.ctypes.data_as(C.c_void_p),
                                                 m.size, W, H, Cc))
        else:
            self._ck(self._L.lfbm5d_bm3d_lf_device(self._h, C.byref(hard), C.byref(wien), _dev_ptr(noisy), mp, _dev_ptr(basic),
                                                   _dev_ptr(denoised), m.size, W, H, Cc))

    def last_windows(self):
        """Processed SAI of every window the last step call ran, in order."""
        n = self._L.lfbm5d_last_windows(self._h, None, 0)
        out = np.zeros(max(n, 1), np.uint32)
        self._L.lfbm5d_last_windows(self._h, out.ctypes.data_as(C.POINTER(C.c_uint)), out.size)
        return out[:max(n, 0)].copy()

    def last_bm(self, N, A, plane):
        """Block-matching tables of the last pass, as numpy arrays."""
        n = C.c_uint()
        self._ck(self._L.lfbm5d_last_bm(self._h, C.byref(n), None, None, None, None, None))
        R = n.value
        refs = np.zeros(R, np.uint32)
        idx = np.zeros((R, max(N, 1)), np.uint32)
        cnt = np.zeros(R, np.uint32)
        best = np.zeros((A, plane), np.uint32)
        shape = np.zeros((A, plane), np.uint8)
        self._ck(self._L.lfbm5d_last_bm(self._h, C.byref(n), refs.ctypes.data, idx.ctypes.data, cnt.ctypes.data,
                                        best.ctypes.data, shape.ctypes.data))
        return refs, idx, cnt, best, shape


    def last_tables(self, n_floats=None):
        """Raw disparity distance tables of the last pass (flat float32 array)."""
        have = self._L.lfbm5d_last_tables(self._h, None, 0)
        n = have if n_floats is None else min(have, int(n_floats))
        out = np.zeros(n, np.float32)
        got = self._L.lfbm5d_last_tables(self._h, out.ctypes.data, n)
        return out[:got]

    def last_scores(self, n_floats=None):
        """Candidate scores of the self-similarity search of the last pass (flat float32 array)."""
        have = self._L.lfbm5d_last_scores(self._h, None, 0)
        n = have if n_floats is None else min(have, int(n_floats))
        out = np.zeros(n, np.float32)
        got = self._L.lfbm5d_last_scores(self._h, out.ctypes.data, n)
        return out[:got]

    def last_weights(self, n_groups, C_):
        """Aggregation weights of the groups of the last pass, [group][channel]."""
        out = np.zeros(n_groups * C_, np.float32)
        got = self._L.lfbm5d_last_weights(self._h, out.ctypes.data, out.size)
        assert got == out.size
        return out.reshape(n_groups, C_)

    def last_scan_version(self):
        return int(self._L.lfbm5d_last_scan_version(self._h))


_default_ctx = None


def _ctx():
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(0)
    return _default_ctx


def run_bm5d_1st_step(sigma, lambdaHard5D, LF_noisy, LF_SAI_mask, LF_basic, ang_major, awidth, aheight,
                      anHard, width, height, chnls, NHard, nSim, nDisp, kHard, pHard, useSD, tau_2D, tau_4D,
                      tau_5D, color_space, nb_threads=1, ctx=None):
    """Same argument list as the reference's run_bm5d_1st_step (src/bm5d.h:11-35).  LF_noisy is
    mutated in place and LF_basic filled, like the reference; nb_threads is accepted and ignored
    (the GPU path always has the untiled, nb_threads == 1 semantics).  Returns 0 (EXIT_SUCCESS)
    or raises LfBm5dError with the library's message."""
    P = make_params(sigma, lambdaHard5D, NHard, nSim, nDisp, kHard, pHard, tau_2D, tau_4D, tau_5D, useSD, color_space)
    (ctx or _ctx()).step1(P, LF_noisy, LF_SAI_mask, LF_basic, ang_major, awidth, aheight, anHard, width, height, chnls)
    return 0


def run_bm5d_2nd_step(sigma, LF_noisy, LF_SAI_mask, LF_basic, LF_denoised, ang_major, awidth, aheight,
                      anWien, width, height, chnls, NWien, nSim, nDisp, kWien, pWien, useSD, tau_2D, tau_4D,
                      tau_5D, color_space, nb_threads=1, ctx=None):
    """Same argument list as the reference's run_bm5d_2nd_step (src/bm5d.h:38-62)."""
    P = make_params(sigma, 0.0, NWien, nSim, nDisp, kWien, pWien, tau_2D, tau_4D, tau_5D, useSD, color_space)
    (ctx or _ctx()).step2(P, LF_noisy, LF_SAI_mask, LF_basic, LF_denoised, ang_major, awidth, aheight, anWien,
                          width, height, chnls)
    return 0


def run_bm3d_LF(sigma, LF_noisy, LF_SAI_mask, LF_basic, LF_denoised, width, height, chnls, nHard, nWien, kHard, kWien,
                NHard, NWien, pHard, pWien, useSD_h, useSD_w, tau_2D_hard, tau_2D_wien, lambdaHard3D, color_space,
                nb_threads=1, sub_img_name="SAI", ctx=None):
    """Same argument list as the reference's run_bm3d_LF (src/bm3d_LF.h:10-35): BM3D on every SAI of the mask.
    nb_threads is accepted and ignored (untiled, nb_threads == 1 semantics)."""
    hard = make_bm3d_params(sigma, lambdaHard3D, NHard, nHard, kHard, pHard, tau_2D_hard, useSD_h, color_space)
    wien = make_bm3d_params(sigma, lambdaHard3D, NWien, nWien, kWien, pWien, tau_2D_wien, useSD_w, color_space)
    (ctx or _ctx()).bm3d_lf(hard, wien, LF_noisy, LF_SAI_mask, LF_basic, LF_denoised, width, height, chnls)
    return 0


_TAU = {"id": 4, "dct": 5, "sadct": 6, "bior": 7, "hw": 8, "hadamard": 8, "haar": 9}
_CS = {"yuv": 0, "ycbcr": 1, "opp": 2, "rgb": 3}


def dropin_probe(noisy, mask, awidth, aheight, width, height, chnls, sigma, lambda_, hard, wien, color_space="opp", ang_major=ROWMAJOR,
                 an=(1, 1), one_job=False, reps=1):
    """Time the reference's own interval through the C++ drop-in (liblfbm5d_dropin.so: run_bm5d_1st_step + run_bm5d_2nd_step on
    vector<vector<float>> light fields, main.cpp:189-201 + :241-247; one_job: run_bm5d).  hard / wien = (N, nSim, nDisp, k, p,
    tau_2D, tau_4D, tau_5D[, useSD]).  Returns (ms [reps][2], noisy, basic, denoised) -- the light fields as the calls leave them."""
    path = os.path.join(os.path.dirname(library_path()), "liblfbm5d_dropin.so")
    if not os.path.exists(path):
        raise LfBm5dError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
    D = C.CDLL(path)
    D.lfbm5d_dropin_probe.argtypes = ([C.c_int] + [C.c_void_p] * 5 + [C.c_uint] * 8 + [C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_uint,
                                      C.c_int, C.c_void_p])

    def pk(t):
        t = tuple(t)
        v = [t[0], t[1], t[2], t[3], t[4], int(t[8]) if len(t) > 8 else 0, _TAU[t[5]], _TAU[t[6]], _TAU[t[7]]]
        return np.array(v, np.uint32)
    noisy = np.ascontiguousarray(noisy, np.float32)
    m = _u32(mask)
    n_out, b_out, d_out = np.zeros_like(noisy), np.zeros_like(noisy), np.zeros_like(noisy)
    ms = np.zeros((reps, 2), np.float64)
    h, w = pk(hard), pk(wien)
    rc = D.lfbm5d_dropin_probe(1 if one_job else 0, noisy.ctypes.data, m.ctypes.data, n_out.ctypes.data, b_out.ctypes.data, d_out.ctypes.data,
                               ang_major, awidth, aheight, an[0], an[1], width, height, chnls, sigma, lambda_, h.ctypes.data, w.ctypes.data,
                               _CS[color_space] if isinstance(color_space, str) else int(color_space), reps, ms.ctypes.data)
    if rc != 0:
        raise LfBm5dError("drop-in probe failed (message on stdout)")
    return ms, n_out, b_out, d_out
