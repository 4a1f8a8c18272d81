"""cp-cals_amd: MI355X-native Concurrent-ALS (CALS) hot path.

Python is only the test/bench harness here: this module is a thin ctypes binding of the C ABI in
include/cals_hip.h (libcals_hip.so, hand-written HIP for gfx950).  There is no CPU fallback: if the
shared library is missing or no gfx950 device is visible, construction fails loudly.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcals_hip.so")
MAX_MODES = 8
MAX_RANK = 256

OK, ERR_ARG, ERR_HIP, ERR_STATE, ERR_FULL, ERR_NO_DEVICE = 0, 1, 2, 3, 4, 5


class CalsHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("cals_hip error %d: %s" % (code, msg))
        self.code = code


class Params(C.Structure):
    """CalsParams fields that steer the loop (reference include/cals.h:138-159)."""
    _fields_ = [
        ("max_iterations", C.c_int64), ("tol", C.c_double), ("line_search", C.c_int),
        ("line_search_interval", C.c_int), ("line_search_step", C.c_double),
        ("line_search_method", C.c_int), ("force_max_iter", C.c_int),
        ("always_evict_first", C.c_int), ("update_method", C.c_int),
    ]


class Report(C.Structure):
    _fields_ = [
        ("iter", C.c_int64), ("n_ktensors", C.c_int64), ("ktensor_comp_sum", C.c_int64),
        ("ls_performed", C.c_int64), ("ls_failed", C.c_int64), ("X_norm", C.c_double),
        ("total_ms", C.c_double), ("loop_ms", C.c_double), ("nnls_status", C.c_int),
    ]


class ModelStatus(C.Structure):
    _fields_ = [("iters", C.c_int64), ("fit", C.c_double), ("old_fit", C.c_double),
                ("approx_error", C.c_double), ("evicted", C.c_int)]


class SweepRecord(C.Structure):
    """cals_hip_sweep_record: one outer sweep of run()/step() while the sweep log is on."""
    _fields_ = [
        ("cols", C.c_int64), ("models", C.c_int64), ("flops", C.c_double), ("iteration_ms", C.c_double),
        ("admit_ms", C.c_double), ("defrag_ms", C.c_double), ("ls_ms", C.c_double),
        ("mttkrp_ms", C.c_double * MAX_MODES), ("update_ms", C.c_double * MAX_MODES),
        ("fused_ms", C.c_double * MAX_MODES), ("ttm_ms", C.c_double * MAX_MODES),
        ("contract_ms", C.c_double * MAX_MODES), ("krp_ms", C.c_double * MAX_MODES),
    ]


class KernelStats(C.Structure):
    _fields_ = [
        ("mttkrp_launches", C.c_int64), ("mttkrp_ms", C.c_double), ("mttkrp_flops", C.c_double),
        ("update_launches", C.c_int64), ("update_ms", C.c_double),
        ("other_launches", C.c_int64), ("other_ms", C.c_double),
        ("ttm_launches", C.c_int64), ("ttm_ms", C.c_double), ("ttm_flops", C.c_double),
        ("contract_launches", C.c_int64), ("contract_ms", C.c_double), ("contract_bytes", C.c_double),
    ]


DTYPES = {"f64": 0, "f32": 1}  # CALS_HIP_F64 / CALS_HIP_F32

EXPORTS = [
    "cals_hip_default_params", "cals_hip_create", "cals_hip_destroy", "cals_hip_last_error",
    "cals_hip_set_tensor", "cals_hip_set_params", "cals_hip_enqueue", "cals_hip_run",
    "cals_hip_model_result", "cals_hip_admit", "cals_hip_sweep", "cals_hip_evict",
    "cals_hip_step", "cals_hip_get_report",
    "cals_hip_active_cols", "cals_hip_models_in_flight", "cals_hip_queue_size",
    "cals_hip_synchronize", "cals_hip_mttkrp", "cals_hip_debug_mttkrp", "cals_hip_debug_mttkrp_path", "cals_hip_debug_get_factor",
    "cals_hip_debug_get_lambda", "cals_hip_debug_get_gramian", "cals_hip_debug_model_status",
    "cals_hip_debug_get_norms", "cals_hip_set_profiling", "cals_hip_get_kernel_stats",
    "cals_hip_reset_kernel_stats", "cals_hip_stream", "cals_hip_device_count",
    "cals_hip_create_ex", "cals_hip_dtype", "cals_hip_tree", "cals_hip_set_tensor_f32",
    "cals_hip_rebind", "cals_hip_capacity", "cals_hip_set_sweep_log", "cals_hip_get_sweep_log",
    "cals_hip_debug_clock", "cals_hip_debug_ttm_trace", "cals_hip_debug_ls_margin", "cals_hip_debug_install_crash_trace", "cals_hip_host_first_fit", "cals_hip_host_compress_plan", "cals_hip_host_active_cols",
]

_LIB = None


def load_library():
    """dlopen libcals_hip.so (no GPU needed for this step); raises if it is missing."""
    global _LIB
    if _LIB is not None:
        return _LIB
    # CALS_HIP_LIB=<path>: an experiment build of the same library (tools/ab_libs.sh, tools/ttm_strip.sh) for THIS
    # process only -- the shipped file is never overwritten
    path = os.environ.get("CALS_HIP_LIB") or LIB_PATH
    if not os.path.exists(path):
        raise CalsHipError(ERR_NO_DEVICE, "%s not built (run __graft_entry__.build()); the engine "
                           "has no CPU fallback" % path)
    lib = C.CDLL(path)
    vp, i64, dp = C.c_void_p, C.c_int64, C.POINTER(C.c_double)
    lib.cals_hip_create.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(i64), i64, C.c_int]
    lib.cals_hip_create_ex.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(i64), i64, C.c_int, C.c_int]
    lib.cals_hip_dtype.argtypes = [vp]
    lib.cals_hip_debug_ttm_trace.argtypes = [vp, C.POINTER(C.c_uint64), C.c_int]
    lib.cals_hip_tree.argtypes = [vp]
    lib.cals_hip_rebind.argtypes = [vp, i64]
    lib.cals_hip_capacity.argtypes = [vp]
    lib.cals_hip_capacity.restype = i64
    lib.cals_hip_set_sweep_log.argtypes = [vp, C.c_int]
    lib.cals_hip_get_sweep_log.argtypes = [vp, C.POINTER(SweepRecord), i64]
    lib.cals_hip_get_sweep_log.restype = i64
    lib.cals_hip_set_tensor_f32.argtypes = [vp, C.POINTER(C.c_float)]
    lib.cals_hip_destroy.argtypes = [vp]
    lib.cals_hip_last_error.argtypes = [vp]
    lib.cals_hip_last_error.restype = C.c_char_p
    lib.cals_hip_set_tensor.argtypes = [vp, dp]
    lib.cals_hip_mttkrp.argtypes = [vp, i64, C.POINTER(dp), C.c_int, dp, C.POINTER(C.c_double)]
    lib.cals_hip_set_params.argtypes = [vp, C.POINTER(Params)]
    lib.cals_hip_enqueue.argtypes = [vp, i64, C.POINTER(dp), dp, C.c_int, i64, C.POINTER(i64)]
    lib.cals_hip_run.argtypes = [vp, C.POINTER(Report)]
    lib.cals_hip_model_result.argtypes = [vp, i64, C.POINTER(ModelStatus)]
    lib.cals_hip_admit.argtypes = [vp, C.POINTER(i64)]
    lib.cals_hip_sweep.argtypes = [vp, i64]
    lib.cals_hip_evict.argtypes = [vp, C.POINTER(i64)]
    lib.cals_hip_step.argtypes = [vp, C.POINTER(i64), C.POINTER(i64)]
    lib.cals_hip_get_report.argtypes = [vp, C.POINTER(Report)]
    for f in ("cals_hip_active_cols", "cals_hip_models_in_flight", "cals_hip_queue_size"):
        getattr(lib, f).argtypes = [vp]
        getattr(lib, f).restype = i64
    lib.cals_hip_synchronize.argtypes = [vp]
    lib.cals_hip_debug_mttkrp.argtypes = [vp, C.c_int, dp]
    lib.cals_hip_debug_mttkrp_path.argtypes = [vp, C.c_int, C.c_int, dp]
    lib.cals_hip_debug_get_factor.argtypes = [vp, C.c_int, dp]
    lib.cals_hip_debug_get_lambda.argtypes = [vp, dp]
    lib.cals_hip_debug_get_gramian.argtypes = [vp, C.c_int, dp]
    lib.cals_hip_debug_model_status.argtypes = [vp, i64, C.POINTER(ModelStatus), C.POINTER(i64)]
    lib.cals_hip_debug_get_norms.argtypes = [vp, dp, dp]
    lib.cals_hip_debug_ls_margin.argtypes = [vp, i64, dp]
    lib.cals_hip_set_profiling.argtypes = [vp, C.c_int]
    lib.cals_hip_get_kernel_stats.argtypes = [vp, C.POINTER(KernelStats)]
    lib.cals_hip_reset_kernel_stats.argtypes = [vp]
    lib.cals_hip_stream.argtypes = [vp]
    lib.cals_hip_stream.restype = vp
    lib.cals_hip_device_count.restype = C.c_int
    lib.cals_hip_debug_clock.argtypes = [vp, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    pi64 = C.POINTER(i64)
    lib.cals_hip_host_first_fit.argtypes = [pi64, i64, i64]
    lib.cals_hip_host_first_fit.restype = i64
    lib.cals_hip_host_compress_plan.argtypes = [pi64, i64, pi64, pi64, i64]
    lib.cals_hip_host_compress_plan.restype = i64
    lib.cals_hip_host_active_cols.argtypes = [pi64, i64]
    lib.cals_hip_host_active_cols.restype = i64
    _LIB = lib
    return lib


def host_first_fit(occupancy, rank):
    occ = np.ascontiguousarray(occupancy, dtype=np.int64)
    return load_library().cals_hip_host_first_fit(occ.ctypes.data_as(C.POINTER(C.c_int64)), occ.size, int(rank))


def host_compress_plan(occupancy):
    occ = np.ascontiguousarray(occupancy, dtype=np.int64)
    ids = np.zeros(occ.size, dtype=np.int64)
    offs = np.zeros(occ.size, dtype=np.int64)
    p = C.POINTER(C.c_int64)
    n = load_library().cals_hip_host_compress_plan(occ.ctypes.data_as(p), occ.size, ids.ctypes.data_as(p),
                                                   offs.ctypes.data_as(p), occ.size)
    return list(zip(ids[:n].tolist(), offs[:n].tolist()))


def host_active_cols(occupancy):
    occ = np.ascontiguousarray(occupancy, dtype=np.int64)
    return load_library().cals_hip_host_active_cols(occ.ctypes.data_as(C.POINTER(C.c_int64)), occ.size)


def default_params(**kw):
    p = Params()
    load_library().cals_hip_default_params(C.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class Model:
    """Host-side Ktensor handed to the engine: col-major factors + lambda (+ jk).  The arrays are
    overwritten with the fitted model when the engine evicts it."""

    def __init__(self, factors, lam, jk=None):
        self.factors = [np.asfortranarray(np.array(f, dtype=np.float64, order="F")) for f in factors]
        self.rank = int(self.factors[0].shape[1])
        self.lam = np.array(lam, dtype=np.float64)
        self.jk = jk
        self.ticket = -1
        self.iters = 0
        self.fit = self.old_fit = self.error = 0.0


class Engine:
    """One CALS engine on one GPU (cals_hip_engine)."""

    def __init__(self, modes, buffer_size, device=0, dtype="f64"):
        """dtype: "f64" (reference arithmetic) or "f32" (fp32 storage + fp32 MFMA MTTKRP,
        BASELINE config 4); factors cross the boundary as float64 either way."""
        self.lib = load_library()
        self.modes = [int(m) for m in modes]
        self.h = C.c_void_p()
        if dtype not in DTYPES:
            raise ValueError("dtype must be one of %s" % sorted(DTYPES))
        self.dtype = dtype
        arr = (C.c_int64 * len(self.modes))(*self.modes)
        rc = self.lib.cals_hip_create_ex(C.byref(self.h), len(self.modes), arr, int(buffer_size),
                                         int(device), DTYPES[dtype])
        if rc != OK:
            msg = self.lib.cals_hip_last_error(self.h).decode() if self.h else "create failed"
            if self.h:
                self.lib.cals_hip_destroy(self.h)
                self.h = C.c_void_p()
            raise CalsHipError(rc, msg)
        self._models = []

    def _chk(self, rc):
        if rc != OK:
            raise CalsHipError(rc, self.lib.cals_hip_last_error(self.h).decode())

    def close(self):
        if self.h:
            self.lib.cals_hip_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def rebind(self, buffer_size):
        """Re-target the idle engine at another buffer_size <= capacity (same X copies, fresh packing)."""
        self._chk(self.lib.cals_hip_rebind(self.h, int(buffer_size)))
        self._models = []

    @property
    def capacity(self):
        return int(self.lib.cals_hip_capacity(self.h))

    def set_sweep_log(self, on):
        self._chk(self.lib.cals_hip_set_sweep_log(self.h, 1 if on else 0))

    def sweep_log(self):
        n = self.lib.cals_hip_get_sweep_log(self.h, None, 0)
        arr = (SweepRecord * max(int(n), 1))()
        self.lib.cals_hip_get_sweep_log(self.h, arr, n)
        return list(arr)[:n]

    def set_tensor(self, X):
        X = np.asarray(X)
        assert X.size == int(np.prod(self.modes))
        if X.dtype == np.float32:
            Xf = np.ascontiguousarray(X.ravel())
            self._chk(self.lib.cals_hip_set_tensor_f32(self.h, Xf.ctypes.data_as(C.POINTER(C.c_float))))
            return
        Xf = np.ascontiguousarray(np.asarray(X, dtype=np.float64).ravel())
        self._chk(self.lib.cals_hip_set_tensor(self.h, _dp(Xf)))

    def set_params(self, params):
        self._chk(self.lib.cals_hip_set_params(self.h, C.byref(params)))

    def enqueue(self, model):
        ptrs = (C.POINTER(C.c_double) * len(self.modes))(*[_dp(f) for f in model.factors])
        t = C.c_int64(-1)
        jm = -1 if model.jk is None else int(model.jk[0])
        jf = 0 if model.jk is None else int(model.jk[1])
        self._chk(self.lib.cals_hip_enqueue(self.h, model.rank, ptrs, _dp(model.lam), jm, jf, C.byref(t)))
        model.ticket = t.value
        self._models.append(model)  # keep the arrays alive
        return t.value

    def run(self):
        rep = Report()
        self._chk(self.lib.cals_hip_run(self.h, C.byref(rep)))
        for m in self._models:
            self.result(m)
        return rep

    def result(self, model):
        st = ModelStatus()
        self._chk(self.lib.cals_hip_model_result(self.h, model.ticket, C.byref(st)))
        model.iters, model.fit, model.old_fit, model.error = st.iters, st.fit, st.old_fit, st.approx_error
        return st

    def admit(self):
        n = C.c_int64(0)
        self._chk(self.lib.cals_hip_admit(self.h, C.byref(n)))
        return n.value

    def sweep(self, n=1):
        self._chk(self.lib.cals_hip_sweep(self.h, int(n)))

    def evict(self):
        n = C.c_int64(0)
        self._chk(self.lib.cals_hip_evict(self.h, C.byref(n)))
        return n.value

    def step(self):
        """One iteration of run()'s loop; returns (admitted, evicted)."""
        na, ne = C.c_int64(0), C.c_int64(0)
        self._chk(self.lib.cals_hip_step(self.h, C.byref(na), C.byref(ne)))
        return na.value, ne.value

    def report(self):
        rep = Report()
        self._chk(self.lib.cals_hip_get_report(self.h, C.byref(rep)))
        return rep

    def synchronize(self):
        self._chk(self.lib.cals_hip_synchronize(self.h))

    @property
    def active_cols(self):
        return self.lib.cals_hip_active_cols(self.h)

    @property
    def models_in_flight(self):
        return self.lib.cals_hip_models_in_flight(self.h)

    @property
    def tree(self):
        """0 = one fused MTTKRP per mode and sweep, 1 = dimension tree A, 2 = tree B, 3 = multi-sweep tree M (3-way);
        4 = two-group dimension tree (N > 3 modes)  (cals_hip_tree)."""
        return int(self.lib.cals_hip_tree(self.h))

    @property
    def queue_size(self):
        return int(self.lib.cals_hip_queue_size(self.h))

    def mttkrp(self, factors, mode):
        """cals_hip_mttkrp: the MTTKRP of `mode` for ONE Ktensor given by host factors (mttkrp::mttkrp,
        src/utils/mttkrp.cpp:562-614) on an idle engine.  Returns (G [I_mode x rank], device milliseconds)."""
        fs = [np.asfortranarray(f, dtype=np.float64) for f in factors]
        rank = fs[0].shape[1]
        G = np.zeros((self.modes[mode], rank), order="F")
        arr = (C.POINTER(C.c_double) * len(fs))(*[_dp(f) for f in fs])
        ms = C.c_double(0.0)
        self._chk(self.lib.cals_hip_mttkrp(self.h, C.c_int64(rank), arr, int(mode), _dp(G), C.byref(ms)))
        return G, ms.value

    def debug_mttkrp(self, mode, path=None):
        """path: None = the path a sweep takes under the engine's plan; "plain" | "first" | "second"
        = cals_hip_debug_mttkrp_path (raises CalsHipError if the plan has no such path)."""
        R = self.active_cols
        G = np.zeros((self.modes[mode], R), order="F")
        if path is None:
            self._chk(self.lib.cals_hip_debug_mttkrp(self.h, int(mode), _dp(G)))
        else:
            code = {"plain": 0, "first": 1, "second": 2}[path]
            self._chk(self.lib.cals_hip_debug_mttkrp_path(self.h, int(mode), code, _dp(G)))
        return G

    def debug_factor(self, mode):
        F = np.zeros((self.modes[mode], self.active_cols), order="F")
        self._chk(self.lib.cals_hip_debug_get_factor(self.h, int(mode), _dp(F)))
        return F

    def debug_lambda(self):
        lam = np.zeros(self.active_cols)
        self._chk(self.lib.cals_hip_debug_get_lambda(self.h, _dp(lam)))
        return lam

    def debug_gramian(self, mode):
        G = np.zeros((MAX_RANK, self.active_cols), order="F")
        self._chk(self.lib.cals_hip_debug_get_gramian(self.h, int(mode), _dp(G)))
        return G

    def debug_status(self, model):
        st = ModelStatus()
        col = C.c_int64(-1)
        self._chk(self.lib.cals_hip_debug_model_status(self.h, model.ticket, C.byref(st), C.byref(col)))
        return st, col.value

    def ls_margin(self, model):
        """cals_hip_debug_ls_margin: the smallest relative distance between the two errors of any accept / revert test
        the model went through (1e300: none).  Rounding-level = the decision was a tie."""
        m = C.c_double(0)
        self._chk(self.lib.cals_hip_debug_ls_margin(self.h, model.ticket, C.byref(m)))
        return m.value

    def debug_norms(self):
        xn = C.c_double(0)
        jk = np.zeros(self.modes[0])
        self._chk(self.lib.cals_hip_debug_get_norms(self.h, C.byref(xn), _dp(jk)))
        return xn.value, jk

    def set_profiling(self, level):
        """0/False off, 1/True every launch, 2 MFMA kernels + contraction only (cheaper, see cals_hip.h), 3 MFMA kernels only."""
        self._chk(self.lib.cals_hip_set_profiling(self.h, int(level)))

    def kernel_stats(self):
        ks = KernelStats()
        self._chk(self.lib.cals_hip_get_kernel_stats(self.h, C.byref(ks)))
        return ks

    def reset_kernel_stats(self):
        self._chk(self.lib.cals_hip_reset_kernel_stats(self.h))

    def debug_clock(self, n_workgroups=252):
        cyc, ghz = C.c_double(0), C.c_double(0)
        self._chk(self.lib.cals_hip_debug_clock(self.h, int(n_workgroups), C.byref(cyc), C.byref(ghz)))
        return cyc.value, ghz.value

    @property
    def stream(self):
        return self.lib.cals_hip_stream(self.h)
