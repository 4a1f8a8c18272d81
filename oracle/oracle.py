"""ctypes front-end of the CPU oracle (oracle/cals_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product (cp-cals_amd/) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
OR_MAX_MODES = 8
MTTKRP, TWOSTEP0, TWOSTEP1, AUTO = 0, 1, 2, 3
LS_NO_ERROR_CHECKING, LS_ERROR_CHECKING_SERIAL = 0, 1
UNCONSTRAINED, NNLS = 0, 1


class OrParams(C.Structure):
    _fields_ = [
        ("max_iterations", C.c_int64), ("tol", C.c_double), ("buffer_size", C.c_int64),
        ("mttkrp_method", C.c_int), ("line_search", C.c_int), ("line_search_interval", C.c_int),
        ("line_search_step", C.c_double), ("line_search_method", C.c_int),
        ("force_max_iter", C.c_int), ("always_evict_first", C.c_int), ("threads", C.c_int),
        ("update_method", C.c_int),
    ]


class OrReport(C.Structure):
    _fields_ = [
        ("iter", C.c_int64), ("n_ktensors", C.c_int64), ("ktensor_comp_sum", C.c_int64),
        ("ls_performed", C.c_int64), ("ls_failed", C.c_int64), ("X_norm", C.c_double),
        ("total_time", C.c_double), ("loop_time", C.c_double), ("mttkrp_time", C.c_double),
        ("nnls_status", C.c_int),
    ]


class OrModel(C.Structure):
    _fields_ = [
        ("rank", C.c_int64), ("factors", C.POINTER(C.c_double) * OR_MAX_MODES),
        ("lambda_", C.POINTER(C.c_double)), ("jk_enabled", C.c_int), ("jk_mode", C.c_int),
        ("jk_fiber", C.c_int64), ("iters", C.c_int64), ("fit", C.c_double),
        ("old_fit", C.c_double), ("approx_error", C.c_double), ("ls_margin", C.c_double),
    ]


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "cals_oracle.c")
    hdr = os.path.join(_HERE, "cals_oracle.h")
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src),
                                                                      os.path.getmtime(hdr)):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"],
                              stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        # CALS_ORACLE_LIB: another build of the same source (tests load the ASan/UBSan one)
        _LIB = C.CDLL(os.environ.get("CALS_ORACLE_LIB") or build())
        _LIB.or_norm.restype = C.c_double
        _LIB.or_fast_error.restype = C.c_double
        # parity runs are single-threaded and deterministic; the cpu_baseline leg raises this.
        # (OpenMP regions on a many-core host cost more than the tiny per-model loops they wrap.)
        _LIB.or_set_threads(1)
    return _LIB


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _modes(modes):
    return (C.c_int64 * len(modes))(*[int(m) for m in modes])


def fcol(a):
    """Column-major float64 copy (what the reference's Matrix holds)."""
    return np.asfortranarray(np.array(a, dtype=np.float64, order="F"))


def default_params(**kw):
    p = OrParams()
    lib().or_default_params(C.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def set_threads(n):
    lib().or_set_threads(int(n))


_MKL = None


def use_mkl(threads=None):
    """Route the big MTTKRP GEMMs through the image's MKL runtime (cblas_dgemm).  Returns True
    when MKL was found.  Must be called before MKL is first used with MKL_THREADING_LAYER=GNU."""
    global _MKL
    os.environ.setdefault("MKL_THREADING_LAYER", "GNU")
    if _MKL is None:
        for cand in ("/opt/conda/lib/libmkl_rt.so", "/opt/conda/lib/libmkl_rt.so.1", "libmkl_rt.so"):
            try:
                _MKL = C.CDLL(cand, mode=C.RTLD_GLOBAL)
                break
            except OSError:
                continue
    if _MKL is None:
        return False
    if threads is not None:
        _MKL.MKL_Set_Num_Threads(int(threads))
    fn = C.cast(_MKL.cblas_dgemm, C.c_void_p)
    lib().or_set_dgemm(fn)
    return True


def use_own_gemm():
    lib().or_set_dgemm(C.c_void_p(0))


# ---------------------------------------------------------------------------------------------
# building blocks
# ---------------------------------------------------------------------------------------------
def mttkrp(X, modes, factors, mode, method=MTTKRP):
    """factors: list of (I_n x R) arrays -> G (I_mode x R), column-major."""
    fs = [fcol(f) for f in factors]
    R = fs[0].shape[1]
    G = np.zeros((modes[mode], R), order="F")
    ptrs = (C.POINTER(C.c_double) * len(modes))(*[_dp(f) for f in fs])
    Xf = np.ascontiguousarray(np.asarray(X, dtype=np.float64).ravel())
    lib().or_mttkrp(_dp(Xf), len(modes), _modes(modes), ptrs, C.c_int64(R), int(mode), int(method),
                    _dp(G))
    return G


def khatri_rao(A, B):
    A, B = fcol(A), fcol(B)
    K = np.zeros((A.shape[0] * B.shape[0], A.shape[1]), order="F")
    lib().or_khatri_rao(_dp(A), C.c_int64(A.shape[0]), _dp(B), C.c_int64(B.shape[0]),
                        C.c_int64(A.shape[1]), _dp(K))
    return K


def gramian(panel):
    P = fcol(panel)
    r = P.shape[1]
    G = np.zeros((r, r), order="F")
    lib().or_update_gramian(_dp(P), C.c_int64(P.shape[0]), C.c_int64(r), C.c_int64(P.shape[0]), _dp(G))
    return G


def update_step(G_panel, gramians, mode, iteration, jk_fiber=None):
    """One per-model update (cals.cpp:239-256) on an MTTKRP result panel.
    Returns (new panel, lambda, new gramian of `mode`, potrf info)."""
    P = fcol(G_panel)
    r = P.shape[1]
    gs = [fcol(g) for g in gramians]
    ptrs = (C.POINTER(C.c_double) * len(gs))(*[_dp(g) for g in gs])
    lib().or_hadamard_but_one(ptrs, len(gs), C.c_int64(r), int(mode))
    info = lib().or_update_factor_unconstrained(_dp(P), C.c_int64(P.shape[0]), C.c_int64(r),
                                                C.c_int64(P.shape[0]), _dp(gs[mode]))
    if jk_fiber is not None:
        P[jk_fiber, :] *= 0.0
    lam = np.zeros(r)
    lib().or_normalize_mode(_dp(P), C.c_int64(P.shape[0]), C.c_int64(r), C.c_int64(P.shape[0]),
                            _dp(lam), C.c_int64(iteration))
    lib().or_update_gramian(_dp(P), C.c_int64(P.shape[0]), C.c_int64(r), C.c_int64(P.shape[0]),
                            _dp(gs[mode]))
    return P, lam, gs[mode], info


def update_factor_nnls(G_panel, H, active=None):
    """update::update_factor_non_negative_constrained (src/utils/update.cpp:61-176) on one panel.
    G_panel: rows x r MTTKRP result, H: r x r Hadamard of the other Gramians, active: rows x r uint8
    (1 = constraint active; None = a fresh Ktensor's all-active sets).
    Returns (solution rows x r, active rows x r, status)."""
    P = fcol(G_panel)
    Hc = fcol(H)
    rows, r = P.shape
    act = np.ones((rows, r), dtype=np.uint8) if active is None else np.ascontiguousarray(active, dtype=np.uint8).copy()
    lib().or_update_factor_nnls.restype = C.c_int
    st = lib().or_update_factor_nnls(_dp(P), C.c_int64(rows), C.c_int64(r), C.c_int64(rows), _dp(Hc),
                                     act.ctypes.data_as(C.POINTER(C.c_uint8)))
    return P, act, int(st)


def nnls_set_termination(bound=0, cycle_rule=True):
    """Test knobs of or_update_factor_nnls: bound = 0 -> the default max(64, 16 r); cycle_rule off + a lifted
    bound = the reference's unbounded loops (src/utils/update.cpp:95-165) wherever those end."""
    lib().or_nnls_set_termination(C.c_int64(int(bound)), C.c_int(1 if cycle_rule else 0))


def nnls_last_max_passes():
    lib().or_nnls_last_max_passes.restype = C.c_int64
    return int(lib().or_nnls_last_max_passes())


def fast_error(X_norm, lam, last_factor, last_G, gram_had):
    F, G, H = fcol(last_factor), fcol(last_G), fcol(gram_had)
    lam = np.ascontiguousarray(lam, dtype=np.float64)
    return lib().or_fast_error(C.c_double(X_norm), _dp(lam), _dp(F), C.c_int64(F.shape[0]),
                               C.c_int64(F.shape[1]), C.c_int64(F.shape[0]), _dp(G),
                               C.c_int64(G.shape[0]), _dp(H))


def jk_norms(X, modes):
    Xf = np.ascontiguousarray(np.asarray(X, dtype=np.float64).ravel())
    out = np.zeros(modes[0])
    lib().or_jk_norms(_dp(Xf), len(modes), _modes(modes), _dp(out))
    return out


def to_tensor(factors, lam, modes):
    """Ktensor::to_tensor -> flat array, mode 0 fastest."""
    fs = [fcol(f) for f in factors]
    r = fs[0].shape[1]
    lam = np.ascontiguousarray(lam, dtype=np.float64)
    out = np.zeros(int(np.prod(modes)))
    ptrs = (C.POINTER(C.c_double) * len(modes))(*[_dp(f) for f in fs])
    lib().or_to_tensor(ptrs, _dp(lam), len(modes), _modes(modes), C.c_int64(r), _dp(out))
    return out


def normalize_all(factors, modes):
    """Ktensor::normalize(): in-place column 2-norm scaling of col-major factors; returns lambda."""
    r = factors[0].shape[1]
    lam = np.zeros(r)
    ptrs = (C.POINTER(C.c_double) * len(modes))(*[_dp(f) for f in factors])
    lib().or_normalize_all(ptrs, len(modes), _modes(modes), C.c_int64(r), _dp(lam))
    return lam


class Model:
    """Host-side Ktensor: col-major factors + lambda (+ jk), results after a run."""

    def __init__(self, factors, lam=None, jk=None):
        self.factors = [fcol(f) for f in factors]
        self.rank = self.factors[0].shape[1]
        self.lam = np.ones(self.rank) if lam is None else np.array(lam, dtype=np.float64)
        self.jk = jk  # (mode, fiber) or None
        self.iters = 0
        self.fit = self.old_fit = self.error = 0.0
        self.ls_margin = 1e300

    def copy(self):
        m = Model([f.copy(order="F") for f in self.factors], self.lam.copy(), self.jk)
        return m

    def _fill(self, om):
        om.rank = self.rank
        for n, f in enumerate(self.factors):
            om.factors[n] = _dp(f)
        om.lambda_ = _dp(self.lam)
        om.jk_enabled = 0 if self.jk is None else 1
        om.jk_mode = 0 if self.jk is None else int(self.jk[0])
        om.jk_fiber = 0 if self.jk is None else int(self.jk[1])

    def _read(self, om):
        self.iters, self.fit, self.old_fit, self.error = om.iters, om.fit, om.old_fit, om.approx_error
        self.ls_margin = om.ls_margin


def cp_als(X, modes, model, params):
    Xf = np.ascontiguousarray(np.asarray(X, dtype=np.float64).ravel())
    om = OrModel()
    model._fill(om)
    rep = OrReport()
    rc = lib().or_cp_als(_dp(Xf), len(modes), _modes(modes), C.byref(om), C.byref(params), C.byref(rep))
    if rc != 0:
        raise RuntimeError("or_cp_als rc=%d" % rc)
    model._read(om)
    return rep


def cp_cals(X, modes, models, params):
    Xf = np.ascontiguousarray(np.asarray(X, dtype=np.float64).ravel())
    arr = (OrModel * len(models))()
    for m, om in zip(models, arr):
        m._fill(om)
    rep = OrReport()
    rc = lib().or_cp_cals(_dp(Xf), len(modes), _modes(modes), arr, C.c_int64(len(models)),
                          C.byref(params), C.byref(rep))
    if rc != 0:
        raise RuntimeError("or_cp_cals rc=%d" % rc)
    for m, om in zip(models, arr):
        m._read(om)
    return rep
