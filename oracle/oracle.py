"""ctypes front-end of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (hypotremormcmc_amd) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

PARAM_FIELDS = [
    ("n_procs", C.c_int), ("n_chains", C.c_int), ("n_cool", C.c_int),
    ("n_iter", C.c_int), ("n_burn", C.c_int), ("n_interval", C.c_int),
    ("temp_high", C.c_double),
    ("prior_z", C.c_double), ("prior_width_z", C.c_double), ("prior_width_xy", C.c_double),
    ("prior_vs", C.c_double), ("prior_width_vs", C.c_double),
    ("prior_qs", C.c_double), ("prior_width_qs", C.c_double),
    ("prior_t_corr", C.c_double), ("prior_width_t_corr", C.c_double),
    ("prior_a_corr", C.c_double), ("prior_width_a_corr", C.c_double),
    ("step_size_z", C.c_double), ("step_size_xy", C.c_double), ("step_size_vs", C.c_double),
    ("step_size_qs", C.c_double), ("step_size_t_corr", C.c_double), ("step_size_a_corr", C.c_double),
    ("solve_vs", C.c_int), ("solve_t_corr", C.c_int), ("solve_qs", C.c_int), ("solve_a_corr", C.c_int),
    ("use_time", C.c_int), ("use_amp", C.c_int),
]


class OrcParams(C.Structure):
    _fields_ = PARAM_FIELDS


class OrcRng(C.Structure):
    _fields_ = [("x", C.c_uint32), ("y", C.c_uint32), ("z", C.c_uint32), ("w", C.c_uint32)]


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "htm_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    L = C.CDLL(build())
    dp = C.POINTER(C.c_double)
    ip = C.POINTER(C.c_int32)
    vp = C.c_void_p
    L.orc_rng_seed.argtypes = [C.POINTER(OrcRng)] + [C.c_int32] * 5
    for n in ("orc_rand_u", "orc_rand_u2", "orc_rand_g", "orc_rand_r"):
        getattr(L, n).argtypes = [C.POINTER(OrcRng)]
        getattr(L, n).restype = C.c_double
    L.orc_forward_create.restype = vp
    L.orc_forward_create.argtypes = [C.c_int, C.c_int] + [dp] * 7 + [C.c_int, C.c_int]
    L.orc_forward_destroy.argtypes = [vp]
    L.orc_forward_travel_time.argtypes = [vp, dp, dp, C.c_double, dp]
    L.orc_forward_amp.argtypes = [vp, dp, dp, C.c_double, C.c_double, dp]
    L.orc_forward_travel_time_single.argtypes = [vp, C.c_int, dp, dp, C.c_double, dp]
    L.orc_forward_amp_single.argtypes = [vp, C.c_int, dp, dp, C.c_double, C.c_double, dp]
    L.orc_forward_loglik_full.restype = C.c_double
    L.orc_forward_loglik_full.argtypes = [vp, dp, dp, C.c_double, dp, C.c_double]
    L.orc_forward_loglik_partial.restype = C.c_double
    L.orc_forward_loglik_partial.argtypes = [vp, C.c_int, dp, C.c_double, dp, dp, C.c_double, dp, C.c_double]
    L.orc_select_regress.argtypes = [C.c_int, C.c_int] + [dp] * 3 + [C.c_double] + [dp] * 5
    L.orc_job_create.restype = vp
    L.orc_job_create.argtypes = [C.POINTER(OrcParams), C.c_int, C.c_int] + [dp] * 7
    L.orc_job_destroy.argtypes = [vp]
    L.orc_job_run.argtypes = [vp, C.c_int]
    L.orc_job_record_words.argtypes = [vp]
    L.orc_job_rank_begin.argtypes = [vp, C.c_int, dp]
    L.orc_job_rank_end.argtypes = [vp, C.c_int, dp]
    L.orc_job_n_lik.argtypes = [vp, C.c_int]
    L.orc_job_get_lik.argtypes = [vp, C.c_int, ip, dp]
    L.orc_job_n_samples.argtypes = [vp, C.c_int]
    L.orc_job_get_sample.argtypes = [vp, C.c_int, C.c_int, ip, dp, dp, dp, dp, dp]
    L.orc_job_get_counts.argtypes = [vp, ip, ip]
    L.orc_job_get_chain.argtypes = [vp, C.c_int, C.c_int, dp, dp, dp, dp, dp, dp, dp, ip, ip]
    L.orc_job_get_priors.argtypes = [vp, C.c_int, C.c_int, dp, dp, dp, ip]
    L.orc_job_get_rng.argtypes = [vp, C.c_int, C.POINTER(C.c_uint32)]
    L.orc_job_enable_steplog.argtypes = [vp, C.c_int]
    L.orc_job_steplog_n.argtypes = [vp]
    L.orc_job_get_steplog.argtypes = [vp, ip, dp]
    _LIB = L
    return L


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(C.POINTER(C.c_double))


def make_params(p: dict) -> OrcParams:
    """p: dict with the reference's parameter-file keys (values as python numbers / 'T' / 'F')."""
    def tf(v):
        if isinstance(v, str):
            return 1 if v.strip().upper().lstrip(".").startswith("T") else 0
        return int(bool(v))

    def num(v):
        if isinstance(v, str):
            return float(v.lower().replace("d", "e"))
        return float(v)

    op = OrcParams()
    for name, ctype in PARAM_FIELDS:
        if name in ("prior_t_corr", "prior_a_corr") and name not in p:
            setattr(op, name, 0.0)  # defaults, src/cls_param.f90:89,:91
            continue
        v = p[name]
        if name.startswith("solve_") or name.startswith("use_"):
            setattr(op, name, tf(v))
        elif ctype is C.c_int:
            setattr(op, name, int(v))
        else:
            setattr(op, name, num(v))
    return op


class Rng:
    def __init__(self, rank: int = 0, seeds=(5551111, 453222, 4444431, 6765)):
        self.s = OrcRng()
        lib().orc_rng_seed(C.byref(self.s), *seeds, rank)

    @property
    def state(self):
        return (self.s.x, self.s.y, self.s.z, self.s.w)

    def rand_u(self):
        return lib().orc_rand_u(C.byref(self.s))

    def rand_u2(self):
        return lib().orc_rand_u2(C.byref(self.s))

    def rand_g(self):
        return lib().orc_rand_g(C.byref(self.s))

    def rand_r(self):
        return lib().orc_rand_r(C.byref(self.s))


class Forward:
    """Mirror of reference `type forward` (src/cls_forward.f90:6-41). Arrays (n_events, n_sta) C-order
    == Fortran (n_sta, n_events)."""

    def __init__(self, sta_x, sta_y, sta_z, t_obs, t_stdv, a_obs, a_stdv, use_time=True, use_amp=True):
        t_obs = np.ascontiguousarray(t_obs, dtype=np.float64)
        self.n_events, self.n_sta = t_obs.shape
        keep = [_d(a) for a in (sta_x, sta_y, sta_z, t_obs, t_stdv, a_obs, a_stdv)]
        self.h = lib().orc_forward_create(self.n_sta, self.n_events, *[k[1] for k in keep],
                                          int(use_time), int(use_amp))

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_forward_destroy(self.h)
            self.h = None

    def calc_log_likelihood(self, hypo, t_corr, vs, a_corr, qs):
        h, hp = _d(hypo); t, tp = _d(t_corr); a, ap = _d(a_corr)
        return lib().orc_forward_loglik_full(self.h, hp, tp, float(vs), ap, float(qs))

    def partially_update_log_likelihood(self, evt_id, hypo_old, loglik_old, hypo, t_corr, vs, a_corr, qs):
        ho, hop = _d(hypo_old); h, hp = _d(hypo); t, tp = _d(t_corr); a, ap = _d(a_corr)
        return lib().orc_forward_loglik_partial(self.h, int(evt_id), hop, float(loglik_old), hp, tp,
                                                float(vs), ap, float(qs))

    def calc_travel_time(self, hypo, t_corr, vs):
        h, hp = _d(hypo); t, tp = _d(t_corr)
        out = np.empty((self.n_events, self.n_sta))
        lib().orc_forward_travel_time(self.h, hp, tp, float(vs), out.ctypes.data_as(C.POINTER(C.c_double)))
        return out

    def calc_amp(self, hypo, a_corr, qs, vs):
        h, hp = _d(hypo); a, ap = _d(a_corr)
        out = np.empty((self.n_events, self.n_sta))
        lib().orc_forward_amp(self.h, hp, ap, float(qs), float(vs), out.ctypes.data_as(C.POINTER(C.c_double)))
        return out

    def calc_travel_time_single(self, evt_id, hypo, t_corr, vs):
        h, hp = _d(hypo); t, tp = _d(t_corr)
        out = np.empty(self.n_sta)
        lib().orc_forward_travel_time_single(self.h, int(evt_id), hp, tp, float(vs),
                                             out.ctypes.data_as(C.POINTER(C.c_double)))
        return out

    def calc_amp_single(self, evt_id, hypo, a_corr, qs, vs):
        h, hp = _d(hypo); a, ap = _d(a_corr)
        out = np.empty(self.n_sta)
        lib().orc_forward_amp_single(self.h, int(evt_id), hp, ap, float(qs), float(vs),
                                     out.ctypes.data_as(C.POINTER(C.c_double)))
        return out


class Job:
    """Whole step-5 job on the CPU (all ranks simulated in lock step)."""

    def __init__(self, params: dict, data):
        self.p = make_params(params)
        self.n_sta, self.n_events = data.n_sta, data.n_events
        keep = [_d(a) for a in (data.sta_x, data.sta_y, data.sta_z, data.t_obs, data.t_stdv,
                                data.a_obs, data.a_stdv)]
        self.h = lib().orc_job_create(C.byref(self.p), self.n_sta, self.n_events, *[k[1] for k in keep])

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_job_destroy(self.h)
            self.h = None

    def run(self, n: int):
        lib().orc_job_run(self.h, int(n))

    # per-rank lock-step mode (the multi-GPU protocol on the CPU)
    def record_words(self) -> int:
        return lib().orc_job_record_words(self.h)

    def rank_begin(self, rank: int, record: np.ndarray):
        assert record.dtype == np.float64 and record.flags.c_contiguous
        lib().orc_job_rank_begin(self.h, int(rank), record.ctypes.data_as(C.POINTER(C.c_double)))

    def rank_end(self, rank: int, gathered: np.ndarray) -> int:
        assert gathered.dtype == np.float64 and gathered.flags.c_contiguous
        return lib().orc_job_rank_end(self.h, int(rank), gathered.ctypes.data_as(C.POINTER(C.c_double)))

    def likelihood_trace(self, rank: int):
        n = lib().orc_job_n_lik(self.h, rank)
        it = np.empty(n, dtype=np.int32)
        lk = np.empty(n, dtype=np.float64)
        if n:
            lib().orc_job_get_lik(self.h, rank, it.ctypes.data_as(C.POINTER(C.c_int32)),
                                  lk.ctypes.data_as(C.POINTER(C.c_double)))
        return it, lk

    def samples(self, rank: int):
        n = lib().orc_job_n_samples(self.h, rank)
        E, S = self.n_events, self.n_sta
        out = dict(iter=np.empty(n, np.int32), vs=np.empty(n), qs=np.empty(n), hypo=np.empty((n, 3 * E)),
                   t_corr=np.empty((n, S)), a_corr=np.empty((n, S)))
        dp = C.POINTER(C.c_double)
        for k in range(n):
            it = C.c_int32(); vs = C.c_double(); qs = C.c_double()
            lib().orc_job_get_sample(self.h, rank, k, C.byref(it), C.byref(vs), C.byref(qs),
                                     out["hypo"][k].ctypes.data_as(dp), out["t_corr"][k].ctypes.data_as(dp),
                                     out["a_corr"][k].ctypes.data_as(dp))
            out["iter"][k] = it.value; out["vs"][k] = vs.value; out["qs"][k] = qs.value
        return out

    def counts(self):
        npr = np.zeros(7, np.int32); nac = np.zeros(7, np.int32)
        ip = C.POINTER(C.c_int32)
        lib().orc_job_get_counts(self.h, npr.ctypes.data_as(ip), nac.ctypes.data_as(ip))
        return npr, nac

    def chain(self, rank: int, chain: int):
        E, S = self.n_events, self.n_sta
        dp = C.POINTER(C.c_double); ip = C.POINTER(C.c_int32)
        hypo = np.empty(3 * E); tc = np.empty(S); ac = np.empty(S)
        vs = C.c_double(); qs = C.c_double(); temp = C.c_double(); ll = C.c_double()
        npr = np.zeros(7, np.int32); nac = np.zeros(7, np.int32)
        lib().orc_job_get_chain(self.h, rank, chain, hypo.ctypes.data_as(dp), tc.ctypes.data_as(dp),
                                C.cast(C.byref(vs), dp), ac.ctypes.data_as(dp), C.cast(C.byref(qs), dp),
                                C.cast(C.byref(temp), dp), C.cast(C.byref(ll), dp),
                                npr.ctypes.data_as(ip), nac.ctypes.data_as(ip))
        return dict(hypo=hypo, t_corr=tc, a_corr=ac, vs=vs.value, qs=qs.value, temp=temp.value,
                    loglik=ll.value, n_propose=npr, n_accept=nac)

    def hypo_priors(self, rank: int, chain: int):
        n = 3 * self.n_events
        dp = C.POINTER(C.c_double)
        mu = np.empty(n); sg = np.empty(n); st = np.empty(n); pt = np.empty(n, np.int32)
        lib().orc_job_get_priors(self.h, rank, chain, mu.ctypes.data_as(dp), sg.ctypes.data_as(dp),
                                 st.ctypes.data_as(dp), pt.ctypes.data_as(C.POINTER(C.c_int32)))
        return mu, sg, st, pt

    def rng_state(self, rank: int):
        st = (C.c_uint32 * 4)()
        lib().orc_job_get_rng(self.h, rank, st)
        return tuple(int(v) for v in st)

    def enable_steplog(self, cap: int):
        lib().orc_job_enable_steplog(self.h, int(cap))

    def steplog(self):
        n = lib().orc_job_steplog_n(self.h)
        ir = np.empty((n, 8), np.int32); dr = np.empty((n, 4), np.float64)
        if n:
            lib().orc_job_get_steplog(self.h, ir.ctypes.data_as(C.POINTER(C.c_int32)),
                                      dr.ctypes.data_as(C.POINTER(C.c_double)))
        return ir, dr


def select_regress(sta_x, sta_y, sta_z, z_guess, t, t_err, a, a_err):
    """Step 4 of the reference per window: rows {vs, b, t0, a0, cc_t, cc_a} (src/cls_selector.f90:75-132).
    t, t_err, a, a_err: shape (n_win, n_sta)."""
    t = np.ascontiguousarray(t, dtype=np.float64)
    n_win, n_sta = t.shape
    out = np.empty((n_win, 6))
    arrs = [np.ascontiguousarray(v, dtype=np.float64) for v in (sta_x, sta_y, sta_z)]
    obs = [np.ascontiguousarray(v, dtype=np.float64) for v in (t, t_err, a, a_err)]
    lib().orc_select_regress(n_sta, n_win, *[_d(v)[1] for v in arrs], float(z_guess), *[_d(v)[1] for v in obs], _d(out)[1])
    return out
