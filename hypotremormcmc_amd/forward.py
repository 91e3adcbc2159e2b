"""Host mirror of `type forward` (reference src/cls_forward.f90:6-41) over the HIP C ABI.

Same constructor keywords and bound-procedure names / argument order as the reference so that callers
(and the parity tests) read like reference code.  Arrays follow the reference layout: observation arrays
are (n_sta, n_events) column-major, i.e. NumPy shape (n_events, n_sta) C-order.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, dp


def _arr(a, n=None):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(-1))
    if n is not None and a.size != n:
        raise ValueError(f"expected {n} values, got {a.size}")
    return a


def _p(a):
    return a.ctypes.data_as(dp)


class ObsArrays:
    """What `obs%get_t_obs() .. get_a_stdv()` return (src/cls_forward.f90:71-74)."""

    def __init__(self, t_obs, t_stdv, a_obs, a_stdv):
        self.t_obs, self.t_stdv, self.a_obs, self.a_stdv = t_obs, t_stdv, a_obs, a_stdv

    def get_t_obs(self): return self.t_obs
    def get_t_stdv(self): return self.t_stdv
    def get_a_obs(self): return self.a_obs
    def get_a_stdv(self): return self.a_stdv


class Forward:
    def __init__(self, n_sta, n_events, sta_x, sta_y, sta_z, obs, use_amp=True, use_time=True, device=0,
                 forward_precision="fp64"):
        self._lib = _lib.load()
        self.n_sta, self.n_events = int(n_sta), int(n_events)
        n = self.n_sta * self.n_events
        arrs = [_arr(sta_x, n_sta), _arr(sta_y, n_sta), _arr(sta_z, n_sta), _arr(obs.get_t_obs(), n),
                _arr(obs.get_t_stdv(), n), _arr(obs.get_a_obs(), n), _arr(obs.get_a_stdv(), n)]
        h = C.c_void_p()
        check(self._lib.htm_forward_create(self.n_sta, self.n_events, *[_p(a) for a in arrs],
                                           int(bool(use_time)), int(bool(use_amp)), int(device), C.byref(h)))
        self.handle = h
        self.device = device
        self.forward_precision = "fp64"
        if str(forward_precision).lower() in ("fp32", "f32", "single"):
            self.set_precision("fp32")

    def close(self):
        if getattr(self, "handle", None):
            self._lib.htm_forward_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_precision(self, precision: str):
        """"fp32": single-precision forward model with fp64 sums and accept (BASELINE configs[4]; statistical
        tolerance, no reference counterpart); "fp64": the reference's arithmetic.  Before creating chains."""
        fp32 = str(precision).lower() in ("fp32", "f32", "single")
        check(self._lib.htm_forward_set_precision(self.handle, int(fp32)))
        self.forward_precision = "fp32" if fp32 else "fp64"

    def set_stream(self, hip_stream: int):
        """Use the caller's HIP stream (0 = the default stream, e.g. torch.cuda.current_stream().cuda_stream)."""
        check(self._lib.htm_forward_set_stream(self.handle, C.c_void_p(int(hip_stream))))

    def reset_stream(self):
        check(self._lib.htm_forward_reset_stream(self.handle))

    def sync(self):
        check(self._lib.htm_forward_sync(self.handle))

    # -- reference bound procedures -------------------------------------------------------------------
    def calc_log_likelihood(self, hypo, t_corr, vs, a_corr, qs) -> float:
        out = C.c_double()
        check(self._lib.htm_forward_loglik_full(self.handle, _p(_arr(hypo, 3 * self.n_events)),
                                                _p(_arr(t_corr, self.n_sta)), float(vs),
                                                _p(_arr(a_corr, self.n_sta)), float(qs), C.byref(out)))
        return out.value

    def partially_update_log_likelihood(self, evt_id, hypo_old, log_likelihood_old, hypo, t_corr, vs, a_corr,
                                        qs) -> float:
        ho = _arr(hypo_old, 3 * self.n_events)
        hn = _arr(hypo, 3 * self.n_events)
        e = int(evt_id)
        if not 1 <= e <= self.n_events:
            raise ValueError(f"evt_id {e} out of range 1..{self.n_events}")
        xo = np.ascontiguousarray(ho[3 * (e - 1):3 * e])
        xn = np.ascontiguousarray(hn[3 * (e - 1):3 * e])
        out = C.c_double()
        check(self._lib.htm_forward_loglik_partial(self.handle, e, _p(xo), float(log_likelihood_old), _p(xn),
                                                   _p(_arr(t_corr, self.n_sta)), float(vs),
                                                   _p(_arr(a_corr, self.n_sta)), float(qs), C.byref(out)))
        return out.value

    def calc_travel_time(self, hypo, t_corr, vs) -> np.ndarray:
        out = np.empty((self.n_events, self.n_sta))
        check(self._lib.htm_forward_travel_time(self.handle, _p(_arr(hypo, 3 * self.n_events)),
                                                _p(_arr(t_corr, self.n_sta)), float(vs), _p(out)))
        return out

    def calc_amp(self, hypo, a_corr, qs, vs) -> np.ndarray:
        out = np.empty((self.n_events, self.n_sta))
        check(self._lib.htm_forward_amp(self.handle, _p(_arr(hypo, 3 * self.n_events)),
                                        _p(_arr(a_corr, self.n_sta)), float(qs), float(vs), _p(out)))
        return out

    def calc_travel_time_single(self, evt_id, hypo, t_corr, vs) -> np.ndarray:
        out = np.empty(self.n_sta)
        check(self._lib.htm_forward_travel_time_single(self.handle, int(evt_id), _p(_arr(hypo, 3 * self.n_events)),
                                                       _p(_arr(t_corr, self.n_sta)), float(vs), _p(out)))
        return out

    def calc_amp_single(self, evt_id, hypo, a_corr, qs, vs) -> np.ndarray:
        out = np.empty(self.n_sta)
        check(self._lib.htm_forward_amp_single(self.handle, int(evt_id), _p(_arr(hypo, 3 * self.n_events)),
                                               _p(_arr(a_corr, self.n_sta)), float(qs), float(vs), _p(out)))
        return out

    # -- batched extension ----------------------------------------------------------------------------
    def calc_log_likelihood_batch(self, hypo, t_corr, vs, a_corr, qs) -> np.ndarray:
        vs = _arr(vs)
        n = vs.size
        out = np.empty(n)
        check(self._lib.htm_forward_loglik_full_batch(self.handle, n, _p(_arr(hypo, n * 3 * self.n_events)),
                                                      _p(_arr(t_corr, n * self.n_sta)), _p(vs),
                                                      _p(_arr(a_corr, n * self.n_sta)), _p(_arr(qs, n)), _p(out)))
        return out

    def calc_log_likelihood_batch_dev(self, n, d_hypo, d_t_corr, d_vs, d_a_corr, d_qs, d_out):
        """All arguments are device pointers (ints); asynchronous on the handle's stream."""
        check(self._lib.htm_forward_loglik_full_batch_dev(self.handle, int(n), d_hypo, d_t_corr, d_vs, d_a_corr,
                                                          d_qs, d_out))

    def time_full_batch_dev(self, n, d_hypo, d_t_corr, d_vs, d_a_corr, d_qs, d_out, reps) -> float:
        us = C.c_double()
        check(self._lib.htm_forward_time_full_batch_dev(self.handle, int(n), d_hypo, d_t_corr, d_vs, d_a_corr,
                                                        d_qs, d_out, int(reps), C.byref(us)))
        return us.value
