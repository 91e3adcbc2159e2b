"""Device-resident MCMC chains of one rank: the `type mcmc` objects held by `type parallel`
(reference src/cls_mcmc.f90:7-53, src/cls_parallel.f90:7-22) plus the main loop body of
src/hypo_tremor_mcmc.f90:236-284, executed by the HIP library."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib
from ._lib import ChainsInit, ModelInit, check, dp, ip
from .forward import Forward
from .model import Model

LABELS = ["vs", "t_corr", "qs", "a_corr", "x", "y", "z"]  # src/cls_mcmc.f90:83 (sic: see SURVEY quirk 2)


@dataclass
class ChainState:
    hypo: np.ndarray
    t_corr: np.ndarray
    vs: float
    a_corr: np.ndarray
    qs: float
    temp: float
    log_likelihood: float
    n_propose: np.ndarray
    n_accept: np.ndarray


def _stack(models, attr, dtype):
    return np.ascontiguousarray(np.stack([getattr(m, attr) for m in models]).astype(dtype))


class ChainSet:
    """n_chains chains of one rank.  `models` is a list (one entry per chain) of dicts with the five
    reference models: hypo, t_corr, vs, a_corr, qs (each a `Model`)."""

    def __init__(self, fwd: Forward, models, temps, rng_state, *, n_procs=1, rank=0, solve_vs=True,
                 solve_t_corr=True, solve_qs=True, solve_a_corr=True, n_burn=0, n_interval=1,
                 lik_capacity=0, sample_capacity=0):
        self._lib = _lib.load()
        self.fwd = fwd
        self.n_chains = len(models)
        self.n_procs, self.rank = int(n_procs), int(rank)
        self.n_sta, self.n_events = fwd.n_sta, fwd.n_events
        init = ChainsInit()
        init.n_chains, init.n_procs, init.rank = self.n_chains, self.n_procs, self.rank
        self._keep = []
        for name in ("hypo", "t_corr", "vs", "a_corr", "qs"):
            ms = [m[name] for m in models]
            mi = ModelInit()
            for attr, ctype, dt in (("x", dp, np.float64), ("mu", dp, np.float64), ("sigma", dp, np.float64),
                                    ("step_size", dp, np.float64), ("prior_type", ip, np.int32)):
                a = _stack(ms, attr, dt)
                self._keep.append(a)
                setattr(mi, attr, a.ctypes.data_as(ctype))
            setattr(init, name, mi)
        t = np.ascontiguousarray(np.asarray(temps, dtype=np.float64))
        self._keep.append(t)
        init.temp = t.ctypes.data_as(dp)
        init.solve_vs, init.solve_t_corr = int(bool(solve_vs)), int(bool(solve_t_corr))
        init.solve_qs, init.solve_a_corr = int(bool(solve_qs)), int(bool(solve_a_corr))
        for k in range(4):
            init.rng_state[k] = int(rng_state[k]) & 0xFFFFFFFF
        init.n_burn, init.n_interval = int(n_burn), int(n_interval)
        init.lik_capacity, init.sample_capacity = int(lik_capacity), int(sample_capacity)
        h = C.c_void_p()
        check(self._lib.htm_chains_create(fwd.handle, C.byref(init), C.byref(h)))
        self.handle = h

    def close(self):
        if getattr(self, "handle", None):
            self._lib.htm_chains_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- single-rank driver ---------------------------------------------------------------------
    def run(self, n_iter: int):
        check(self._lib.htm_chains_run(self.handle, int(n_iter)))

    # ---- lock-step pieces (multi-rank) ------------------------------------------------------------
    def step_begin(self):
        check(self._lib.htm_chains_step_begin(self.handle))

    def swap_record(self):
        """(device pointer, bytes) of this rank's swap record."""
        p = C.c_void_p()
        n = C.c_size_t()
        check(self._lib.htm_chains_swap_record(self.handle, C.byref(p), C.byref(n)))
        return p.value, n.value

    def step_end(self, d_gathered: int):
        check(self._lib.htm_chains_step_end(self.handle, C.c_void_p(d_gathered)))

    def swap_record_host(self, out: np.ndarray):
        """wait for the iteration and copy this rank's record into `out` (float64[4 + 2 n_chains], host)"""
        check(self._lib.htm_chains_swap_record_host(self.handle, out.ctypes.data_as(dp)))

    def step_end_host(self, gathered: np.ndarray):
        """all ranks' records from host memory (float64[n_procs * (4 + 2 n_chains)])"""
        check(self._lib.htm_chains_step_end_host(self.handle, gathered.ctypes.data_as(dp)))

    def run_lockstep(self, n_iter: int, allgather_fn: int, comm: int, d_gathered: int):
        """n_iter lock-step iterations driven from C (see htm_chains_run_lockstep): allgather_fn is the
        address of an ncclAllGather-compatible function, comm the rank's communicator handle."""
        check(self._lib.htm_chains_run_lockstep(self.handle, int(n_iter), C.c_void_p(allgather_fn),
                                                C.c_void_p(comm), C.c_void_p(d_gathered)))

    # ---- persistent lock-step: swap records exchanged inside the kernel (peer-mapped inboxes) -------------
    XCHG_HANDLE_BYTES = 64

    def xchg_handle(self) -> bytes:
        """IPC handle of this rank's inbox (to be all-gathered over the ranks)."""
        buf = C.create_string_buffer(self.XCHG_HANDLE_BYTES)
        check(self._lib.htm_chains_xchg_handle(self.handle, buf, self.XCHG_HANDLE_BYTES))
        return buf.raw

    def xchg_connect(self, handles: bytes | None):
        """handles = the n_procs handles in rank order, concatenated (None for a single-rank job)."""
        if handles is None:
            check(self._lib.htm_chains_xchg_connect(self.handle, None, 0))
        else:
            buf = C.create_string_buffer(handles, len(handles))
            check(self._lib.htm_chains_xchg_connect(self.handle, buf, self.XCHG_HANDLE_BYTES))

    def xchg_probe(self, token: int = 1, seconds: float = 10.0):
        """Collective: every rank's probe kernel must see the tokens of all ranks in its inbox (raises otherwise)."""
        check(self._lib.htm_chains_xchg_probe(self.handle, int(token), float(seconds)))

    def run_lockstep_direct(self, n_iter: int):
        check(self._lib.htm_chains_run_lockstep_direct(self.handle, int(n_iter)))

    def share_gpu(self, ranks_on_this_gpu: int):
        """several ranks' persistent launches must be resident on this GPU at once: this rank takes its share of the CUs
        (before the first run; HTM_RANKS_PER_GPU, if set, stands instead)"""
        check(self._lib.htm_chains_share_gpu(self.handle, int(ranks_on_this_gpu)))

    def sync(self):
        check(self._lib.htm_chains_sync(self.handle))

    def drain(self):
        check(self._lib.htm_chains_drain(self.handle))

    # ---- results ------------------------------------------------------------------------------------
    @property
    def iterations_done(self) -> int:
        n = C.c_int()
        check(self._lib.htm_chains_iterations_done(self.handle, C.byref(n)))
        return n.value

    def state(self, chain: int) -> ChainState:
        E, S = self.n_events, self.n_sta
        hypo = np.empty(3 * E); tc = np.empty(S); ac = np.empty(S)
        vs = C.c_double(); qs = C.c_double(); temp = C.c_double(); ll = C.c_double()
        npr = np.zeros(7, np.int32); nac = np.zeros(7, np.int32)
        check(self._lib.htm_chains_get_state(self.handle, int(chain), hypo.ctypes.data_as(dp), tc.ctypes.data_as(dp),
                                             C.cast(C.byref(vs), dp), ac.ctypes.data_as(dp), C.cast(C.byref(qs), dp),
                                             C.cast(C.byref(temp), dp), C.cast(C.byref(ll), dp),
                                             npr.ctypes.data_as(ip), nac.ctypes.data_as(ip)))
        return ChainState(hypo, tc, vs.value, ac, qs.value, temp.value, ll.value, npr, nac)

    def checkpoint(self) -> bytes:
        """Everything the main loop carries between iterations (htm_chains_checkpoint_save): iteration counter,
        mod_random state, parameter vectors, temperatures, log-likelihoods, proposal counters."""
        n = C.c_size_t()
        check(self._lib.htm_chains_checkpoint_size(self.handle, C.byref(n)))
        buf = C.create_string_buffer(n.value)
        check(self._lib.htm_chains_checkpoint_save(self.handle, buf, n.value))
        return buf.raw

    def restore(self, blob: bytes):
        """Load a checkpoint into a chain set created from the same inputs; continuing gives the bits of the
        uninterrupted run.  Records held so far are dropped."""
        buf = C.create_string_buffer(blob, len(blob))
        check(self._lib.htm_chains_checkpoint_load(self.handle, buf, len(blob)))

    def rng_state(self):
        st = (C.c_uint32 * 4)()
        check(self._lib.htm_chains_get_rng(self.handle, st))
        return tuple(int(v) for v in st)

    def likelihood_trace(self):
        """(iter, chain, log-likelihood) exactly in the order the reference appends to likelihoodRR.out."""
        n = C.c_int()
        check(self._lib.htm_chains_lik_count(self.handle, C.byref(n)))
        it = np.empty(n.value, np.int32); ch = np.empty(n.value, np.int32); lk = np.empty(n.value)
        if n.value:
            check(self._lib.htm_chains_lik_read(self.handle, it.ctypes.data_as(ip), ch.ctypes.data_as(ip),
                                                lk.ctypes.data_as(dp)))
        return it, ch, lk

    def samples(self):
        n = C.c_int()
        check(self._lib.htm_chains_sample_count(self.handle, C.byref(n)))
        n = n.value
        E, S = self.n_events, self.n_sta
        out = dict(iter=np.empty(n, np.int32), chain=np.empty(n, np.int32), vs=np.empty(n), qs=np.empty(n),
                   hypo=np.empty((n, 3 * E)), t_corr=np.empty((n, S)), a_corr=np.empty((n, S)))
        for k in range(n):
            it = C.c_int32(); ch = C.c_int32(); vs = C.c_double(); qs = C.c_double()
            check(self._lib.htm_chains_sample_read(self.handle, k, C.byref(it), C.byref(ch),
                                                   C.cast(C.byref(vs), dp), C.cast(C.byref(qs), dp),
                                                   out["hypo"][k].ctypes.data_as(dp),
                                                   out["t_corr"][k].ctypes.data_as(dp),
                                                   out["a_corr"][k].ctypes.data_as(dp)))
            out["iter"][k] = it.value; out["chain"][k] = ch.value; out["vs"][k] = vs.value; out["qs"][k] = qs.value
        return out

    def clear_records(self):
        check(self._lib.htm_chains_clear_records(self.handle))

    def counts(self):
        """Sum of the proposal / acceptance counters over this rank's chains (src/cls_parallel.f90:259-263)."""
        npr = np.zeros(7, np.int64); nac = np.zeros(7, np.int64)
        for c in range(self.n_chains):
            s = self.state(c)
            npr += s.n_propose; nac += s.n_accept
        return npr, nac

    def enable_steplog(self, capacity: int):
        check(self._lib.htm_chains_enable_steplog(self.handle, int(capacity)))

    def steplog(self):
        n = C.c_int()
        check(self._lib.htm_chains_steplog_read(self.handle, C.byref(n), None, None))
        ir = np.empty((n.value, 8), np.int32); dr = np.empty((n.value, 4))
        if n.value:
            check(self._lib.htm_chains_steplog_read(self.handle, C.byref(n), ir.ctypes.data_as(ip),
                                                    dr.ctypes.data_as(dp)))
        return ir, dr

    def profile(self, n_iter: int):
        """Run n_iter iterations with every kernel launch bracketed by HIP events (see htm_chains_profile)."""
        su = C.c_double(); fu = C.c_double(); sn = C.c_int(); fn = C.c_int(); fe = C.c_int64(); pe = C.c_int64()
        check(self._lib.htm_chains_profile(self.handle, int(n_iter), C.byref(su), C.byref(sn), C.byref(fu),
                                           C.byref(fn), C.byref(fe), C.byref(pe)))
        return dict(step_us=su.value, step_launches=sn.value, full_us=fu.value, full_launches=fn.value,
                    full_evals=fe.value, partial_evals=pe.value)

    def handoff_stats(self):
        """orders worker block 0 put aside (named commit not visible within 20 us); 0 in a healthy run"""
        n = C.c_int64()
        check(self._lib.htm_chains_handoff_stats(self.handle, C.byref(n)))
        return dict(orders_put_aside=n.value)

    def master_stats(self):
        """which main loop the launches run on (5 / 6 pipelined master, 3 / 4 free-running, 0 / 2 barriers, -1 two kernels)
        and how often the pipelined master flushed its speculative records"""
        a = C.c_int(); b = C.c_int(); n = C.c_int64()
        check(self._lib.htm_chains_master_stats(self.handle, C.byref(a), C.byref(b), C.byref(n)))
        return dict(single_rank_loop=a.value, lockstep_loop=b.value, flushes=n.value)

    def last_run_stats(self):
        us = C.c_double(); g = C.c_int(); f = C.c_int64(); p = C.c_int64()
        check(self._lib.htm_chains_last_run_stats(self.handle, C.byref(us), C.byref(g), C.byref(f), C.byref(p)))
        return dict(device_us=us.value, graph_launches=g.value, full_evals=f.value, partial_evals=p.value)
