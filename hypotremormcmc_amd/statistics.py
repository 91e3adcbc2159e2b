"""Step 6 of the pipeline on the GPU: the build's counterpart of `hypo_tremor_statistics`
(reference src/hypo_tremor_statistics.f90, src/cls_statistics.f90).

The reference gathers every rank's recorded samples on rank 0, sorts each parameter's column with quick_sort
and prints three elements of it.  Here the three elements come from `htm_quantiles` (exact radix select on the
device, hypotremormcmc_amd/csrc/htm_select.hpp); files, names and formats are the reference's:

    uniform_structure.stat      Vs / Qs                      (src/cls_statistics.f90:394-431)
    station_corrections.stat    t_corr / a_corr per station  (:345-390)
    hypo.stat                   x, y, z per window           (:216-264)
    hypo.stat.removed           ... without double counts    (:120-211)

    python -m hypotremormcmc_amd.statistics <parameter file>      # in the directory of the step-5 outputs
"""
from __future__ import annotations

import ctypes as C
import os
import sys

import numpy as np

from . import _lib
from .param import Param

HYPO_HEADER = ("# window ID, x (50%), x (2.5%) x (97.5%), y (50%), y (2.5%), y (97.5%)"
               "z (50 %), z (2.5%), z (97.5%)")
CORR_HEADER = ("# station name, t_corr (50%), t_corr (2.5%) t_corr (97.5%), a_corr (50%), a_corr (2.5%), "
               "a_corr (97.5%)")
VQ_HEADER = "# Vs (50%), Vs (2.5%) Vs (97.5%), Qs (50%), Qs (2.5%), Qs (97.5%)"


def expected_n_mod(n_iter, n_burn, n_procs, n_cool, n_interval) -> int:
    """src/cls_statistics.f90:64 (integer arithmetic, evaluated left to right)"""
    return (n_iter - n_burn) * n_procs * n_cool // n_interval


def ranks(n_mod: int):
    """il, im, iu of src/cls_statistics.f90:229-231: single-precision products truncated to integer"""
    f = np.float32
    return tuple(int(f(c) * f(n_mod)) for c in (0.025, 0.5, 0.975))


def quantiles(samples: np.ndarray, n_mod: int | None = None, device: int = 0) -> np.ndarray:
    """[n_par][3] = (il-th, im-th, iu-th smallest) of every column of samples [n_rows][n_par], on the GPU."""
    x = np.ascontiguousarray(samples, dtype=np.float64)
    if x.ndim == 1:
        x = x[:, None]
    n_rows, n_par = x.shape
    il, im, iu = ranks(n_rows if n_mod is None else n_mod)
    rk = (C.c_int * 3)(il, im, iu)
    out = np.empty((n_par, 3))
    lib = _lib.load()
    _lib.check(lib.htm_quantiles(device, x.ctypes.data_as(_lib.dp), n_rows, n_par, rk, out.ctypes.data_as(_lib.dp)))
    return out          # columns: il (2.5 %), im (50 %), iu (97.5 %)


def read_sample_file(path: str, n_val: int, endian=None):
    """stream-unformatted records [int32 iteration][n_val float64] (src/hypo_tremor_mcmc.f90:216-233);
    byte order: HTM_SAMPLE_ENDIAN=big|little (driver.sample_byte_order; big = written by a stock-gfortran build of
    the reference, src/Makefile:7-9)"""
    from .driver import sample_byte_order
    bo = sample_byte_order(endian)
    rec = np.dtype([("it", bo + "i4"), ("v", bo + "f8", (n_val,))])
    a = np.fromfile(path, dtype=rec)
    return a["it"].astype(np.int32), a["v"].reshape(len(a), n_val).astype(np.float64)


def remove_double_counts(win_id, q):
    """src/cls_statistics.f90:150-185.  q[i] = (x, x_lo, x_hi, y, y_lo, y_hi, z, z_lo, z_hi) of window i.
    Repeatedly drops a window whose id follows its predecessor's (in the current list) when both medians lie
    strictly inside the intersection of the two 95 % boxes.  Returns the indices kept."""
    keep = list(range(len(win_id)))
    while True:
        new, flag = [keep[0]], False
        for j in range(1, len(keep)):
            i, ii = keep[j], keep[j - 1]
            drop = False
            if win_id[i] == win_id[ii] + 1:
                inside = True
                for c in range(3):
                    med_i, lo_i, hi_i = q[i][3 * c:3 * c + 3]
                    med_p, lo_p, hi_p = q[ii][3 * c:3 * c + 3]
                    lo, hi = max(lo_i, lo_p), min(hi_i, hi_p)
                    inside = inside and lo < med_i and lo < med_p and hi > med_i and hi > med_p
                drop = inside
            if drop:
                flag = True
            else:
                new.append(i)
        keep = new
        if not flag:
            return keep


class Statistics:
    """`type statistics` of the reference: constructor arguments as src/cls_statistics.f90:50-53."""

    def __init__(self, n_procs, n_iter, n_burn, n_interval, n_cool, station_names, win_id, device=0):
        self.n_procs = int(n_procs)
        self.n_mod = expected_n_mod(int(n_iter), int(n_burn), self.n_procs, int(n_cool), int(n_interval))
        self.station_names = [str(s) for s in station_names]
        self.win_id = [int(w) for w in win_id]
        self.device = device
        if ranks(self.n_mod)[0] < 1:
            raise ValueError(f"n_mod = {self.n_mod}: il = int(0.025 * n_mod) = 0, the reference reads outside its "
                             f"sorted column; record at least 40 samples")

    # -- the three estimators; `*_samples` = all ranks' records stacked in rank order -----------------------
    def _q(self, samples):
        samples = np.asarray(samples, dtype=np.float64)
        if samples.shape[0] != self.n_mod:
            raise ValueError(f"{samples.shape[0]} samples recorded, the reference expects n_mod = {self.n_mod} "
                             f"(= (n_iter - n_burn) * n_procs * n_cool / n_interval)")
        return quantiles(samples, self.n_mod, self.device)

    def estimate_vs_qs(self, vs_samples, qs_samples, out_dir="."):
        v, q = self._q(vs_samples)[0], self._q(qs_samples)[0]
        with open(os.path.join(out_dir, "uniform_structure.stat"), "w") as fh:
            fh.write(VQ_HEADER + "\n")
            fh.write("".join("%13.6f" % x for x in (v[1], v[0], v[2], q[1], q[0], q[2])) + "\n")

    def estimate_corr_factors(self, t_corr_samples, a_corr_samples, out_dir="."):
        t, a = self._q(t_corr_samples), self._q(a_corr_samples)
        with open(os.path.join(out_dir, "station_corrections.stat"), "w") as fh:
            fh.write(CORR_HEADER + "\n")
            for k, name in enumerate(self.station_names):
                fh.write("%12s" % name.strip()[:12] +
                         "".join("%13.6f" % x for x in (t[k][1], t[k][0], t[k][2], a[k][1], a[k][0], a[k][2])) + "\n")

    def estimate_hypo(self, hypo_samples, out_dir="."):
        h = self._q(hypo_samples)                               # [3E][3], x y z interleaved per window
        rows = []
        for i in range(len(self.win_id)):
            r = []
            for c in range(3):
                lo, med, hi = h[3 * i + c]
                r += [med, lo, hi]
            rows.append(r)
        self._write_hypo(os.path.join(out_dir, "hypo.stat"), range(len(rows)), rows)
        # the reference re-reads the F13.6 text it just wrote before removing double counts (:136-142)
        rounded = [[float("%13.6f" % x) for x in r] for r in rows]
        keep = remove_double_counts(self.win_id, rounded)
        self._write_hypo(os.path.join(out_dir, "hypo.stat.removed"), keep, rows)

    def _write_hypo(self, path, idx, rows):
        with open(path, "w") as fh:
            fh.write(HYPO_HEADER + "\n")
            for i in idx:
                fh.write("%9d" % self.win_id[i] + "".join("%13.6f" % x for x in rows[i]) + "\n")


def gather_output_files(work_dir, n_procs, n_sta, n_events):
    """every rank's vs/qs/t_corr/a_corr/hypo sample files, stacked in rank order (the reference's gather)"""
    out = {}
    for nm, nv in (("vs", 1), ("qs", 1), ("t_corr", n_sta), ("a_corr", n_sta), ("hypo", 3 * n_events)):
        parts = [read_sample_file(os.path.join(work_dir, "%s.%02d.out" % (nm, r)), nv)[1] for r in range(n_procs)]
        out[nm] = np.concatenate(parts, axis=0)
    return out


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    if len(argv) != 1:
        raise SystemExit("USAGE: python -m hypotremormcmc_amd.statistics [parameter file]")
    par = Param(argv[0])
    work = os.path.dirname(os.path.abspath(argv[0]))
    win_id = [int(ln.split()[0]) for ln in open(os.path.join(work, "selected_win.dat")) if ln.strip()]
    st = Statistics(par.get_n_procs(), par.get_n_iter(), par.get_n_burn(), par.get_n_interval(), par.get_n_cool(),
                    par.stations, win_id)
    s = gather_output_files(work, st.n_procs, par.n_stations, len(win_id))
    st.estimate_vs_qs(s["vs"], s["qs"], work)
    st.estimate_corr_factors(s["t_corr"], s["a_corr"], work)
    st.estimate_hypo(s["hypo"], work)


if __name__ == "__main__":
    main()
