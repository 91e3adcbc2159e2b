"""Step 4 on the HIP path: `hypo_tremor_select` (reference src/hypo_tremor_select.f90, src/cls_selector.f90,
src/mod_regress.f90) -- the producer of the `selected_win.dat` step 5 reads.

    python -m hypotremormcmc_amd.select <parameter file>

Same inputs in the working directory (station file, `detected_win.dat`, `opt_data.NNNNNN.dat`), same outputs:
`regress.dat` (window id, vs, b, t0, a0, cc_t, cc_a: src/hypo_tremor_select.f90:124-125) and `selected_win.dat` (the
windows with vs_min <= vs <= vs_max and b_min <= b <= b_max, :126-131).  The regressions run on the GPU
(`htm_select_regress`, one wavefront per window); there is no CPU fallback.  `dist_plot.NNNNNN.dat`
(src/cls_selector.f90:106-111, a plotting aid nothing reads) is not written.
"""
from __future__ import annotations

import ctypes as C
import os
import sys

import numpy as np

from . import _lib
from ._lib import check, dp
from .obs_data import ObsData
from .param import Param


def _p(a):
    return a.ctypes.data_as(dp)


def regress(sta_x, sta_y, sta_z, z_guess, t, t_err, a, a_err, device=0):
    """rows {vs, b, t0, a0, cc_t, cc_a} per window; t, t_err, a, a_err of shape (n_win, n_sta)
    (src/cls_selector.f90:75-132)."""
    t = np.ascontiguousarray(t, dtype=np.float64)
    n_win, n_sta = t.shape
    arrs = [np.ascontiguousarray(v, dtype=np.float64) for v in (sta_x, sta_y, sta_z, t, t_err, a, a_err)]
    for v in arrs[:3]:
        if v.size != n_sta:
            raise ValueError("station arrays must have n_sta entries")
    out = np.empty((n_win, 6))
    check(_lib.load().htm_select_regress(int(device), n_sta, n_win, _p(arrs[0]), _p(arrs[1]), _p(arrs[2]), float(z_guess),
                                         _p(arrs[3]), _p(arrs[4]), _p(arrs[5]), _p(arrs[6]), _p(out)))
    return out


def select(reg, vs_min, vs_max, b_min, b_max):
    """src/hypo_tremor_select.f90:126-129: which windows go to step 5"""
    vs, b = reg[:, 0], reg[:, 1]
    return (vs >= vs_min) & (vs <= vs_max) & (b >= b_min) & (b <= b_max)


def read_detected_win(path="detected_win.dat"):
    if not os.path.exists(path):
        raise SystemExit("ERROR: detected_win.dat is not found")      # src/hypo_tremor_select.f90:50-52
    ids, times = [], []
    with open(path) as f:
        for line in f:
            tok = line.split()
            if len(tok) >= 2:
                ids.append(int(tok[0])); times.append(float(tok[1].lower().replace("d", "e")))
    return ids, times


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    if len(argv) != 1:
        raise SystemExit("USAGE: hypo_tremor_select [parameter file]")
    para = Param(argv[0], verb=True, from_where="select")
    win_id, win_t = read_detected_win()
    obs = ObsData(win_id, para.n_stations, para.sta_x, para.sta_y, verb=False)
    g = para.values
    reg = regress(para.sta_x, para.sta_y, para.sta_z, g["z_guess"], obs.get_t_obs(), obs.get_t_stdv(), obs.get_a_obs(),
                  obs.get_a_stdv(), device=int(os.environ.get("HTM_DEVICE", "0")))
    keep = select(reg, g["vs_min"], g["vs_max"], g["b_min"], g["b_max"])
    with open("regress.dat", "w") as f:
        for i, r in zip(win_id, reg):
            f.write(" %d %s\n" % (i, " ".join("%.17g" % v for v in r)))
    with open("selected_win.dat", "w") as f:
        for i, tt, k in zip(win_id, win_t, keep):
            if k:
                f.write(" %d %.17g\n" % (i, tt))


if __name__ == "__main__":
    main()
