"""Run ON THE GPU BOX: step 4's regressions (htm_select_regress: one wavefront per detected window, host pointers,
synchronous -- so the figure includes the PCIe copies) beside the CPU restatement of the reference's loop
(src/cls_selector.f90:75-132, one core), on synthetic windows; results compared.  python tools/bench_regress.py"""
import os
import sys
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import numpy as np

from hypotremormcmc_amd import select
from oracle import oracle

rng = np.random.default_rng(3)
for n_win, n_sta in ((1000, 64), (20000, 64), (100000, 128)):
    sx, sy = rng.uniform(-50, 50, n_sta), rng.uniform(-50, 50, n_sta)
    sz = rng.uniform(-1, 0, n_sta)
    ex, ey = rng.uniform(-30, 30, n_win), rng.uniform(-30, 30, n_win)
    d = np.sqrt((ex[:, None] - sx) ** 2 + (ey[:, None] - sy) ** 2 + (30.0 - sz) ** 2)
    t = d / 3.5 + rng.normal(0, 0.3, d.shape)
    a = -np.log(d) - 0.02 * d + rng.normal(0, 0.2, d.shape)
    te, ae = rng.uniform(0.1, 0.5, d.shape), rng.uniform(0.1, 0.5, d.shape)
    select.regress(sx, sy, sz, 30.0, t[:8], te[:8], a[:8], ae[:8])          # warm-up (library load, first launch)
    t0 = time.perf_counter(); g = select.regress(sx, sy, sz, 30.0, t, te, a, ae); tg = time.perf_counter() - t0
    t0 = time.perf_counter(); c = oracle.select_regress(sx, sy, sz, 30.0, t, te, a, ae); tc = time.perf_counter() - t0
    err = float(np.max(np.max(np.abs(g - c), axis=0) / np.max(np.abs(c), axis=0)))     # per output column, against its largest value
    mb = 4 * 8 * n_win * n_sta / 1e6
    print("%6d windows x %3d stations (%.0f MB of measurements): GPU call %.2f ms (%.0f k windows/s, PCIe copies included), "
          "CPU restatement on one core %.1f ms (%.0f k windows/s); largest difference relative to a column's range %.1e"
          % (n_win, n_sta, mb, 1e3 * tg, n_win / tg / 1e3, 1e3 * tc, n_win / tc / 1e3, err), flush=True)
