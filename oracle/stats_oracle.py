"""TEST INFRASTRUCTURE ONLY: CPU restatement of the reference's step-6 estimators (src/cls_statistics.f90).

Sorts every column (the reference: quick_sort, src/mod_sort.f90) and takes the elements il, im, iu of
src/cls_statistics.f90:229-231 -- single-precision products `0.025 * n_mod` etc. truncated to integer, 1-based.
Pinned against the .stat files the compiled reference step 6 wrote for the golden cases
(tests/golden/*.npz, keys stat_*); only tests may import this module.
"""
import numpy as np


def ranks(n_mod):
    f = np.float32
    return tuple(int(f(c) * f(n_mod)) for c in (0.025, 0.5, 0.975))


def quantiles(samples, n_mod=None):
    x = np.sort(np.asarray(samples, dtype=np.float64).reshape(len(samples), -1), axis=0)
    il, im, iu = ranks(len(x) if n_mod is None else n_mod)
    return np.stack([x[il - 1], x[im - 1], x[iu - 1]], axis=1)      # [n_par][3]


def f13(x):
    return "%13.6f" % x


def uniform_structure_text(vs, qs):
    v, q = quantiles(vs)[0], quantiles(qs)[0]
    return ("# Vs (50%), Vs (2.5%) Vs (97.5%), Qs (50%), Qs (2.5%), Qs (97.5%)\n" +
            "".join(f13(t) for t in (v[1], v[0], v[2], q[1], q[0], q[2])) + "\n")


def station_corrections_text(names, t_corr, a_corr):
    t, a = quantiles(t_corr), quantiles(a_corr)
    out = ["# station name, t_corr (50%), t_corr (2.5%) t_corr (97.5%), a_corr (50%), a_corr (2.5%), a_corr (97.5%)\n"]
    for k, nm in enumerate(names):
        out.append("%12s" % nm + "".join(f13(v) for v in (t[k][1], t[k][0], t[k][2], a[k][1], a[k][0], a[k][2])) + "\n")
    return "".join(out)


def hypo_text(win_id, hypo):
    h = quantiles(hypo)
    out = ["# window ID, x (50%), x (2.5%) x (97.5%), y (50%), y (2.5%), y (97.5%)z (50 %), z (2.5%), z (97.5%)\n"]
    for i, w in enumerate(win_id):
        vals = []
        for c in range(3):
            lo, med, hi = h[3 * i + c]
            vals += [med, lo, hi]
        out.append("%9d" % w + "".join(f13(v) for v in vals) + "\n")
    return "".join(out)
