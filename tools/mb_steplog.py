"""Run ON THE GPU BOX: the one stress configuration two master workgroups have differed on (tools/mb_repro.py), with the step log
on, several runs against the oracle's step log -- WHAT the first differing step did (type, element, prior check, decision, values).
   HTM_MB=1 python tools/mb_steplog.py [runs n_iter]"""
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from hypotremormcmc_amd import driver, synth
from hypotremormcmc_amd.obs_data import ObsData
from oracle import oracle

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 6
n_iter = int(sys.argv[2]) if len(sys.argv) > 2 else 9000
E, S, nc, seed, sz = 100, 64, 16, 4, 20.0
data = synth.make_synthetic(E, S, 100 + seed)
params = dict(synth.DEFAULT_PARAMS, n_procs=1, n_chains=nc, n_cool=2, n_iter=n_iter, n_burn=n_iter // 2, n_interval=3, step_size_z=sz, step_size_vs=0.4)
job = oracle.Job(params, data); job.enable_steplog(n_iter * nc + 16); job.run(n_iter)
oi, od = job.steplog()
oi = np.concatenate([oi[:, 0:1], oi[:, 2:8], np.zeros((len(oi), 1), np.int32)], axis=1)      # (oracle rows carry the rank: gpu layout)
print("oracle rows", oi.shape, "prior rejections", int((oi[:, 4] == 0).sum()), flush=True)
obs = ObsData.from_arrays(data.sta_x, data.sta_y, data.t_obs, data.t_stdv, data.a_obs, data.a_stdv)


def show(tag, ir, dr, k):
    print("   %s it %d c %2d type %d idx %3d ok %d acc %d full %d | x_new %.9g L_new %.10e L_post %.10e T %.6g"
          % (tag, ir[k, 0], ir[k, 1], ir[k, 2], ir[k, 3], ir[k, 4], ir[k, 5], ir[k, 6], dr[k, 0], dr[k, 1], dr[k, 2], dr[k, 3]))


for r in range(runs):
    fwd, cs = driver.build_rank(params, data.sta_x, data.sta_y, data.sta_z, obs, 0, n_procs=1)
    cs.enable_steplog(n_iter * nc + 16)
    cs.run(n_iter)
    gi, gd = cs.steplog()
    n = min(len(gi), len(oi))
    hist = gi[:, 6].astype(np.int64) >> 4; gi[:, 6] &= 1      # (a -DHTM_MB_DIAG library: the step's history above the need_full bit)
    # (rows in (iteration, chain) order on both sides; column 7 is not compared)
    bad_i = np.nonzero((gi[:n, :7] != oi[:n, :7]).any(axis=1))[0]
    bad_d = np.nonzero(~np.isclose(gd[:n, 2], od[:n, 2], rtol=1e-9, atol=0))[0]
    first = min([int(b[0]) for b in (bad_i, bad_d) if len(b)] or [-1])
    print("run %d loop %s rows %d/%d: integer rows that differ %d, L_post that differ %d, first %d" %
          (r, cs.master_stats()["single_rank_loop"], len(gi), len(oi), len(bad_i), len(bad_d), first), flush=True)
    if first >= 0:
        it0 = int(oi[first, 0])
        for k in range(n):
            if it0 - 2 <= oi[k, 0] <= it0 and (oi[k, 4] == 0 or k == first or oi[k, 1] == oi[first, 1]):
                show("oracle", oi, od, k); show("gpu   ", gi, gd, k)
        # where the steps around it started (gpu rows, column 7: position | epoch << 24) and how many draws each took
        for k in range(max(0, first - 20), min(n, first + 4)):
            print("      gpu it %d c %2d type %d ok %d  p %8d epoch %3d  history %05x  p - W.rpos %3d%s" % (gi[k, 0], gi[k, 1], gi[k, 2], gi[k, 4], gi[k, 7] & 0xffffff, (gi[k, 7] >> 24) & 0xff,
                                                                                        hist[k] & 0xfffff, (hist[k] >> 20) & 0x7f, "   <--" if k == first else ""))
    del cs, fwd
