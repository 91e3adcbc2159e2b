"""Run ON THE GPU BOX: one rejection-heavy configuration on two master workgroups, several times, against the oracle -- where the
first record that differs lies.   python tools/mb_repro.py [E S nc seed sz n_iter runs]"""
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from hypotremormcmc_amd import driver, synth
from hypotremormcmc_amd.obs_data import ObsData
from oracle import oracle

a = sys.argv[1:]
E, S, nc, seed = (int(a[k]) if len(a) > k else d for k, d in enumerate((100, 64, 16, 4)))
sz = float(a[4]) if len(a) > 4 else 20.0
n_iter = int(a[5]) if len(a) > 5 else 20000
runs = int(a[6]) if len(a) > 6 else 4
data = synth.make_synthetic(E, S, 100 + seed)
params = dict(synth.DEFAULT_PARAMS, n_procs=1, n_chains=nc, n_cool=2, n_iter=n_iter, n_burn=n_iter // 2, n_interval=3, step_size_z=sz, step_size_vs=0.4)
job = oracle.Job(params, data); job.run(n_iter)
it, lk = job.likelihood_trace(0)
obs = ObsData.from_arrays(data.sta_x, data.sta_y, data.t_obs, data.t_stdv, data.a_obs, data.a_stdv)
for r in range(runs):
    fwd, cs = driver.build_rank(params, data.sta_x, data.sta_y, data.sta_z, obs, 0, n_procs=1)
    cs.run(n_iter)
    gi, gc, gl = cs.likelihood_trace()
    same = np.array_equal(gi, it)
    bad = np.nonzero(~np.isclose(gl, lk, rtol=1e-9, atol=0))[0] if len(gl) == len(lk) else np.array([0])
    print("run %d loop %s: iters equal %s, rng equal %s, records that differ %d of %d" % (r, cs.master_stats()["single_rank_loop"], same, cs.rng_state() == job.rng_state(0), len(bad), len(lk)), flush=True)
    if len(bad):
        k = int(bad[0])
        for j in range(max(0, k - 3), min(len(lk), k + 4)):
            print("   record %d: iteration %d chain %d   gpu %.10e   oracle %.10e %s" % (j, gi[j], gc[j], gl[j], lk[j], "<-- first" if j == k else ""))
    del cs, fwd
