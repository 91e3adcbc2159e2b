"""Run ON THE GPU BOX: long default-regime runs against the oracle (end state, counters, RNG position, every record).
python tools/soak_parity.py [n_iter_1000x64] [n_iter_64x64]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from hypotremormcmc_amd import driver, synth
from hypotremormcmc_amd.obs_data import ObsData
from oracle import oracle

n1 = int(sys.argv[1]) if len(sys.argv) > 1 else 60000
n2 = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
bad = 0
# (16 chains: two master workgroups; 4 and 5 chains: the rings that grew from 256 to 512 positions)
for (E, S, nc, n_iter, seed) in ((1000, 64, 8, n1, 7), (64, 64, 8, n2, 8), (64, 64, 5, n2, 9), (1000, 64, 16, n1 // 2, 11), (64, 64, 16, n2, 12), (64, 64, 4, n2, 13)):
    data = synth.make_synthetic(E, S, seed)
    params = dict(synth.DEFAULT_PARAMS, n_procs=1, n_chains=nc, n_cool=2, n_iter=n_iter, n_burn=n_iter // 2, n_interval=50)
    t0 = time.time()
    job = oracle.Job(params, data); job.run(n_iter)
    t1 = time.time()
    obs = ObsData.from_arrays(data.sta_x, data.sta_y, data.t_obs, data.t_stdv, data.a_obs, data.a_stdv)
    fwd, cs = driver.build_rank(params, data.sta_x, data.sta_y, data.sta_z, obs, 0, n_procs=1)
    cs.run(n_iter)
    t2 = time.time()
    it, lk = job.likelihood_trace(0)
    gi, _, gl = cs.likelihood_trace()
    a, b = cs.counts(); oa, ob = job.counts()
    ok = (np.array_equal(gi, it) and np.allclose(gl, lk, rtol=1e-9, atol=0) and cs.rng_state() == job.rng_state(0)
          and np.array_equal(a, oa) and np.array_equal(b, ob))
    bad += 0 if ok else 1
    print("%5d x %3d, %d chains, %d iterations: %s  (oracle %.1f s, gpu %.1f s, %d records)" % (E, S, nc, n_iter, "equal" if ok else "MISMATCH", t1 - t0, t2 - t1, len(it)), flush=True)
    del cs, fwd
sys.exit(1 if bad else 0)
