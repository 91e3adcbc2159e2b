"""Run ON THE GPU BOX: rejection-heavy runs against the oracle (Rayleigh-prior rejections shift stream positions, the
validation repeats passes, orders sent ahead by role P miss their step).  python tools/stress_rejections.py [n_iter] [a:b] [sz,sz,...] [seed offset]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from hypotremormcmc_amd import driver, synth
from hypotremormcmc_amd.obs_data import ObsData
from oracle import oracle

n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
bad = 0
SEED0 = int(sys.argv[4]) if len(sys.argv) > 4 else 0      # another set of data sets and random streams
SHAPES = ((64, 64, 8), (1000, 64, 8), (200, 64, 3), (30, 20, 7), (100, 64, 16), (64, 32, 27), (64, 64, 16), (300, 128, 13))      # (9..16 chains: two master workgroups)
if len(sys.argv) > 2:        # "a:b": a slice of the shapes
    a_, b_ = (int(x) for x in sys.argv[2].split(":"))
    SHAPES = SHAPES[a_:b_]
for (E, S, nc) in SHAPES:
    SZS = ((1, 4.0), (2, 8.0), (3, 12.0), (4, 20.0))
    if len(sys.argv) > 3:    # "0.4,1.0": depth step sizes to run instead
        SZS = tuple((10 + k, float(x)) for k, x in enumerate(sys.argv[3].split(",")))
    for seed, sz in SZS:
        data = synth.make_synthetic(E, S, 100 + seed + SEED0)
        params = dict(synth.DEFAULT_PARAMS, n_procs=1, n_chains=nc, n_cool=2 if nc > 2 else 1, n_iter=n_iter, n_burn=n_iter // 2,
                      n_interval=3, step_size_z=sz, step_size_vs=0.4)
        t0 = time.time()
        job = oracle.Job(params, data); job.run(n_iter)
        obs = ObsData.from_arrays(data.sta_x, data.sta_y, data.t_obs, data.t_stdv, data.a_obs, data.a_stdv)
        fwd, cs = driver.build_rank(params, data.sta_x, data.sta_y, data.sta_z, obs, 0, n_procs=1)
        try:
            cs.run(n_iter)
            it, lk = job.likelihood_trace(0)
            gi, _, gl = cs.likelihood_trace()
            a, b = cs.counts(); oa, ob = job.counts()
            ok = (np.array_equal(gi, it) and np.allclose(gl, lk, rtol=1e-9, atol=0) and cs.rng_state() == job.rng_state(0)
                  and np.array_equal(a, oa) and np.array_equal(b, ob))
            msg = "ok" if ok else "MISMATCH iters %s trace %s rng %s counts %s first bad record %s" % (
                np.array_equal(gi, it), len(gl) == len(lk) and np.allclose(gl, lk, rtol=1e-9, atol=0), cs.rng_state() == job.rng_state(0),
                np.array_equal(a, oa) and np.array_equal(b, ob),
                (int(np.argmax(~np.isclose(gl, lk, rtol=1e-9, atol=0))), ) if len(gl) == len(lk) else None)
        except Exception as e:      # noqa: BLE001
            ok, msg = False, "ERROR %s" % e
        bad += 0 if ok else 1
        print("%5d x %3d, %d chains, step_size_z %5.1f: %s  (%d records, %.1f s)" % (E, S, nc, sz, msg, len(job.likelihood_trace(0)[0]), time.time() - t0), flush=True)
        del cs, fwd
print("FAILED %d" % bad if bad else "all equal")
sys.exit(1 if bad else 0)
