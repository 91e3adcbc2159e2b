"""Soak runs (GPU box): long persistent run, long lock-step run, long 2-rank LocalWorld run, with end-state checks
against the oracle where that is affordable.  Prints one line per stage."""
import sys
import time

sys.path.insert(0, ".")
import numpy as np

from hypotremormcmc_amd import driver, synth
from hypotremormcmc_amd.obs_data import ObsData
from hypotremormcmc_amd.parallel import LocalWorld
from tests.helpers import load_case


def build(params, data, n_procs):
    obs = ObsData.from_arrays(data.sta_x, data.sta_y, data.t_obs, data.t_stdv, data.a_obs, data.a_stdv)
    fwd, sets = None, []
    for r in range(n_procs):
        fwd, cs = driver.build_rank(params, data.sta_x, data.sta_y, data.sta_z, obs, r, n_procs=n_procs, fwd=fwd)
        sets.append(cs)
    return sets


# 1. persistent kernel, bench workload, 1.5 M iterations (ring wraps ~60 times)
data = synth.make_synthetic(1000, 64, 1)
params = dict(synth.DEFAULT_PARAMS, n_procs=1, n_chains=8, n_cool=1, n_iter=10**8, n_burn=10**9, n_interval=1000)
cs = build(params, data, 1)[0]
t0 = time.time()
for k in range(15):
    cs.run(100000)
    print("persistent: %d iterations, %.1f s, %.0f steps/s" % (cs.iterations_done, time.time() - t0,
                                                                8 * cs.iterations_done / (time.time() - t0)), flush=True)
# 2. lock-step, 2 ranks x 2 chains (c1), 400 k iterations against the oracle's final RNG state
from oracle import oracle
fx, d1, p1 = load_case("c1")
n = 400000
p1 = dict(p1, n_iter=str(n), n_burn=str(n), n_interval="1000")
job = oracle.Job(p1, d1); t0 = time.time(); job.run(n); print("oracle c1 %d it: %.1f s" % (n, time.time() - t0), flush=True)
sets = build(p1, d1, 2)
w = LocalWorld(sets); t0 = time.time()
for k in range(8):
    w.run(n // 8)
    print("lock-step 2 ranks: %d iterations, %.1f s" % (sets[0].iterations_done, time.time() - t0), flush=True)
ok = all(sets[r].rng_state() == job.rng_state(r) for r in range(2))
it0, lk0 = job.likelihood_trace(0); gi, _, gl = sets[0].likelihood_trace()
ok = ok and np.array_equal(gi, it0) and np.allclose(gl, lk0, rtol=1e-9, atol=0)
print("lock-step end state equals the oracle:", ok, flush=True)
assert ok
