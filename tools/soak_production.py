#!/usr/bin/env python3
"""Run ON THE GPU BOX, once per round (kept under profiles/): a run of the reference sample file's length --
sample/hypo_tremor.in: n_iter = 4 000 000, n_burn = 2 000 000, n_interval = 1 000, 5 chains per rank, n_cool = 1,
its priors and step sizes -- on ONE rank, against the CPU oracle: record counts and iterations, every recorded
log-likelihood within 1e-9 relative, proposal counters, final RNG state, final parameter vectors.  It crosses
hundreds of wrap-arounds of the random-stream ring, thousands of launches and 2 x 10^6 work-order tags.
The data set is 100 events x 16 stations (the oracle needs minutes, not hours, for 2 x 10^7 proposal steps).

    python tools/soak_production.py [n_iter] > gpurun_out/soak_production.txt
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from hypotremormcmc_amd import driver, synth  # noqa: E402
from hypotremormcmc_amd.obs_data import ObsData  # noqa: E402
from oracle import oracle  # noqa: E402

n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 4000000
E, S, nc = 100, 16, 5
data = synth.make_synthetic(E, S, 11)
params = dict(synth.DEFAULT_PARAMS, n_procs=1, n_chains=nc, n_cool=1, n_iter=n_iter, n_burn=n_iter // 2, n_interval=1000)
obs = ObsData.from_arrays(data.sta_x, data.sta_y, data.t_obs, data.t_stdv, data.a_obs, data.a_stdv)
t0 = time.time()
fwd, cs = driver.build_rank(params, data.sta_x, data.sta_y, data.sta_z, obs, 0, n_procs=1)
done = 0
while done < n_iter:                       # in slices, with a sign of life (gpurun takes 7 silent minutes for a hang)
    k = min(500000, n_iter - done)
    cs.run(k); done += k
    print(f"# gpu: {done} iterations, {time.time() - t0:.1f} s", flush=True)
t_gpu = time.time() - t0
st = cs.last_run_stats()
t0 = time.time()
job = oracle.Job(params, data)
done = 0
while done < n_iter:
    k = min(250000, n_iter - done)
    job.run(k); done += k
    print(f"# oracle: {done} iterations, {time.time() - t0:.1f} s", flush=True)
t_cpu = time.time() - t0
it, lk = job.likelihood_trace(0)
gi, gc, gl = cs.likelihood_trace()
a, b = cs.counts(); oa, ob = job.counts()
ok_rec = np.array_equal(gi, it)
rel = float(np.max(np.abs(gl - lk) / np.abs(lk))) if ok_rec and len(lk) else float("nan")
ok_rng = cs.rng_state() == job.rng_state(0)
ok_cnt = np.array_equal(a, oa) and np.array_equal(b, ob)
smp = cs.samples()
ok_smp = len(smp["iter"]) == len(job.samples(0)["iter"]) if hasattr(job, "samples") else True
worst_x = 0.0
for c in range(nc):
    g, o = cs.state(c), job.chain(0, c)
    for key in ("hypo", "t_corr", "a_corr"):
        worst_x = max(worst_x, float(np.max(np.abs(getattr(g, key) - o[key]))))
    worst_x = max(worst_x, abs(g.vs - o["vs"]), abs(g.qs - o["qs"]), abs(g.temp - o["temp"]))
print(f"{E} x {S}, {nc} chains, {n_iter} iterations (n_burn {n_iter // 2}, n_interval 1000): gpu {t_gpu:.1f} s, oracle {t_cpu:.1f} s")
print(f"likelihood records: {len(gi)} (oracle {len(it)}), same iterations: {ok_rec}, max relative difference {rel:.3e}")
print(f"sample records: {len(smp['iter'])}; proposal counters equal: {ok_cnt}; final RNG state equal: {ok_rng}; "
      f"final state max |difference| {worst_x:.3e}")
print(f"proposed {a.tolist()} accepted {b.tolist()}")
good = ok_rec and rel <= 1e-9 and ok_rng and ok_cnt and worst_x <= 1e-9
print("RESULT:", "equal" if good else "MISMATCH")
sys.exit(0 if good else 1)
