"""Run ON THE GPU BOX: rejection-heavy runs of the OTHER code paths against the oracle -- two ranks in lock-step, chains in
rounds (more chains than waves), two stations per lane, the generic station path, runs cut in pieces with tiny record
buffers, a small random-stream ring, checkpoint/restore in the middle.  python tools/stress_variants.py [n_iter]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
from hypotremormcmc_amd import driver, synth
from hypotremormcmc_amd.obs_data import ObsData
from hypotremormcmc_amd.parallel import LocalWorld
from oracle import oracle


def build(data, params, **caps):
    obs = ObsData.from_arrays(data.sta_x, data.sta_y, data.t_obs, data.t_stdv, data.a_obs, data.a_stdv)
    n_procs = int(params["n_procs"])
    fwd, sets = None, []
    for r in range(n_procs):
        fwd, cs = driver.build_rank(params, data.sta_x, data.sta_y, data.sta_z, obs, r, n_procs=n_procs, fwd=fwd, **caps)
        sets.append(cs)
    return sets


def compare(job, sets):
    ok = True
    npr = np.zeros(7, np.int64); nac = np.zeros(7, np.int64)
    for r, cs in enumerate(sets):
        it, lk = job.likelihood_trace(r)
        gi, _, gl = cs.likelihood_trace()
        a, b = cs.counts()
        npr += a; nac += b
        ok = ok and np.array_equal(gi, it) and len(gl) == len(lk) and np.allclose(gl, lk, rtol=1e-9, atol=0)
        ok = ok and cs.rng_state() == job.rng_state(r)
    oa, ob = job.counts()
    return ok and np.array_equal(npr, oa) and np.array_equal(nac, ob)


bad = 0
cases = [
    ("two ranks, lock-step", dict(E=200, S=64, nc=4, n_procs=2, sz=8.0), "lockstep"),
    ("three ranks, lock-step", dict(E=64, S=64, nc=3, n_procs=3, sz=12.0), "lockstep"),
    ("19 chains in rounds", dict(E=200, S=64, nc=19, n_procs=1, sz=8.0), "run"),
    ("two stations per lane", dict(E=300, S=128, nc=8, n_procs=1, sz=8.0), "run"),
    ("generic station path", dict(E=100, S=300, nc=6, n_procs=1, sz=8.0), "run"),
    ("pieces + tiny record buffers", dict(E=64, S=64, nc=8, n_procs=1, sz=12.0), "pieces"),
    ("checkpoint / restore", dict(E=64, S=64, nc=8, n_procs=1, sz=12.0), "ckpt"),
    ("single rank driven in lock-step", dict(E=64, S=64, nc=8, n_procs=1, sz=12.0), "lockstep"),
    ("small random-stream ring", dict(E=64, S=64, nc=8, n_procs=1, sz=12.0), "smallring"),
    ("small ring, two ranks in lock-step", dict(E=64, S=64, nc=4, n_procs=2, sz=8.0), "smallring-lockstep"),
]
for name, c, how in cases:
    data = synth.make_synthetic(c["E"], c["S"], 300 + len(name))
    params = dict(synth.DEFAULT_PARAMS, n_procs=c["n_procs"], n_chains=c["nc"], n_cool=2, n_iter=n_iter, n_burn=n_iter // 2,
                  n_interval=3, step_size_z=c["sz"], step_size_vs=0.4)
    t0 = time.time()
    try:
        job = oracle.Job(params, data); job.run(n_iter)
        if how == "pieces":
            sets = build(data, params, lik_capacity=16, sample_capacity=16)
            done = 0
            for k in (1, 7, 500, 3, n_iter):
                k = min(k, n_iter - done)
                if k > 0:
                    sets[0].run(k); done += k
        elif how == "ckpt":
            sets = build(data, params)
            sets[0].run(n_iter // 3)
            blob = sets[0].checkpoint()
            lik0 = [x.copy() for x in sets[0].likelihood_trace()]
            sets2 = build(data, params)
            sets2[0].restore(blob)
            sets2[0].run(n_iter - n_iter // 3)
            # the restored set continues the run: its records follow those of the first third
            it, lk = job.likelihood_trace(0)
            gi = np.concatenate([lik0[0], sets2[0].likelihood_trace()[0]]); gl = np.concatenate([lik0[2], sets2[0].likelihood_trace()[2]])
            ok = np.array_equal(gi, it) and np.allclose(gl, lk, rtol=1e-9, atol=0) and sets2[0].rng_state() == job.rng_state(0)
            bad += 0 if ok else 1
            print("%-34s %s  (%.1f s)" % (name, "ok" if ok else "MISMATCH", time.time() - t0), flush=True)
            continue
        elif how == "lockstep":
            sets = build(data, params)
            LocalWorld(sets).run(n_iter)
        elif how.startswith("smallring"):
            os.environ["HTM_STREAM_CAP"] = "131072"      # the ring wraps every ~2 600 iterations; launches end at its edge
            try:
                sets = build(data, params)
            finally:
                del os.environ["HTM_STREAM_CAP"]
            if how.endswith("lockstep"):
                LocalWorld(sets).run(n_iter)
            else:
                sets[0].run(n_iter)
        else:
            sets = build(data, params)
            sets[0].run(n_iter)
        ok = compare(job, sets)
        msg = "ok" if ok else "MISMATCH"
    except Exception as e:      # noqa: BLE001
        ok, msg = False, "ERROR %r" % (e,)
    bad += 0 if ok else 1
    print("%-34s %s  (%.1f s)" % (name, msg, time.time() - t0), flush=True)
print("FAILED %d" % bad if bad else "all equal")
sys.exit(1 if bad else 0)
