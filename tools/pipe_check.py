"""Development check of another chain master on the GPU box: the same job on the pipelined master (csrc/htm_pipe.hpp; or, with
CHECK_MB=1, on several master workgroups: HTM_MB=1, csrc/htm_flow.hpp MbShared) and on the free-running one (HTM_PIPE=0 / HTM_MB=0) -- every step's type / element / prior_ok / accept, values to rounding, RNG position, counters --
then a timing of the pipelined master.      python tools/pipe_check.py [n_chains] [n_events] [n_sta] [n_iter] [time_iters]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from hypotremormcmc_amd import driver, synth
from hypotremormcmc_amd.obs_data import ObsData

nc = int(sys.argv[1]) if len(sys.argv) > 1 else 8
E_ = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
S_ = int(sys.argv[3]) if len(sys.argv) > 3 else 64
n_iter = int(sys.argv[4]) if len(sys.argv) > 4 else 3000
n_time = int(sys.argv[5]) if len(sys.argv) > 5 else 200000
prec = os.environ.get("PIPE_CHECK_PREC", "fp64")
seed = int(os.environ.get("PIPE_CHECK_SEED", "1"))
data = synth.make_synthetic(E_, S_, seed)
params = dict(synth.DEFAULT_PARAMS, n_procs=1, n_chains=nc, n_cool=int(os.environ.get("PIPE_CHECK_COOL", "1")), n_iter=10**7, n_burn=100, n_interval=50, forward_precision=prec)
if os.environ.get("PIPE_CHECK_INT"):
    params.update(n_interval=int(os.environ["PIPE_CHECK_INT"]), n_burn=int(os.environ.get("PIPE_CHECK_BURN", "100")))
if os.environ.get("PIPE_CHECK_SZ"):
    params.update(step_size_z=float(os.environ["PIPE_CHECK_SZ"]), step_size_vs=0.4)
obs = ObsData.from_arrays(data.sta_x, data.sta_y, data.t_obs, data.t_stdv, data.a_obs, data.a_stdv)


MB = os.environ.get("CHECK_MB") == "1"


def run(pipe):
    if MB:
        os.environ["HTM_MB"] = "1" if pipe else "0"
    else:
        os.environ["HTM_PIPE"] = "1" if pipe else "0"
    fwd, cs = driver.build_rank(params, data.sta_x, data.sta_y, data.sta_z, obs, 0, n_procs=1)
    if not os.environ.get("PIPE_CHECK_NOSLOG"):
        cs.enable_steplog(n_iter * nc)
    cs.run(n_iter)
    return fwd, cs


fa, a = run(True)
fb, b = run(False)
print("loops:", a.master_stats(), b.master_stats())
ia, da = a.steplog(); ib, db = b.steplog()
print("rows", len(ia), len(ib))
n = min(len(ia), len(ib))
bad = np.nonzero(np.any(ia[:n] != ib[:n], axis=1))[0]
if n == 0:
    print("(no step log)")
elif len(bad):
    k = bad[0]
    print("FIRST DIFFERENT STEP row %d:\n pipe %s %s\n flow %s %s" % (k, ia[k], da[k], ib[k], db[k]))
    for j in range(max(0, k - 2 * nc), k):
        print("   ", ia[j], da[j][:3], "|", ib[j], db[j][:3])
else:
    print("all %d steps: same type / element / prior_ok / accept / full" % n)
    okm = ia[:n, 4] == 1
    print("max rel diff  x_new %.3e  L_new %.3e  L_post %.3e  T equal %s" % (
        np.max(np.abs(da[:n, 0] - db[:n, 0]) / np.maximum(1e-300, np.abs(db[:n, 0]))),
        np.max(np.abs(da[:n][okm, 1] - db[:n][okm, 1]) / np.abs(db[:n][okm, 1])),
        np.max(np.abs(da[:n, 2] - db[:n, 2]) / np.abs(db[:n, 2])), np.array_equal(da[:n, 3], db[:n, 3])))
print("rng equal:", a.rng_state() == b.rng_state())
ca, cb = a.counts(), b.counts()
print("counters equal:", np.array_equal(ca[0], cb[0]) and np.array_equal(ca[1], cb[1]))
la, lb = a.likelihood_trace(), b.likelihood_trace()
print("lik records:", len(la[0]), len(lb[0]), "iters equal", np.array_equal(la[0], lb[0]), "chains equal", np.array_equal(la[1], lb[1]),
      "max rel", float(np.max(np.abs(la[2] - lb[2]) / np.abs(lb[2]))) if len(la[2]) == len(lb[2]) and len(la[2]) else None)
if len(la[2]) == len(lb[2]) and len(la[2]):
    rel = np.abs(la[2] - lb[2]) / np.abs(lb[2])
    badr = np.nonzero(rel > 1e-9)[0]
    if len(badr):
        print("first differing lik record %d: iteration %d chain %d  pipe %.12g flow %.12g  (%d differ)" % (badr[0], la[0][badr[0]], la[1][badr[0]], la[2][badr[0]], lb[2][badr[0]], len(badr)))
sa, sb = a.samples(), b.samples()
print("samples:", len(sa["iter"]), len(sb["iter"]), "iter/chain equal", np.array_equal(sa["iter"], sb["iter"]) and np.array_equal(sa["chain"], sb["chain"]),
      "values equal", all(np.array_equal(sa[k], sb[k]) for k in ("vs", "qs", "hypo", "t_corr", "a_corr")) if len(sa["iter"]) == len(sb["iter"]) else None)
for label, cs in (("pipe", a), ("flow", b)):
    cs.run(2000)
    t0 = time.perf_counter()
    cs.run(n_time)
    dt = time.perf_counter() - t0
    print("%s: %.3f us/iteration, %.3f M steps/s   %s" % (label, 1e6 * dt / n_time, nc * n_time / dt / 1e6, cs.master_stats()))
