"""Run ON THE GPU BOX: first chain step that differs from the oracle, with its neighbourhood.
python tools/diag_steps.py E S chains seed_offset step_size_z n_iter"""
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from hypotremormcmc_amd import driver, synth
from hypotremormcmc_amd.obs_data import ObsData
from oracle import oracle

E, S, nc, seed = (int(x) for x in sys.argv[1:5])
sz = float(sys.argv[5]); n_iter = int(sys.argv[6])
data = synth.make_synthetic(E, S, 100 + seed)
params = dict(synth.DEFAULT_PARAMS, n_procs=1, n_chains=nc, n_cool=2 if nc > 2 else 1, n_iter=n_iter, n_burn=n_iter // 2,
              n_interval=3, step_size_z=sz, step_size_vs=0.4)
job = oracle.Job(params, data); job.enable_steplog(n_iter * nc); job.run(n_iter)
obs = ObsData.from_arrays(data.sta_x, data.sta_y, data.t_obs, data.t_stdv, data.a_obs, data.a_stdv)
fwd, cs = driver.build_rank(params, data.sta_x, data.sta_y, data.sta_z, obs, 0, n_procs=1)
cs.enable_steplog(n_iter * nc)
cs.run(n_iter)
oi, od = job.steplog(); gi, gd = cs.steplog()
n = min(len(gi), len(oi))
print("rows", len(gi), len(oi))
same_i = np.all(gi[:n, 2:7] == oi[:n, 3:8], axis=1) & (gi[:n, 0] == oi[:n, 0]) & (gi[:n, 1] == oi[:n, 2])
ok = oi[:n, 5] == 1
same_d = np.isclose(gd[:n, 0], od[:n, 0], rtol=1e-12, atol=1e-13) & (~ok | np.isclose(gd[:n, 1], od[:n, 1], rtol=1e-9)) & np.isclose(gd[:n, 2], od[:n, 2], rtol=1e-9)
bad = np.nonzero(~(same_i & same_d))[0]
if len(bad) == 0:
    print("all steps equal"); sys.exit(0)
k = bad[0]
it, ch = gi[k, 0], gi[k, 1]
print("first differing row", k, "iteration", it, "chain", ch)
print("columns: iter chain type idx prior_ok accepted need_full | x_new L_new L_post T   (gpu / oracle)")
for r in range(n):
    if gi[r, 1] == ch and it - 3 <= gi[r, 0] <= it + 1:
        print("gpu   ", gi[r, :7], ["%.12g" % v for v in gd[r]])
        print("oracle", oi[r, [0, 2, 3, 4, 5, 6, 7]], ["%.12g" % v for v in od[r]])
print("other chains in iterations", it - 1, it)
for r in range(n):
    if gi[r, 1] != ch and it - 1 <= gi[r, 0] <= it:
        print("gpu   ", gi[r, :7], "oracle", oi[r, [0, 2, 3, 4, 5, 6, 7]])
