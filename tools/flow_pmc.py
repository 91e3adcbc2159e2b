"""Run ON THE GPU BOX under rocprofv3 --pmc ...: the free-running master alone -- hypocentre proposals only (solve_* = F: no
full evaluations after the first iteration) and ONE worker block (HTM_MAX_WORKERS=1: one polling wave), so that the
kernel's instruction counters are the master's eight chain waves'.  20 000 iterations x 8 chains = 160 000 partial steps
in the counted launches (plus set-up).
    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM --kernel-trace -d gpurun_out/pmc1 -o p --output-format csv -- python3 tools/flow_pmc.py"""
import os
import sys

os.environ["HTM_MAX_WORKERS"] = "1"
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from hypotremormcmc_amd import driver, synth
from hypotremormcmc_amd.obs_data import ObsData

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
data = synth.make_synthetic(1000, 64, 1)
params = dict(synth.DEFAULT_PARAMS, n_procs=1, n_chains=8, n_cool=1, n_iter=10**7, n_burn=10**9, n_interval=1000,
              solve_vs="F", solve_t_corr="F", solve_qs="F", solve_a_corr="F")
obs = ObsData.from_arrays(data.sta_x, data.sta_y, data.t_obs, data.t_stdv, data.a_obs, data.a_stdv)
fwd, cs = driver.build_rank(params, data.sta_x, data.sta_y, data.sta_z, obs, 0, n_procs=1)
cs.run(10)          # the first iteration: full evaluations (by the one worker block)
import time
t0 = time.perf_counter()
cs.run(n)
print("us/iteration %.3f" % (1e6 * (time.perf_counter() - t0) / n))
