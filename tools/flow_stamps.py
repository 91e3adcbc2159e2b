"""Diagnostic: where the waves of the free-running chain master (csrc/htm_flow.hpp) spend their cycles, from a -DHTM_STAMPS
build (never used for timing claims).  Build: make -C hypotremormcmc_amd/csrc stamps ; run on the GPU box:
    python tools/flow_stamps.py [n_chains] [n_events] [n_sta]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from hypotremormcmc_amd import _lib

_lib.LIB_PATH = os.environ.get("HTM_STAMPS_LIB") or os.path.join(ROOT, "hypotremormcmc_amd", "lib", "libhtm_hip_stamps.so")
from hypotremormcmc_amd import driver, synth
from hypotremormcmc_amd.obs_data import ObsData

nc = int(sys.argv[1]) if len(sys.argv) > 1 else 8
E_ = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
S_ = int(sys.argv[3]) if len(sys.argv) > 3 else 64
data = synth.make_synthetic(E_, S_, 1)
params = dict(synth.DEFAULT_PARAMS, n_procs=1, n_chains=nc, n_cool=1, n_iter=10**7, n_burn=10**9, n_interval=1000)
obs = ObsData.from_arrays(data.sta_x, data.sta_y, data.t_obs, data.t_stdv, data.a_obs, data.a_stdv)
fwd, cs = driver.build_rank(params, data.sta_x, data.sta_y, data.sta_z, obs, 0, n_procs=1)
if os.environ.get("HTM_STAMPS_LOCK") == "1":        # the lock-step ranks' loop (k_mcmc<.., 4>, MODE_LOCKRUN), one rank: its own inbox only
    cs.xchg_handle()
    cs.xchg_connect(None)
    _run = cs.run_lockstep_direct
else:
    _run = cs.run
_run(2000 if E_ <= 1000 else 300)
lib = _lib.load()
lib.htm_chains_read_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
a = (C.c_uint64 * 128)()
lib.htm_chains_read_stamps(cs.handle, a)
base = list(a)
n = 20000 if E_ <= 1000 else 2000
t0 = time.perf_counter()
_run(n)
print("wall us/iteration %.3f (stamps build)" % (1e6 * (time.perf_counter() - t0) / n))
lib.htm_chains_read_stamps(cs.handle, a)
d = [a[k] - base[k] for k in range(128)]
names = ["front: loads issued", "proposal, check published", "evaluation (2 positions)", "turn", "swap + decision + commit", "records + orders"]
print("ticks per step, by wave (partial-update steps); full-evaluation steps: total ticks per step (of which waiting for the workers)")
print("%-4s %8s " % ("wave", "steps") + " ".join("%9s" % s[:9] for s in names) + " %9s %9s | %7s %9s %9s | %9s %9s" % ("lock-step", "sum", "jobs", "ticks", "wait", "loop/iter", "between"))
for w in range(min(8, nc)):
    b = d[32 + 12 * w:32 + 12 * w + 12]
    ns, nj = max(1, b[7]), max(1, b[8])
    ph = [b[k] / ns for k in range(6)]
    steps = b[7] + b[8]
    iters = steps / max(1, len(range(w, nc, 8)))
    inside = sum(b[:6]) + b[6] + b[9]
    print("%-4d %8d " % (w, b[7]) + " ".join("%9.0f" % x for x in ph) + " %9.0f %9.0f | %7d %9.0f %9.0f | %9.0f %9.0f" %
          (b[9] / ns, sum(ph) + b[9] / ns, b[8], b[6] / nj, b[10] / nj, b[11] / max(1.0, iters), (b[11] - inside) / max(1, steps)))

# worker block 0, thread 0 (100 MHz ticks summed over its jobs): order seen -> its events evaluated -> block sum stored
jobs_all = sum(d[32 + 12 * w + 8] for w in range(min(8, nc)))
if jobs_all:
    print("worker block 0 per order: evaluation of its events %.2f us, block sum + store %.2f us  (%d orders)" %
          ((d[22] - d[21]) / jobs_all / 100.0, (d[23] - d[22]) / jobs_all / 100.0, jobs_all))
