"""Diagnostic: per-phase cycle shares of k_step from a -DHTM_STAMPS build (never used for timing claims).
Build:  make -C hypotremormcmc_amd/csrc stamps ;  run on the GPU box:  python tools/stamps.py [n_chains]"""
import ctypes as C
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from hypotremormcmc_amd import _lib

_lib.LIB_PATH = os.path.join(ROOT, "hypotremormcmc_amd", "lib", "libhtm_hip_stamps.so")
from hypotremormcmc_amd import driver, synth
from hypotremormcmc_amd.obs_data import ObsData

nc = int(sys.argv[1]) if len(sys.argv) > 1 else 8
E_ = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
S_ = int(sys.argv[4]) if len(sys.argv) > 4 else 64
data = synth.make_synthetic(E_, S_, 1)
params = dict(synth.DEFAULT_PARAMS, n_procs=1, n_chains=nc, n_cool=1, n_iter=10**7, n_burn=10**9, n_interval=1000)
obs = ObsData.from_arrays(data.sta_x, data.sta_y, data.t_obs, data.t_stdv, data.a_obs, data.a_stdv)
fwd, cs = driver.build_rank(params, data.sta_x, data.sta_y, data.sta_z, obs, 0, n_procs=1)
LOCK = len(sys.argv) > 2 and sys.argv[2] == "lockstep"
if LOCK:
    from hypotremormcmc_amd.parallel import LocalWorld
    world = LocalWorld([cs])
    run = world.run
else:
    run = cs.run
run(2000 if E_ <= 1000 else 300)
lib = _lib.load()
lib.htm_chains_read_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
a = (C.c_uint64 * 128)()
lib.htm_chains_read_stamps(cs.handle, a)
base = list(a)
n = 10000 if E_ <= 1000 else 2000
import time
t0 = time.perf_counter()
run(n)
print("wall us/iter", 1e6 * (time.perf_counter() - t0) / n)
lib.htm_chains_read_stamps(cs.handle, a)
names = ["prologue", "P0 (resume judge + window)", "passes (propose+partial+decide+commit)", "validate/records/swap roles",
         "post (swap apply, records)", "epilogue", "hand-over to workers (wait+judge)"]
tot = sum(a[k] - base[k] for k in range(len(names)))
for k, nm in enumerate(names):
    d = a[k] - base[k]
    print("%-34s %9.0f ticks/iter  %5.1f %%" % (nm, d / n, 100.0 * d / tot))
cn = ["decode+load issue", "proposal arith (load wait)", "event_misfit", "final sum+decision", "commit+LDS write-back"]
for k, nm in enumerate(cn):
    print("  chain_pass[chain 0] %-28s %8.0f ticks/iter" % (nm, (a[32 + k] - base[32 + k]) / n))
nj = a[32 + 9] - base[32 + 9]
if nj:
    for k, nm in ((5, "order recognised / sent"), (6, "own event (two-ahead)"), (7, "wait for all granules"), (8, "sum + decision")):
        print("  chain_pass[chain 0] with job: %-24s %8.0f ticks/job  (%d jobs)" % (nm, (a[32 + k] - base[32 + k]) / nj, nj))
for b, nm in ((10, "two-ahead orders"), (13, "one-ahead / own orders")):
    cnt = a[32 + b + 1] - base[32 + b + 1]
    if cnt:
        print("  chain_pass[chain 0] wait for granules, %-24s %8.0f ticks/job, %.2f poll rounds  (%d jobs)" %
              (nm, (a[32 + b] - base[32 + b]) / cnt, (a[32 + b + 2] - base[32 + b + 2]) / cnt, cnt))
for k, nm in enumerate(("role V (validate, plan, bookkeeping)", "role R (records)", "role W (swap)", "window extension")):
    print("  roles phase, barrier A -> end of %-38s %7.0f ticks/iter" % (nm, (a[120 + k] - base[120 + k]) / n))
for wv in range(8):
    for job in (0, 1):
        cnt = a[64 + wv + 8 * job] - base[64 + wv + 8 * job]
        if cnt:
            print("  wave %d %s: loop top -> barrier A  %7.0f ticks  (%d iterations)" %
                  (wv, "with job" if job else "partial ", (a[48 + wv + 8 * job] - base[48 + wv + 8 * job]) / cnt, cnt))
print("  full-evaluation steps per chain: sent two ahead / other:",
      [(a[88 + k] - base[88 + k], a[80 + k] - base[80 + k]) for k in range(8)])
jobs = a[26] - base[26]
if jobs:
    d = lambda k: (a[k] - base[k]) / jobs
    print("  hand-over timeline (100 MHz ticks = 10 ns, per job, %d jobs):" % jobs)
    print("    publish -> worker 0 has seen job+order   %7.1f ns" % (10 * (d(21) - d(20))))
    print("    ... -> worker 0 first model reduced      %7.1f ns" % (10 * (d(22) - d(21))))
    print("    ... -> worker 0 last model combined      %7.1f ns" % (10 * (d(23) - d(22))))
    print("    ... -> master has all partials + judged  %7.1f ns" % (10 * (d(25) - d(23))))
    print("    orders sent ahead by role P: %d of %d" % (a[28] - base[28], jobs))
    print("    polls of worker 0: %d  (one per %.0f ns of device time)" % (a[27] - base[27], 1e3 * 0 + 0))
    print("    publish -> master done                   %7.1f ns" % (10 * (d(25) - d(20))))
st = cs.last_run_stats()
print("total ticks/iter", tot / n, " device_us/iter", st["device_us"] / n, st)
