"""Run ON THE GPU BOX with the -DHTM_STAMPS build: what a timed-out wait of the master was waiting for.
HTM_LIB=hypotremormcmc_amd/lib/libhtm_hip_stamps.so python tools/diag_wait.py E S chains seed step_size_z n_iter
(the run of tools/stress_rejections.py with the same numbers)"""
import ctypes as C
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from hypotremormcmc_amd import _lib, driver, synth
from hypotremormcmc_amd.obs_data import ObsData

E, S, nc, seed = (int(x) for x in sys.argv[1:5])
sz = float(sys.argv[5]); n_iter = int(sys.argv[6])
data = synth.make_synthetic(E, S, 100 + seed)
params = dict(synth.DEFAULT_PARAMS, n_procs=1, n_chains=nc, n_cool=2 if nc > 2 else 1, n_iter=n_iter, n_burn=n_iter // 2,
              n_interval=3, step_size_z=sz, step_size_vs=0.4)
obs = ObsData.from_arrays(data.sta_x, data.sta_y, data.t_obs, data.t_stdv, data.a_obs, data.a_stdv)
fwd, cs = driver.build_rank(params, data.sta_x, data.sta_y, data.sta_z, obs, 0, n_procs=1)
try:
    cs.run(n_iter)
    print("ran fine", cs.handoff_stats())
except Exception as e:      # noqa: BLE001
    print("ERR", e)
lib = _lib.load()
lib.htm_chains_read_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
a = (C.c_uint64 * 128)()
lib.htm_chains_read_stamps(cs.handle, a)
names = ["wait (1 sums, 2 seen value)", "chain", "tag", "pre", "pre_mode", "iter", "p", "type", "idx", "granule hi", "granule lo", "start[c]"]
print("master:", {n: hex(v) for n, v in zip(names, a[100:112])})
print("worker 0 (commit wait timed out):", [hex(x) for x in a[112:120]])
if a[100]:
    lib.htm_chains_read_handoff.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint64)]
    h = (C.c_uint64 * 64)()
    lib.htm_chains_read_handoff(cs.handle, int(a[101]), h)
    print("order slot of the chain (tag:payload):", ["%x:%x" % (v >> 32, v & 0xffffffff) for v in h[0:8]])
    print("tags of the first workers' sums:", ["%x" % (v >> 32) for v in h[8:64:2]])
