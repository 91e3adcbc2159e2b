"""Diagnostic: a time line of a few iterations of the pipelined master (csrc/htm_pipe.hpp) from a -DHTM_STAMPS build.
    python tools/pipe_trace.py [n_chains] [n_events] [n_sta]"""
import ctypes as C
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from hypotremormcmc_amd import _lib

os.environ.setdefault("HTM_PIPE", "1")

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
cs.run(1000)
cs.run(6000)          # (the trace covers iterations 3000..3039 of a launch that does not start at iteration 0)
lib = C.CDLL(_lib.LIB_PATH) if False else _lib.load()
lib.htm_chains_read_trace.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int]
n = 8192
a = (C.c_uint64 * (2 * n))()
assert lib.htm_chains_read_trace(cs.handle, a, n) == 0
ev = [(a[2 * k], a[2 * k + 1] >> 48, (a[2 * k + 1] >> 8) & 0xffffffffff, a[2 * k + 1] & 0xff) for k in range(n) if a[2 * k]]
ev.sort()
names = {1: "F publish", 2: "F order", 3: "E start", 4: "E done", 5: "C answer", 6: "D top", 7: "D has all", 8: "D done"}
t0 = ev[0][0] if ev else 0
for t, code, it, c in ev:
    print("%8d  %-10s it %d chain %d" % (t - t0, names.get(code, code), it, c))
