"""Diagnostic: where the three roles of the pipelined master (csrc/htm_pipe.hpp) spend their cycles, from a -DHTM_STAMPS build
(never used for timing claims).  Build: make -C hypotremormcmc_amd/csrc stamps ; on the GPU box:
    python tools/pipe_stamps.py [n_chains] [n_events] [n_sta]"""
import ctypes as C
import os
import sys
import time

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
cs.run(2000)
lib = _lib.load()
lib.htm_chains_read_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
a = (C.c_uint64 * 128)()
lib.htm_chains_read_stamps(cs.handle, a)
base = list(a)
n = 50000 if E_ <= 1000 else 5000
t0 = time.perf_counter()
cs.run(n)
dt = time.perf_counter() - t0
print("wall us/iteration %.3f (stamps build)   %s" % (1e6 * dt / n, cs.master_stats()))
lib.htm_chains_read_stamps(cs.handle, a)
d = [a[k] - base[k] for k in range(128)]
f, dd, e, cc = d[32:40], d[40:48], d[48:56], d[56:64]
it = max(1, dd[4])
print("shader cycles (s_memtime) per iteration unless stated")
print("F  produced %d (%.2f of them with inputs not requested ahead): looking ahead %.1f  records %.1f  window %.1f  orders %.1f  idle %.1f" %
      (f[2], f[4] / max(1, f[2]), f[5] / it, f[6] / it, f[7] / it, f[3] / it, f[1] / it))
print("D  iterations %d  flushes %d: waiting for the records %.1f  for the evaluations %.1f  deciding %.1f  top %.1f" %
      (dd[4], dd[5], dd[0] / it, dd[1] / it, dd[2] / it, dd[3] / it))
print("E  (five waves, summed) single-event tasks %d (%d evaluated twice): waiting for records %.1f  evaluation %.1f per task  loop top %.1f" %
      (e[4], e[6], e[0] / it, e[1] / max(1, e[4]), e[3] / it))
print("C  answers taken %d: looking %.1f  polling %.1f  idle %.1f" % (cc[5], cc[0] / it, cc[2] / it, cc[3] / it))
