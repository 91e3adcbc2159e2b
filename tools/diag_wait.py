import sys, ctypes as C
sys.path.insert(0, '.')
import numpy as np
from hypotremormcmc_amd import synth, driver, _lib
from hypotremormcmc_amd.obs_data import ObsData
E,S,nc,ncool,seed,n_iter,over = (1000, 64, 8, 1, 17, 700, {"step_size_z": 8.0, "n_interval": 2})
data = synth.make_synthetic(E,S,seed)
params = dict(synth.DEFAULT_PARAMS, n_procs=1, n_chains=nc, n_cool=ncool, n_iter=n_iter, n_burn=n_iter//3, n_interval=5); params.update(over)
obs = ObsData.from_arrays(data.sta_x, data.sta_y, data.t_obs, data.t_stdv, data.a_obs, data.a_stdv)
fwd, cs = driver.build_rank(params, data.sta_x, data.sta_y, data.sta_z, obs, 0, n_procs=1)
try:
    cs.run(n_iter)
    print("ran fine")
except Exception as e:
    print("ERR", e)
lib = _lib.load()
lib.htm_chains_read_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
a = (C.c_uint64 * 128)()
lib.htm_chains_read_stamps(cs.handle, a)
print("master:", [hex(x) for x in a[100:111]])
print("worker:", [hex(x) for x in a[112:120]])
