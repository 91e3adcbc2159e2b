import ctypes as C, os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from hypotremormcmc_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "hypotremormcmc_amd", "lib", "libhtm_hip_stamps.so")
from tests.helpers import load_case
from tests.test_gpu_chains import _build_world
fx, data, params = load_case("c2")
fwd, sets = _build_world(data, params)
cs = sets[0]; cs.run(6)
lib = _lib.load()
lib.htm_chains_read_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
a = (C.c_double * 64)(); lib.htm_chains_read_stamps(cs.handle, a)
names = "px0 px1 py0 py1 pz0 pz1 out0 out1 x_old x_new cmp ev beta q idx type".split()
for n, v in zip(names, list(a)[16:32]): print(n, repr(v))
