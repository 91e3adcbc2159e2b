import os, sys
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from tests.helpers import load_case
from tests.test_gpu_chains import _build_world
fx, data, params = load_case("c2")
params = dict(params, n_chains="1", n_interval="3")
_, sets = _build_world(data, params)
cs = sets[0]; cs.enable_steplog(64)
cs.run(3)
ir, dr = cs.steplog()
for r in ir: print(r.tolist())
