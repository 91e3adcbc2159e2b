"""Debug aid: first chain step at which the HIP path and the oracle disagree (run on the GPU box)."""
import os, sys
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from tests.helpers import load_case
from tests.test_gpu_chains import _build_world
from oracle import oracle

name = sys.argv[1] if len(sys.argv) > 1 else "c2"
n_iter = int(sys.argv[2]) if len(sys.argv) > 2 else 300
fx, data, params = load_case(name)
nc = int(params["n_chains"])
job = oracle.Job(params, data); job.enable_steplog(n_iter * nc); job.run(n_iter)
fwd, sets = _build_world(data, params)
cs = sets[0]; cs.enable_steplog(n_iter * nc); cs.run(n_iter)
oi, od = job.steplog(); gi, gd = cs.steplog()
print("rows", len(oi), len(gi), "rng", cs.rng_state() == job.rng_state(0))
n = min(len(oi), len(gi))
for k in range(n):
    a = (gi[k, 0], gi[k, 1], *gi[k, 2:7]); b = (oi[k, 0], oi[k, 2], *oi[k, 3:8])
    bad = a != b or abs(gd[k, 0] - od[k, 0]) > 1e-9 * max(1, abs(od[k, 0])) or abs(gd[k, 2] - od[k, 2]) > 1e-9 * abs(od[k, 2]) or gd[k,3] != od[k,3]
    if bad:
        for j in range(max(0, k - 2), min(n, k + 3)):
            print("gpu", gi[j].tolist(), gd[j].tolist())
            print("orc", oi[j].tolist(), od[j].tolist())
        break
else:
    print("all", n, "rows agree")
