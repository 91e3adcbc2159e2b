"""Shared helpers for the parity tests (fixtures -> inputs)."""
import os

import numpy as np

from hypotremormcmc_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = ["c1", "c2", "missing", "timeonly", "fixedcorr", "rejects", "c3", "c4"]


def load_case(name):
    """Returns (fixture npz, SynthData inputs, params dict). Inputs come from the fixture when stored,
    else from the seeded generator (checksum-verified against what the reference was run on)."""
    import hashlib

    fx = np.load(os.path.join(GOLDEN, name + ".npz"))
    E, S = (int(v) for v in fx["in_shape"])
    data = synth.make_synthetic(E, S, int(fx["in_seed"]), int(fx["in_n_missing"]))
    if "in_t_obs" in fx:
        for key in ("sta_x", "sta_y", "sta_z", "t_obs", "t_stdv", "a_obs", "a_stdv"):
            assert np.array_equal(getattr(data, key), fx["in_" + key]), f"generator drift in {key}"
    h = hashlib.sha256()
    for a in (data.sta_x, data.sta_y, data.sta_z, data.t_obs, data.t_stdv, data.a_obs, data.a_stdv):
        h.update(np.ascontiguousarray(a, dtype="<f8").tobytes())
    assert h.hexdigest() == str(fx["in_checksum"]), "synthetic generator no longer reproduces the fixture inputs"
    params = dict(zip(fx["param_keys"].tolist(), fx["param_vals"].tolist()))
    return fx, data, params


def tf(v):
    return str(v).strip().upper().lstrip(".").startswith("T")


class OracleRank:
    """A rank of the job computed by the CPU oracle, in the shape TorchWorld drives."""

    def __init__(self, job, rank, n_procs):
        import torch

        self.job, self.rank, self.n_procs = job, rank, n_procs
        self._rec = np.zeros(job.record_words())
        self.record = torch.from_numpy(self._rec)   # shares memory

    def step_begin(self):
        self.job.rank_begin(self.rank, self._rec)

    def step_end(self, gathered):
        g = gathered.numpy()
        rc = self.job.rank_end(self.rank, np.ascontiguousarray(g))
        assert rc == 0, f"rank_end returned {rc}"

    def drain(self):
        pass

    def counts(self):
        npr = np.zeros(7, np.int64); nac = np.zeros(7, np.int64)
        for c in range(int(self.job.p.n_chains)):
            st = self.job.chain(self.rank, c)
            npr += st["n_propose"]; nac += st["n_accept"]
        return npr, nac
