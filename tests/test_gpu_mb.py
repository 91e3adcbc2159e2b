"""GPU parity of SEVERAL MASTER WORKGROUPS in one launch (csrc/htm_flow.hpp MbShared, k_mcmc<.., 7>; 9..16 chains on a rank: two
workgroups of eight chain waves; the default there, HTM_MB=0: one workgroup): what the chains of a rank share -- checks, epoch and anchor, the swap, the
end of the launch -- goes through memory instead of LDS, so the runs that matter are the rejection-heavy ones (every Rayleigh
rejection, src/cls_model.f90:178-181, is an epoch change every chain of the other workgroup has to learn from memory) and the
ones cut into many launches.  Criteria as in tests/test_gpu_chains.py: the oracle step by step, bit-equality with the single
workgroup."""
import numpy as np
import pytest

from tests.test_gpu_chains import RTOL_TRACE, _build_world

import os

# (The one stress configuration that used to differ from the oracle in one run of seven -- tools/mb_repro.py: 100 x 64 x 16, depth
# steps of 20 -- is test_the_configuration_that_used_to_differ below; what it was: DESIGN.md 3.0, profiles/r04_y_mb_steplog_*.txt.)
pytestmark = [pytest.mark.gpu]


def _job(E, S, nc, seed, sz, n_iter, **kw):
    from hypotremormcmc_amd import synth

    data = synth.make_synthetic(E, S, 100 + seed)
    params = dict(synth.DEFAULT_PARAMS, n_procs=1, n_chains=nc, n_cool=2, n_iter=n_iter, n_burn=n_iter // 2, n_interval=3,
                  step_size_z=sz, step_size_vs=0.4)
    params.update(kw)
    return data, params


@pytest.mark.parametrize("E,S,nc,seed,sz,n_iter", [(64, 64, 16, 1, 12.0, 3000), (64, 32, 11, 3, 20.0, 4000), (300, 128, 13, 2, 6.0, 2000),
                                                  (1000, 64, 16, 4, 0.4, 1500)])
def test_two_master_workgroups_against_oracle(E, S, nc, seed, sz, n_iter, monkeypatch):
    from oracle import oracle

    monkeypatch.setenv("HTM_MB", "1")
    data, params = _job(E, S, nc, seed, sz, n_iter)
    job = oracle.Job(params, data); job.run(n_iter)
    _, sets = _build_world(data, params)
    assert sets[0].master_stats()["single_rank_loop"] == 7, "two master workgroups were not selected"
    sets[0].run(n_iter)
    it, lk = job.likelihood_trace(0)
    gi, _, gl = sets[0].likelihood_trace()
    assert len(gi) == len(it) > n_iter // 2 and np.array_equal(gi, it)
    np.testing.assert_allclose(gl, lk, rtol=RTOL_TRACE)
    assert sets[0].rng_state() == job.rng_state(0)
    a, b = sets[0].counts(); oa, ob = job.counts()
    assert np.array_equal(a, oa) and np.array_equal(b, ob)


@pytest.mark.parametrize("E,S,nc,sz", [(64, 64, 16, 12.0), (1000, 64, 16, 0.4), (300, 64, 9, 6.0)])
def test_two_master_workgroups_give_the_bits_of_one(E, S, nc, sz, monkeypatch):
    """every step's proposal, decision, proposed and running log-likelihood and temperature: bit for bit what ONE workgroup
    computes (same arithmetic, same association: only the medium of the chains' hand-shakes differs) -- also when the run is cut
    into launches of odd lengths with tiny record buffers (the launch's end goes through memory too)"""
    data, params = _job(E, S, nc, 7, sz, 1200)
    logs = []
    for mb, cut in (("0", None), ("1", None), ("1", (1, 7, 400, 13))):
        monkeypatch.setenv("HTM_MB", mb)
        _, sets = _build_world(data, params, **({} if cut is None else dict(lik_capacity=5 * nc, sample_capacity=5 * nc)))
        cs = sets[0]
        assert cs.master_stats()["single_rank_loop"] == (7 if mb == "1" else 3)
        cs.enable_steplog(1200 * nc)
        done = 0
        for n in (cut or ()):
            cs.run(n); done += n
        cs.run(1200 - done)
        gi, gd = cs.steplog()
        logs.append((gi.copy(), gd.copy(), cs.likelihood_trace(), cs.rng_state(), cs.counts(), cs.samples()["hypo"].copy()))
    for k in (1, 2):
        assert np.array_equal(logs[0][0], logs[k][0]) and np.array_equal(logs[0][1], logs[k][1])
        assert all(np.array_equal(a, b) for a, b in zip(logs[0][2], logs[k][2]))
        assert logs[0][3] == logs[k][3]
        assert all(np.array_equal(a, b) for a, b in zip(logs[0][4], logs[k][4]))
        assert np.array_equal(logs[0][5], logs[k][5])


def test_two_master_workgroups_checkpoint_and_fp32(monkeypatch):
    """resume from a checkpoint == the uninterrupted run (bits), and the fp32-forward instantiation against its one-workgroup run"""
    monkeypatch.setenv("HTM_MB", "1")
    data, params = _job(1000, 128, 16, 5, 2.0, 600)
    _, a = _build_world(data, params); a[0].run(600)
    _, b = _build_world(data, params); b[0].run(250)
    blob = b[0].checkpoint()
    _, c = _build_world(data, params); c[0].restore(blob); c[0].run(350)
    ia, ca, la = a[0].likelihood_trace(); ic, cc, lc = c[0].likelihood_trace()
    keep = ia > 250
    assert np.array_equal(ia[keep], ic) and np.array_equal(ca[keep], cc) and np.array_equal(la[keep], lc)
    assert a[0].rng_state() == c[0].rng_state()
    out = []
    for mb in ("0", "1"):
        monkeypatch.setenv("HTM_MB", mb)
        _, s = _build_world(data, dict(params, forward_precision="fp32")); s[0].run(600)
        out.append((s[0].likelihood_trace(), s[0].rng_state()))
    assert all(np.array_equal(x, y) for x, y in zip(out[0][0], out[1][0])) and out[0][1] == out[1][1]


def test_the_configuration_that_used_to_differ(monkeypatch):
    """100 x 64 x 16, depth steps of 20 (tools/mb_repro.py): a chain late from a full evaluation adopts an anchor up to three
    iterations behind the wave that keeps its workgroup's stream window -- the window must still hold it (csrc/htm_hip.hip:
    mb_need).  With a 512-position ring one run in seven took ONE step of chain 2 from a wrong stream position (iterations
    5171, 5636, 6596, 7283, 8537, ...); every step of several runs against the oracle's step log."""
    from oracle import oracle

    monkeypatch.setenv("HTM_MB", "1")
    n_iter = 9000
    data, params = _job(100, 64, 16, 4, 20.0, n_iter)
    job = oracle.Job(params, data); job.enable_steplog(n_iter * 16); job.run(n_iter)
    oi, od = job.steplog()
    for run in range(6):
        _, sets = _build_world(data, params)
        cs = sets[0]
        assert cs.master_stats()["single_rank_loop"] == 7, "two master workgroups were not selected"
        cs.enable_steplog(n_iter * 16)
        cs.run(n_iter)
        gi, gd = cs.steplog()
        assert len(gi) == len(oi)
        assert np.array_equal(gi[:, 0], oi[:, 0]) and np.array_equal(gi[:, 1], oi[:, 2]), run
        assert np.array_equal(gi[:, 2:7], oi[:, 3:8]), "run %d: a step differs (type / element / prior check / decision)" % run
        np.testing.assert_allclose(gd[:, 2], od[:, 2], rtol=RTOL_TRACE)
        assert cs.rng_state() == job.rng_state(0)
        del cs, sets
