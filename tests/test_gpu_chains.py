"""GPU parity: device-resident chains (propose -> forward -> judge -> swap on the MI355X) against the
reference's own output (golden fixtures) and against the oracle.

Criterion (north_star / SURVEY §8d): log-likelihood traces within 1e-9 relative per record, identical
accept/reject decisions (=> identical proposal_count.txt), identical RNG consumption (=> identical final
RNG state)."""
import os

import numpy as np
import pytest

from tests.helpers import load_case, tf

pytestmark = pytest.mark.gpu

RTOL_TRACE = 1e-9


def _obs(data):
    from hypotremormcmc_amd.obs_data import ObsData

    return ObsData.from_arrays(data.sta_x, data.sta_y, data.t_obs, data.t_stdv, data.a_obs, data.a_stdv)


def _build_world(data, params, **caps):
    from hypotremormcmc_amd import driver

    obs = _obs(data)
    n_procs = int(params["n_procs"])
    fwd, sets = None, []
    for r in range(n_procs):
        fwd, cs = driver.build_rank(params, data.sta_x, data.sta_y, data.sta_z, obs, r, n_procs=n_procs, fwd=fwd, **caps)
        sets.append(cs)
    return fwd, sets


def _check_against_fixture(fx, params, sets):
    n_procs = int(params["n_procs"])
    npr = np.zeros(7, np.int64); nac = np.zeros(7, np.int64)
    for r in range(n_procs):
        it, ch, lk = sets[r].likelihood_trace()
        assert np.array_equal(it, fx[f"lik_iter_{r}"]), f"rank {r}: recorded iterations differ"
        np.testing.assert_allclose(lk, fx[f"lik_{r}"], rtol=RTOL_TRACE, atol=0)
        smp = sets[r].samples()
        assert np.array_equal(smp["iter"], fx[f"vs_iter_{r}"])
        np.testing.assert_allclose(smp["vs"], fx[f"vs_{r}"][:, 0], rtol=1e-12)
        np.testing.assert_allclose(smp["qs"], fx[f"qs_{r}"][:, 0], rtol=1e-12)
        np.testing.assert_allclose(smp["t_corr"], fx[f"t_corr_{r}"], rtol=1e-11, atol=1e-13)
        np.testing.assert_allclose(smp["a_corr"], fx[f"a_corr_{r}"], rtol=1e-11, atol=1e-13)
        n_h = len(fx[f"hypo_{r}"])
        if n_h:
            np.testing.assert_allclose(smp["hypo"][-n_h:], fx[f"hypo_{r}"], rtol=1e-11, atol=1e-12)
        a, b = sets[r].counts()
        npr += a; nac += b
    assert np.array_equal(npr, fx["n_propose"])
    assert np.array_equal(nac, fx["n_accept"])


@pytest.mark.parametrize("name", ["c2", "missing", "c3"])
def test_single_rank_run_matches_reference_trace(name):
    fx, data, params = load_case(name)
    fwd, sets = _build_world(data, params)
    sets[0].run(int(params["n_iter"]))
    assert sets[0].iterations_done == int(params["n_iter"])
    _check_against_fixture(fx, params, sets)


@pytest.mark.parametrize("name", ["c1", "timeonly", "fixedcorr", "rejects", "c4"])
def test_multi_rank_lockstep_matches_reference_trace(name):
    """2-8 simulated ranks on one GPU; the record exchange is a device copy instead of the RCCL all-gather.
    c4 = BASELINE configs[3]: 1000 x 64, 8 ranks x 8 chains, temp_high 200, against the reference under mpiexec -np 8"""
    from hypotremormcmc_amd.parallel import LocalWorld

    fx, data, params = load_case(name)
    fwd, sets = _build_world(data, params)
    LocalWorld(sets).run(int(params["n_iter"]))
    _check_against_fixture(fx, params, sets)


def test_final_state_rng_and_steps_vs_oracle():
    """every step of every chain against the oracle: proposal type/index, prior_ok, accept, full/partial"""
    from oracle import oracle

    fx, data, params = load_case("c2")
    n_iter = 1500
    job = oracle.Job(params, data)
    job.enable_steplog(n_iter * 2)
    job.run(n_iter)
    fwd, sets = _build_world(data, params)
    cs = sets[0]
    cs.enable_steplog(n_iter * 2)
    cs.run(n_iter)
    assert cs.rng_state() == job.rng_state(0)              # same number of draws consumed
    oi, od = job.steplog()
    gi, gd = cs.steplog()
    assert len(gi) == len(oi) == n_iter * 2
    # oracle rows: iter, rank, chain, type, idx, prior_ok, accepted, used_full ; gpu rows: iter, chain, type, ...
    assert np.array_equal(gi[:, 0], oi[:, 0]) and np.array_equal(gi[:, 1], oi[:, 2])
    assert np.array_equal(gi[:, 2:7], oi[:, 3:8])
    ok = oi[:, 5] == 1
    np.testing.assert_allclose(gd[:, 0], od[:, 0], rtol=1e-12, atol=1e-13)          # x_new
    np.testing.assert_allclose(gd[ok, 1], od[ok, 1], rtol=RTOL_TRACE)              # proposed log-likelihood
    np.testing.assert_allclose(gd[:, 2], od[:, 2], rtol=RTOL_TRACE)                # current log-likelihood
    assert np.array_equal(gd[:, 3], od[:, 3])                                      # temperatures (swaps)
    for c in range(2):
        s, o = cs.state(c), job.chain(0, c)
        np.testing.assert_allclose(s.hypo, o["hypo"], rtol=1e-11, atol=1e-12)
        np.testing.assert_allclose(s.t_corr, o["t_corr"], rtol=1e-11, atol=1e-13)
        assert s.temp == o["temp"]
        assert np.array_equal(s.n_propose, o["n_propose"]) and np.array_equal(s.n_accept, o["n_accept"])


def test_lockstep_equals_single_rank_driver():
    """n_procs = 1: step_begin/step_end (the multi-rank code path, one launch per iteration) and htm_chains_run (one
    persistent launch, orders sent up to two iterations ahead and corrected for the step in between) take the same
    decisions and consume the same randoms; log-likelihoods agree to rounding (the correction adds and subtracts one
    event's misfit)"""
    from hypotremormcmc_amd.parallel import LocalWorld

    fx, data, params = load_case("c2")
    n_iter = 700
    _, a = _build_world(data, params)
    a[0].run(n_iter)
    _, b = _build_world(data, params)
    LocalWorld(b).run(n_iter)
    ia, ca, la = a[0].likelihood_trace(); ib, cb, lb = b[0].likelihood_trace()
    assert np.array_equal(ia, ib)
    np.testing.assert_allclose(la, lb, rtol=1e-13, atol=0)
    assert a[0].rng_state() == b[0].rng_state()
    for c in range(2):
        assert np.array_equal(a[0].state(c).hypo, b[0].state(c).hypo)
        assert np.array_equal(a[0].state(c).n_accept, b[0].state(c).n_accept)


def test_run_in_pieces_and_small_record_buffers():
    """run(n) == run(a) + run(b); tiny device record buffers force mid-run drains without changing results"""
    fx, data, params = load_case("missing")
    n_iter = int(params["n_iter"])
    _, a = _build_world(data, params)
    a[0].run(n_iter)
    _, b = _build_world(data, params, lik_capacity=8, sample_capacity=6)
    b[0].run(1000); b[0].run(1); b[0].run(n_iter - 1001)
    ia, _, la = a[0].likelihood_trace(); ib, _, lb = b[0].likelihood_trace()
    # same decisions and the same bits: a full evaluation's sum does not depend on where a launch ended or when its order went
    # out (the event of the chain's previous hypocentre step is always the chain wave's, csrc/htm_flow.hpp)
    assert np.array_equal(ia, ib)
    assert np.array_equal(la, lb)
    sa, sb = a[0].samples(), b[0].samples()
    assert np.array_equal(sa["iter"], sb["iter"]) and np.array_equal(sa["hypo"], sb["hypo"])
    _check_against_fixture(fx, params, b)


@pytest.mark.parametrize("E,S,nc,sz", [(64, 64, 8, 12.0), (1000, 64, 8, 0.4), (300, 64, 19, 6.0), (1000, 128, 8, 2.0)])
def test_free_running_master_twice_gives_the_same_bits(E, S, nc, sz):
    """Two runs of a job -- and a third cut into launches of odd lengths -- give bit-identical proposed and running
    log-likelihoods at every step: when a full evaluation's order went out (one or two steps ahead, late after an epoch change:
    a matter of timing) does not reach the sum (cls_forward.f90:277-300 sums events in order; here the event of the chain's previous
    hypocentre step is always added by the chain's own wave, the rest by the workers in a fixed tree)."""
    from hypotremormcmc_amd import synth

    data = synth.make_synthetic(E, S, 11)
    n_iter = 1200
    params = dict(synth.DEFAULT_PARAMS, n_procs=1, n_chains=nc, n_cool=2, n_iter=n_iter, n_burn=n_iter // 2, n_interval=3,
                  step_size_z=sz, step_size_vs=0.4)
    logs = []
    for cut in (None, None, (1, 7, 400, 13)):
        _, sets = _build_world(data, params)
        cs = sets[0]
        assert cs.master_stats()["single_rank_loop"] == 3
        cs.enable_steplog(n_iter * nc)
        if cut is None:
            cs.run(n_iter)
        else:
            done = 0
            for n in cut:
                cs.run(n); done += n
            cs.run(n_iter - done)
        gi, gd = cs.steplog()
        logs.append((gi.copy(), gd.copy(), cs.likelihood_trace()[2].copy()))
    for k in (1, 2):
        assert np.array_equal(logs[0][0], logs[k][0])
        assert np.array_equal(logs[0][1], logs[k][1]), "step log values differ in their bits (run %d)" % k
        assert np.array_equal(logs[0][2], logs[k][2])


def test_single_chain_job_has_no_swap():
    """quirk 1: the reference hangs for n_procs*n_chains == 1; defined as 'no swap, no draws' (oracle too)"""
    from oracle import oracle

    fx, data, params = load_case("c2")
    params = dict(params, n_chains="1")
    job = oracle.Job(params, data); job.run(800)
    _, sets = _build_world(data, params)
    sets[0].run(800)
    it, lk = job.likelihood_trace(0)
    gi, _, gl = sets[0].likelihood_trace()
    assert np.array_equal(gi, it)
    np.testing.assert_allclose(gl, lk, rtol=RTOL_TRACE)
    assert sets[0].rng_state() == job.rng_state(0)


@pytest.mark.parametrize("nc", [19, 32])
def test_many_chains_per_rank(nc):
    """more chains than k_step has waves (8): chains are processed in rounds (32: the LDS stream window at its
    one-iteration look-ahead)"""
    from oracle import oracle

    fx, data, params = load_case("c1")
    params = dict(params, n_procs="1", n_chains=str(nc), n_cool="3", n_iter="1200", n_burn="100", n_interval="7")
    job = oracle.Job(params, data); job.run(1200)
    _, sets = _build_world(data, params)
    sets[0].run(1200)
    it, lk = job.likelihood_trace(0)
    gi, _, gl = sets[0].likelihood_trace()
    assert np.array_equal(gi, it)
    np.testing.assert_allclose(gl, lk, rtol=RTOL_TRACE)
    assert sets[0].rng_state() == job.rng_state(0)
    a, b = sets[0].counts(); oa, ob = job.counts()
    assert np.array_equal(a, oa) and np.array_equal(b, ob)


@pytest.mark.parametrize("transport", ["direct", "rccl"])
def test_rccl_lockstep_loop_equals_single_rank_driver(transport, monkeypatch):
    """The multi-GPU driver loops on a real RCCL process group of one rank must produce the bits of the single-rank
    driver: "direct" = persistent lock-step (htm_chains_run_lockstep_direct: swap records exchanged inside the kernel
    through the inboxes), "rccl" = htm_chains_run_lockstep (one k_mcmc launch + one ncclAllGather per iteration,
    enqueued from C)."""
    import socket

    monkeypatch.setenv("HTM_XCHG", "1" if transport == "direct" else "0")

    import torch
    import torch.distributed as dist

    from hypotremormcmc_amd.parallel import TorchWorld

    fx, data, params = load_case("c2")
    n_iter = 700
    _, a = _build_world(data, params)
    a[0].run(n_iter)
    _, b = _build_world(data, params)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    dist.init_process_group(backend="nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        tw = TorchWorld(b[0])
        if transport == "direct":
            assert tw.direct, "the in-kernel exchange was not set up"
        else:
            assert tw.fast is not None and not tw.direct, "direct RCCL entry not found: the C loop was not exercised"
        tw.run(300)
        tw.run(n_iter - 300)
        torch.cuda.synchronize()
    finally:
        dist.destroy_process_group()
    assert b[0].iterations_done == n_iter
    ia, ca, la = a[0].likelihood_trace(); ib, cb, lb = b[0].likelihood_trace()
    assert np.array_equal(ia, ib)
    np.testing.assert_allclose(la, lb, rtol=1e-13, atol=0)
    assert a[0].rng_state() == b[0].rng_state()
    for c in range(2):
        assert np.array_equal(a[0].state(c).hypo, b[0].state(c).hypo)
        assert a[0].state(c).temp == b[0].state(c).temp


def test_two_kernel_fallback_matches_reference_trace(monkeypatch):
    """HTM_PERSIST=0: k_step + k_full as separate launches (hipGraph) instead of the persistent k_mcmc"""
    monkeypatch.setenv("HTM_PERSIST", "0")
    fx, data, params = load_case("c2")
    fwd, sets = _build_world(data, params)
    sets[0].run(int(params["n_iter"]))
    _check_against_fixture(fx, params, sets)


def test_checkpoint_resume_continues_the_run_and_checks_shapes():
    """run(700) == run(300) -> checkpoint -> a NEW chain set from the same inputs -> restore -> run(400)"""
    from hypotremormcmc_amd._lib import HtmError

    fx, data, params = load_case("c2")
    _, a = _build_world(data, params)
    a[0].run(700)
    _, b = _build_world(data, params)
    b[0].run(300)
    blob = b[0].checkpoint()
    _, c = _build_world(data, params)
    c[0].restore(blob)
    assert c[0].iterations_done == 300 and c[0].rng_state() == b[0].rng_state()
    c[0].run(400)
    assert c[0].iterations_done == 700 and c[0].rng_state() == a[0].rng_state()
    ia, ca, la = a[0].likelihood_trace(); ic, cc, lc = c[0].likelihood_trace()
    keep = ia > 300
    assert np.array_equal(ia[keep], ic) and np.array_equal(ca[keep], cc)
    assert np.array_equal(la[keep], lc)      # (continuing gives the bits of the uninterrupted run: the blob carries each chain's last step)
    for k in range(2):
        sa, sc = a[0].state(k), c[0].state(k)
        assert np.array_equal(sa.hypo, sc.hypo) and np.array_equal(sa.t_corr, sc.t_corr) and sa.temp == sc.temp
        assert sa.log_likelihood == sc.log_likelihood
        assert np.array_equal(sa.n_propose, sc.n_propose) and np.array_equal(sa.n_accept, sc.n_accept)
    # a blob of another shape is refused
    fx3, data3, params3 = load_case("missing")
    _, d = _build_world(data3, params3)
    if (data3.t_obs.shape != data.t_obs.shape) or int(params3["n_chains"]) != int(params["n_chains"]):
        with pytest.raises(HtmError):
            d[0].restore(blob)
    with pytest.raises(HtmError):
        c[0].restore(blob[:64])


def test_c5_size_16_chains_against_oracle():
    """BASELINE configs[4] per-GPU shape in fp64: 10 000 events x 128 stations (2 stations per lane), 16 chains
    (two rounds of chain waves), every recorded log-likelihood and the RNG consumption against the oracle.
    150 iterations: the all-full first iteration, ~240 full evaluations of 41 MB each, ~2 100 partial updates."""
    from hypotremormcmc_amd import synth
    from oracle import oracle

    E, S, nc, n_iter = 10000, 128, 16, 150
    data = synth.make_synthetic(E, S, 5)
    params = dict(synth.DEFAULT_PARAMS, n_procs=1, n_chains=nc, n_cool=2, n_iter=n_iter, n_burn=50, n_interval=3)
    job = oracle.Job(params, data); job.run(n_iter)
    _, sets = _build_world(data, params)
    sets[0].run(n_iter)
    it, lk = job.likelihood_trace(0)
    gi, _, gl = sets[0].likelihood_trace()
    assert len(gi) > 50 and np.array_equal(gi, it)
    np.testing.assert_allclose(gl, lk, rtol=RTOL_TRACE)
    assert sets[0].rng_state() == job.rng_state(0)
    a, b = sets[0].counts(); oa, ob = job.counts()
    assert np.array_equal(a, oa) and np.array_equal(b, ob)
    for c in (0, nc - 1):
        s, o = sets[0].state(c), job.chain(0, c)
        np.testing.assert_allclose(s.hypo, o["hypo"], rtol=1e-11, atol=1e-12)
        assert s.temp == o["temp"]


_SHAPES = [
    # events, stations, chains, cool chains, seed, iterations, parameter overrides
    (7, 5, 3, 1, 11, 900, {}),                                             # fewer events than waves of one worker block
    (64, 64, 8, 2, 12, 900, {"step_size_z": 6.0}),                          # Rayleigh-prior rejections every few steps
    (130, 70, 5, 1, 13, 700, {}),                                          # two stations per lane, odd counts
    (33, 200, 4, 2, 14, 500, {"solve_qs": "F"}),                           # generic (strided) station path
    (500, 64, 8, 8, 15, 600, {"n_interval": 2}),                           # every chain cool: all of them record
    (100, 16, 2, 1, 16, 900, {"use_amp": "F", "step_size_z": 6.0}),
    (1000, 64, 8, 1, 17, 700, {"step_size_z": 8.0, "n_interval": 2}),      # the bench shape with many rejections
    (40, 64, 6, 1, 18, 900, {"solve_t_corr": "F", "solve_a_corr": "F"}),   # only vs / qs steps need the full evaluation
]


@pytest.mark.parametrize("shape", _SHAPES, ids=lambda s: "%dx%d_%dch_seed%d" % (s[:3] + (s[4],)))
def test_assorted_shapes_against_oracle(shape):
    """Shapes and settings the fixtures do not hold -- ragged sizes, rejection-heavy step sizes (orders sent ahead
    then miss their step and the validation repeats passes), switched-off parameter groups -- every recorded
    log-likelihood, the counters, the final states and the RNG consumption against the oracle."""
    from hypotremormcmc_amd import synth
    from oracle import oracle

    E, S, nc, n_cool, seed, n_iter, over = shape
    data = synth.make_synthetic(E, S, seed)
    params = dict(synth.DEFAULT_PARAMS, n_procs=1, n_chains=nc, n_cool=n_cool, n_iter=n_iter, n_burn=n_iter // 3,
                  n_interval=5)
    params.update(over)
    job = oracle.Job(params, data); job.run(n_iter)
    _, sets = _build_world(data, params)
    sets[0].run(n_iter)
    it, lk = job.likelihood_trace(0)
    gi, _, gl = sets[0].likelihood_trace()
    assert len(gi) > 20 and np.array_equal(gi, it)
    np.testing.assert_allclose(gl, lk, rtol=RTOL_TRACE)
    assert sets[0].rng_state() == job.rng_state(0)
    a, b = sets[0].counts(); oa, ob = job.counts()
    assert np.array_equal(a, oa) and np.array_equal(b, ob)
    for c in range(nc):
        s, o = sets[0].state(c), job.chain(0, c)
        np.testing.assert_allclose(s.hypo, o["hypo"], rtol=1e-11, atol=1e-12)
        assert s.temp == o["temp"]


@pytest.mark.parametrize("E,S,nc,seed,sz,n_iter", [(64, 64, 8, 1, 4.0, 3000), (64, 64, 8, 3, 12.0, 3000), (1000, 64, 8, 2, 8.0, 3000),
                                                  (1000, 64, 8, 4, 20.0, 3000), (30, 20, 7, 3, 12.0, 3000),
                                                  (64, 32, 27, 3, 20.0, 4000), (64, 32, 27, 4, 20.0, 9000)])
def test_rejection_heavy_runs_against_oracle(E, S, nc, seed, sz, n_iter):
    """Depth steps several times the prior width: the Rayleigh prior rejects every few steps, each rejection shifts the
    stream positions of the chains behind it, passes are repeated, and the orders role P sent one and two iterations
    ahead miss their steps, are voided or come back into position.  (Each of these configurations exposed a fault of
    the two-ahead orders once: sums overwritten by the chain's own order, a step coming back to the position an order
    was written for after a repeated pass, workers reading a step that was taken back; the two 27-chain runs: with more
    chains than waves a step starts where the wave's previous chain really ended, not where role P predicted, and
    overwrote the very element a two-ahead order told the workers to wait for -- iterations 3139 and 8614 of these
    runs stopped with "workers did not answer" on some boxes.  tools/stress_rejections.py runs the longer version.)"""
    from hypotremormcmc_amd import synth
    from oracle import oracle

    data = synth.make_synthetic(E, S, 100 + seed)
    params = dict(synth.DEFAULT_PARAMS, n_procs=1, n_chains=nc, n_cool=2, n_iter=n_iter, n_burn=n_iter // 2,
                  n_interval=3, step_size_z=sz, step_size_vs=0.4)
    job = oracle.Job(params, data); job.run(n_iter)
    _, sets = _build_world(data, params)
    sets[0].run(n_iter)
    it, lk = job.likelihood_trace(0)
    gi, _, gl = sets[0].likelihood_trace()
    assert len(gi) == len(it) > n_iter // 2 and np.array_equal(gi, it)
    np.testing.assert_allclose(gl, lk, rtol=RTOL_TRACE)
    assert sets[0].rng_state() == job.rng_state(0)
    a, b = sets[0].counts(); oa, ob = job.counts()
    assert np.array_equal(a, oa) and np.array_equal(b, ob)


@pytest.mark.parametrize("lockstep", [False, True])
def test_shared_prior_records_equal_per_chain_records(lockstep, monkeypatch):
    """Every chain of a rank has the same priors and step sizes (the reference's set-up, cls_model.f90 via one parameter file):
    the chain steps then read chain 0's packed records (PriorRec, htm_device.hpp).  HTM_PRIOR_SAME=0 makes every chain read its
    own: same decisions, same draws, same parameter values -- on a rejection-heavy job (Rayleigh rejections read the records on both paths of the orders too)."""
    from hypotremormcmc_amd import synth
    from oracle import oracle

    data = synth.make_synthetic(64, 64, 103)
    n_iter = 1500
    params = dict(synth.DEFAULT_PARAMS, n_procs=1, n_chains=11, n_cool=2, n_iter=n_iter, n_burn=n_iter // 2,
                  n_interval=3, step_size_z=12.0, step_size_vs=0.4)
    from hypotremormcmc_amd.parallel import LocalWorld

    out = []
    for same in ("1", "0"):
        monkeypatch.setenv("HTM_PRIOR_SAME", same)
        _, sets = _build_world(data, params)
        if lockstep:
            LocalWorld(sets).run(n_iter)        # (the lock-step ranks' loop, one rank)
        else:
            sets[0].run(n_iter)
        gi, gc, gl = sets[0].likelihood_trace()
        out.append((gi.copy(), gl.copy(), sets[0].rng_state(), sets[0].counts(), [sets[0].state(c).hypo.copy() for c in range(11)]))
    assert np.array_equal(out[0][0], out[1][0])
    # (a full evaluation's sum is grouped the same way however its order went out -- one or two steps ahead, or late after an epoch
    # change: the free-running master's runs agree bit for bit; the per-launch lock-step loop of LocalWorld sums in its own order)
    if lockstep:
        np.testing.assert_allclose(out[0][1], out[1][1], rtol=1e-12, atol=0)
    else:
        assert np.array_equal(out[0][1], out[1][1])
    assert out[0][2] == out[1][2]
    assert all(np.array_equal(a, b) for a, b in zip(out[0][3], out[1][3]))
    assert all(np.array_equal(a, b) for a, b in zip(out[0][4], out[1][4]))
    job = oracle.Job(params, data); job.run(n_iter)
    it, lk = job.likelihood_trace(0)
    assert np.array_equal(out[0][0], it)
    np.testing.assert_allclose(out[0][1], lk, rtol=RTOL_TRACE)
    assert out[0][2] == job.rng_state(0)


@pytest.mark.parametrize("seed,n_iter", [(3, 4000), (4, 9000)])
def test_workers_put_an_unsatisfiable_wait_aside(seed, n_iter, monkeypatch):
    """Second line of defence of the hand-off: with the chain waves' take-back of disproved orders switched off
    (HTM_DEBUG_NO_DROP=1) the two 27-chain runs above meet the order whose named commit never shows.  The workers put it
    aside after 20 us and serve the other chains; role P's stale check voids it an iteration later: the run ends, equal
    to the oracle, instead of stopping with error -8."""
    from hypotremormcmc_amd import synth
    from oracle import oracle

    monkeypatch.setenv("HTM_DEBUG_NO_DROP", "1")
    data = synth.make_synthetic(64, 32, 100 + seed)
    params = dict(synth.DEFAULT_PARAMS, n_procs=1, n_chains=27, n_cool=2, n_iter=n_iter, n_burn=n_iter // 2,
                  n_interval=3, step_size_z=20.0, step_size_vs=0.4)
    job = oracle.Job(params, data); job.run(n_iter)
    _, sets = _build_world(data, params)
    sets[0].run(n_iter)
    it, lk = job.likelihood_trace(0)
    gi, _, gl = sets[0].likelihood_trace()
    assert np.array_equal(gi, it)
    np.testing.assert_allclose(gl, lk, rtol=RTOL_TRACE)
    assert sets[0].rng_state() == job.rng_state(0)


@pytest.mark.parametrize("lockstep", [False, True])
def test_random_stream_ring_wraps_around(lockstep, monkeypatch):
    """The rank's random stream lives in rings over the absolute position (2^20 by default, i.e. one wrap every
    ~25 000 iterations at 8 chains).  With the smallest ring (2^17) a 5 000-iteration run of 19 chains goes
    round it three to four times, in the persistent driver and in the per-iteration lock-step driver."""
    from hypotremormcmc_amd.parallel import LocalWorld
    from oracle import oracle

    monkeypatch.setenv("HTM_STREAM_CAP", str(1 << 17))
    fx, data, params = load_case("c1")
    n_iter = 5000
    params = dict(params, n_procs="1", n_chains="19", n_cool="3", n_iter=str(n_iter), n_burn="100", n_interval="11")
    job = oracle.Job(params, data); job.run(n_iter)
    _, sets = _build_world(data, params)
    if lockstep:
        LocalWorld(sets).run(n_iter)
    else:
        sets[0].run(n_iter)
    it, lk = job.likelihood_trace(0)
    gi, _, gl = sets[0].likelihood_trace()
    assert np.array_equal(gi, it)
    np.testing.assert_allclose(gl, lk, rtol=RTOL_TRACE)
    assert sets[0].rng_state() == job.rng_state(0)
    a, b = sets[0].counts(); oa, ob = job.counts()
    assert np.array_equal(a, oa) and np.array_equal(b, ob)


def _gloo_device_worker(rank, world, port, name, q, transport="staged"):
    import sys

    sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
    if transport.startswith("pipe-"):           # the lock-step ranks on the pipelined master (csrc/htm_pipe.hpp, opt-in)
        os.environ["HTM_PIPE_LOCK"] = "1"
        transport = transport[5:]
    fails = transport == "direct-fails"
    if fails:
        # fault injection: the LAST rank stops posting its swap records at iteration 40 of the first direct run; every rank's
        # collector gives up after 0.4 s (error -10), all reload the state saved before the run and repeat it on the
        # per-iteration all-gather path (parallel.py, TorchWorld.run)
        os.environ["HTM_XCHG_TIMEOUT_MS"] = "400"
        if rank == world - 1:
            os.environ["HTM_DEBUG_XCHG_FAIL_ITER"] = "40"
        transport = "direct"
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HTM_XCHG="1" if transport.startswith("direct") else "0")
    caps = {}
    if transport == "direct-stops":
        # the ranks leave a launch only TOGETHER: tiny record buffers and a short random-stream ring make some rank ask
        # everybody to stop every few iterations (its buffers) / few thousand (its stream), at different times per rank
        os.environ["HTM_STREAM_CAP"] = "131072"
        n_ch = int(name.split(":")[3]) if name.startswith("synth:") else 8
        caps = dict(lik_capacity=max(24, 3 * n_ch), sample_capacity=max(24, 3 * n_ch))
        transport = "direct"
    import torch.distributed as dist

    from hypotremormcmc_amd import driver
    from hypotremormcmc_amd.obs_data import ObsData
    from hypotremormcmc_amd.parallel import TorchWorld

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        if name.startswith("synth:"):
            # rejection-heavy synthetic job, more chains per rank than chain waves: checked against the oracle's lock-step run
            from hypotremormcmc_amd import synth
            from oracle import oracle

            E, S, nc, seed, n_iter = (int(x) for x in name.split(":")[1:6])
            data = synth.make_synthetic(E, S, 100 + seed)
            params = dict(synth.DEFAULT_PARAMS, n_procs=world, n_chains=nc, n_cool=2, n_iter=n_iter, n_burn=n_iter // 2,
                          n_interval=3, step_size_z=20.0, step_size_vs=0.4)
            job = oracle.Job(params, data); job.run(n_iter)
            oa, ob = job.counts()
            fx = {f"lik_iter_{rank}": job.likelihood_trace(rank)[0], f"lik_{rank}": job.likelihood_trace(rank)[1],
                  "n_propose": oa, "n_accept": ob}
        else:
            fx, data, params = load_case(name)
        obs = ObsData.from_arrays(data.sta_x, data.sta_y, data.t_obs, data.t_stdv, data.a_obs, data.a_stdv)
        fwd, cs = driver.build_rank(params, data.sta_x, data.sta_y, data.sta_z, obs, rank, n_procs=world, device=0, **caps)
        tw = TorchWorld(cs)
        if transport == "direct":
            assert tw.direct, "peer mapping of the inboxes failed"
        else:
            assert tw.host_staged and tw.fast is None and not tw.direct
        n_iter = int(params["n_iter"])
        try:
            tw.run(n_iter // 3)                 # in pieces: a launch ends and the next one picks the exchange up
            tw.run(n_iter - n_iter // 3)
        finally:
            if tw.fell_back is not None and not fails:
                print("rank %d: %s" % (rank, tw.fell_back), file=sys.stderr, flush=True)
        it, ch, lk = cs.likelihood_trace()
        ok = np.array_equal(it, fx[f"lik_iter_{rank}"]) and np.allclose(lk, fx[f"lik_{rank}"], rtol=RTOL_TRACE, atol=0)
        npr, nac = tw.reduce_counts()
        ok = ok and np.array_equal(npr, fx["n_propose"]) and np.array_equal(nac, fx["n_accept"])
        if fails:       # EVERY rank fell back, and says so
            ok = ok and tw.fell_back is not None and not tw.direct
        else:
            ok = ok and tw.fell_back is None
        q.put((rank, bool(ok), len(it)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name,world,transport", [("c1", 2, "staged"), ("timeonly", 3, "staged"), ("c1", 2, "direct"),
                                                  ("timeonly", 3, "direct"), ("rejects", 2, "direct"), ("fixedcorr", 2, "direct"),
                                                  ("rejects", 2, "direct-stops"), ("timeonly", 3, "direct-stops"),
                                                  ("synth:64:32:12:4:6000", 2, "direct"), ("synth:64:32:19:2:4000", 2, "direct-stops"),
                                                  ("c1", 2, "direct-fails"), ("timeonly", 3, "direct-fails"),
                                                  # BASELINE configs[3]'s per-rank shape (1000 x 64, 8 chains per rank) through the in-kernel
                                                  # exchange with 4 processes on the one GPU, against the oracle's lock-step job.  (Five or six
                                                  # processes do not stay co-resident on one GPU: with 5 x 48 blocks -- an even share of every
                                                  # XCD -- a rank's master is still switched out for seconds while all its workers have answered;
                                                  # the runs end in the guarded fall-back)
                                                  ("synth:1000:64:8:2:500", 4, "direct"), ("synth:1000:64:8:3:400", 4, "direct-stops"),
                                                  ("c1", 2, "pipe-direct"), ("timeonly", 3, "pipe-direct-stops"),
                                                  ("synth:64:32:12:4:6000", 2, "pipe-direct")])
def test_torchworld_across_processes_sharing_the_gpu(name, world, transport):
    """TorchWorld + device-resident chains in 2-3 separate processes sharing the one GPU: per-rank traces and the
    reduced counters against the reference's MPI run.  "staged": gloo with host-staged records per iteration;
    "direct": persistent lock-step -- each process's kernel writes its swap records into the other processes'
    inboxes (IPC-mapped device memory, the mapping a multi-GPU node uses over xGMI) and stays resident;
    "direct-stops": the same with record buffers of 24 entries and the shortest random-stream ring, so that launches
    end by a rank's request every few iterations and every rank must leave after the same iteration."""
    import socket

    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gloo_device_worker, args=(r, world, port, name, q, transport)) for r in range(world)]
    for p in procs:
        p.start()
    import queue
    import time

    res, t_end = [], time.time() + 400
    while len(res) < world and time.time() < t_end:
        try:
            res.append(q.get(timeout=2))
        except queue.Empty:
            if any(p.exitcode not in (None, 0) for p in procs):      # a rank died: do not wait for its answer
                break
    for p in procs:
        p.join(timeout=60)
        if p.is_alive():
            p.kill()
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == list(range(world)) and all(r[1] for r in res), res


def test_lockstep_on_the_two_kernel_fallback(monkeypatch):
    """HTM_PERSIST=0 in the multi-rank code path: k_step(advance) -> k_full -> k_step(finish) per iteration"""
    from hypotremormcmc_amd.parallel import LocalWorld

    monkeypatch.setenv("HTM_PERSIST", "0")
    fx, data, params = load_case("c1")
    params = dict(params, n_iter="4000", n_burn="2000")
    fwd, sets = _build_world(data, params)
    LocalWorld(sets).run(4000)
    for r in range(2):
        it, ch, lk = sets[r].likelihood_trace()
        n = len(it)
        assert n > 0 and np.array_equal(it, fx[f"lik_iter_{r}"][:n])
        np.testing.assert_allclose(lk, fx[f"lik_{r}"][:n], rtol=RTOL_TRACE, atol=0)


@pytest.mark.parametrize("E,S,nc,n_iter,prec", [(1000, 64, 8, 30000, "fp64"), (10000, 128, 16, 1500, "fp64"), (10000, 128, 16, 1500, "fp32")])
def test_running_loglik_equals_full_evaluation_of_the_final_state(E, S, nc, n_iter, prec):
    """Size-independent property at BASELINE's full sizes (configs[2] and the configs[4] per-GPU shape): the chain's
    log-likelihood is carried incrementally (one-event updates, src/cls_forward.f90:307-362, refreshed only by accepted full
    evaluations) over thousands of steps; evaluated afresh on the final state (calc_log_likelihood, :268-303) it must be
    the same number -- to 1e-9 relative in fp64, and to the fp32 mode's stated tolerance (3e-6) with the fp32 forward."""
    from hypotremormcmc_amd import driver, synth

    data = synth.make_synthetic(E, S, 1 if E == 1000 else 5)
    params = dict(synth.DEFAULT_PARAMS, n_procs=1, n_chains=nc, n_cool=2, n_iter=n_iter, n_burn=n_iter, n_interval=1000,
                  forward_precision=prec)
    fwd, cs = driver.build_rank(params, data.sta_x, data.sta_y, data.sta_z, _obs(data), 0, n_procs=1)
    cs.run(n_iter)
    assert cs.iterations_done == n_iter
    tol = 1e-9 if prec == "fp64" else 3e-6
    for c in (0, 1, nc // 2, nc - 1):
        s = cs.state(c)
        L = fwd.calc_log_likelihood(s.hypo, s.t_corr, s.vs, s.a_corr, s.qs)
        assert abs(L - s.log_likelihood) <= tol * abs(L), (c, L, s.log_likelihood)
