"""GPU parity of the PIPELINED chain master (csrc/htm_pipe.hpp, opt-in with HTM_PIPE=1 / HTM_PIPE_LOCK=1): the same criteria as
tests/test_gpu_chains.py -- reference fixtures, the oracle step by step, rejection-heavy runs (every rejection moves the stream
positions its front predicted; toy sizes make its speculation conflict all the time, so the flush path is what these runs
live on), runs in pieces, and the lock-step variant through the in-kernel exchange."""
import socket

import numpy as np
import pytest

from tests.helpers import load_case
from tests.test_gpu_chains import RTOL_TRACE, _build_world, _check_against_fixture

pytestmark = pytest.mark.gpu


@pytest.fixture
def pipe(monkeypatch):
    monkeypatch.setenv("HTM_PIPE", "1")
    monkeypatch.setenv("HTM_PIPE_LOCK", "1")


@pytest.mark.parametrize("name", ["c2", "missing", "c3"])
def test_pipelined_master_matches_reference_trace(name, pipe):
    fx, data, params = load_case(name)
    fwd, sets = _build_world(data, params)
    assert sets[0].master_stats()["single_rank_loop"] == 5, "the pipelined master was not selected"
    sets[0].run(int(params["n_iter"]))
    _check_against_fixture(fx, params, sets)


def test_pipelined_master_steps_vs_oracle(pipe):
    """every step of every chain against the oracle: proposal type/index, prior_ok, accept, full/partial; RNG position"""
    from oracle import oracle

    fx, data, params = load_case("c2")
    n_iter = 1500
    job = oracle.Job(params, data)
    job.enable_steplog(n_iter * 2)
    job.run(n_iter)
    fwd, sets = _build_world(data, params)
    cs = sets[0]
    assert cs.master_stats()["single_rank_loop"] == 5
    cs.enable_steplog(n_iter * 2)
    cs.run(n_iter)
    assert cs.rng_state() == job.rng_state(0)
    oi, od = job.steplog()
    gi, gd = cs.steplog()
    assert len(gi) == len(oi) == n_iter * 2
    assert np.array_equal(gi[:, 0], oi[:, 0]) and np.array_equal(gi[:, 1], oi[:, 2])
    assert np.array_equal(gi[:, 2:7], oi[:, 3:8])
    ok = oi[:, 5] == 1
    np.testing.assert_allclose(gd[:, 0], od[:, 0], rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(gd[ok, 1], od[ok, 1], rtol=RTOL_TRACE)
    np.testing.assert_allclose(gd[:, 2], od[:, 2], rtol=RTOL_TRACE)
    assert np.array_equal(gd[:, 3], od[:, 3])


@pytest.mark.parametrize("E,S,nc,seed,sz,n_iter", [(64, 64, 8, 1, 4.0, 3000), (64, 64, 8, 3, 12.0, 3000), (1000, 64, 8, 2, 8.0, 3000),
                                                  (30, 20, 7, 3, 12.0, 3000), (64, 32, 19, 3, 20.0, 4000), (1000, 128, 16, 5, 2.0, 1500)])
def test_pipelined_master_rejection_heavy_runs_against_oracle(E, S, nc, seed, sz, n_iter, pipe):
    from hypotremormcmc_amd import synth
    from oracle import oracle

    data = synth.make_synthetic(E, S, 100 + seed)
    params = dict(synth.DEFAULT_PARAMS, n_procs=1, n_chains=nc, n_cool=2, n_iter=n_iter, n_burn=n_iter // 2,
                  n_interval=3, step_size_z=sz, step_size_vs=0.4)
    job = oracle.Job(params, data); job.run(n_iter)
    _, sets = _build_world(data, params)
    assert sets[0].master_stats()["single_rank_loop"] == 5
    sets[0].run(n_iter)
    it, lk = job.likelihood_trace(0)
    gi, _, gl = sets[0].likelihood_trace()
    assert len(gi) == len(it) > n_iter // 2 and np.array_equal(gi, it)
    np.testing.assert_allclose(gl, lk, rtol=RTOL_TRACE)
    assert sets[0].rng_state() == job.rng_state(0)
    a, b = sets[0].counts(); oa, ob = job.counts()
    assert np.array_equal(a, oa) and np.array_equal(b, ob)


def test_pipelined_master_in_pieces_is_the_uninterrupted_run(pipe):
    """launch boundaries (small record buffers, short calls) change nothing but the last bits of a full evaluation right after one:
    which events the workers leave out is a function of the steps of the same launch"""
    fx, data, params = load_case("c2")
    n_iter = 2400
    _, a = _build_world(data, params)
    a[0].run(n_iter)
    _, b = _build_world(data, params, lik_capacity=64, sample_capacity=16)
    done = 0
    for piece in (1, 7, 300, 5, 1000, 87, 1000):
        b[0].run(piece); done += piece
    assert done == n_iter
    ia, ca, la = a[0].likelihood_trace(); ib, cb, lb = b[0].likelihood_trace()
    assert np.array_equal(ia, ib) and np.array_equal(ca, cb)
    np.testing.assert_allclose(la, lb, rtol=1e-12, atol=0)
    assert a[0].rng_state() == b[0].rng_state()
    for c in range(int(params["n_chains"])):
        assert np.array_equal(a[0].state(c).hypo, b[0].state(c).hypo)
        assert np.array_equal(a[0].state(c).n_accept, b[0].state(c).n_accept)


def test_pipelined_master_twice_gives_the_same_bits(pipe):
    """two runs of one job: identical traces, bit for bit (a full evaluation's association is a function of stream and state)"""
    from hypotremormcmc_amd import synth

    data = synth.make_synthetic(1000, 64, 1)
    params = dict(synth.DEFAULT_PARAMS, n_procs=1, n_chains=8, n_cool=1, n_iter=6000, n_burn=100, n_interval=7)
    out = []
    for _ in range(2):
        _, s = _build_world(data, params)
        s[0].run(6000)
        out.append(s[0].likelihood_trace())
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][2], out[1][2])


def test_pipelined_lockstep_rank_equals_single_rank_driver(pipe, monkeypatch):
    """one rank through MODE_LOCKRUN on the pipelined master (swap records through the inbox, htm_chains_run_lockstep_direct)
    against the single-rank driver"""
    monkeypatch.setenv("HTM_XCHG", "1")
    import torch
    import torch.distributed as dist

    from hypotremormcmc_amd.parallel import TorchWorld

    fx, data, params = load_case("c2")
    n_iter = 700
    _, a = _build_world(data, params)
    a[0].run(n_iter)
    _, b = _build_world(data, params)
    assert b[0].master_stats()["lockstep_loop"] == 6
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    dist.init_process_group(backend="nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        tw = TorchWorld(b[0])
        assert tw.direct, "the in-kernel exchange was not set up"
        tw.run(300)
        tw.run(n_iter - 300)
        torch.cuda.synchronize()
    finally:
        dist.destroy_process_group()
    assert b[0].iterations_done == n_iter
    ia, ca, la = a[0].likelihood_trace(); ib, cb, lb = b[0].likelihood_trace()
    assert np.array_equal(ia, ib)
    np.testing.assert_allclose(la, lb, rtol=1e-12, atol=0)
    assert a[0].rng_state() == b[0].rng_state()
    for c in range(2):
        assert np.array_equal(a[0].state(c).hypo, b[0].state(c).hypo)
        assert a[0].state(c).temp == b[0].state(c).temp
