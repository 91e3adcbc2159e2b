"""CPU, multi-process: the N > 1 path (hypotremormcmc_amd.parallel.TorchWorld -- one all-gather of the
ranks' swap records per iteration, identical swap decision on every rank) driven over gloo with
world_size 2 and 3.  The per-rank compute is supplied by the oracle's per-rank lock-step mode (the HIP
chains need a GPU; tests/test_gpu_chains.py covers them with the same record layout), and the result must
reproduce the traces of the reference run under real MPI (golden fixtures c1: 2 ranks, timeonly: 3 ranks)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _worker(rank, world, port, name, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    from hypotremormcmc_amd.parallel import TorchWorld
    from oracle import oracle
    from tests.helpers import OracleRank, load_case

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        fx, data, params = load_case(name)
        assert int(params["n_procs"]) == world
        job = oracle.Job(params, data)
        tw = TorchWorld(OracleRank(job, rank, world))
        tw.run(int(params["n_iter"]))
        it, lk = job.likelihood_trace(rank)
        ok = np.array_equal(it, fx[f"lik_iter_{rank}"]) and np.array_equal(lk, fx[f"lik_{rank}"])
        smp = job.samples(rank)
        ok = ok and np.array_equal(smp["t_corr"], fx[f"t_corr_{rank}"]) and np.array_equal(smp["iter"], fx[f"vs_iter_{rank}"])
        npr, nac = tw.reduce_counts()
        ok = ok and np.array_equal(npr, fx["n_propose"]) and np.array_equal(nac, fx["n_accept"])
        q.put((rank, bool(ok), len(it)))
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("name,world", [("c1", 2), ("timeonly", 3), ("rejects", 2), ("c4", 8)])   # c4 = BASELINE configs[3]
def test_lockstep_allgather_protocol_reproduces_mpi_reference(name, world):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, name, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == list(range(world))
    assert all(r[1] for r in res), res


def test_world_mismatch_is_rejected():
    """TorchWorld refuses a chain set whose n_procs / rank disagree with the process group (the reference
    aborts when n_procs != mpirun -np, src/hypo_tremor_mcmc.f90:65-69)."""
    import torch.distributed as dist

    from hypotremormcmc_amd.parallel import TorchWorld

    port = _free_port()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        class Fake:
            n_procs, rank, record = 2, 0, None
        with pytest.raises(ValueError):
            TorchWorld(Fake())
    finally:
        dist.destroy_process_group()
