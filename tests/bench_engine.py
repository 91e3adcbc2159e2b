"""Per-rank engine for bench.py's test hook (HTM_BENCH_TEST_ENGINE=tests.bench_engine:make): the ranks are computed
by the CPU oracle's lock-step mode and exchanged over gloo, so that `python bench.py --gpus N` -- launcher,
rendezvous, barrier/timing protocol, JSON contract -- can be exercised on a box without a GPU.  Test
infrastructure only; bench.py marks such a line as a test double."""
from oracle import oracle

from tests.helpers import OracleRank


class _Engine:
    def __init__(self, params, data, rank, world):
        self.job = oracle.Job(params, data)
        self.world, self.rank, self.iterations_done = world, rank, 0
        self.tw = None
        if world > 1:
            from hypotremormcmc_amd.parallel import TorchWorld

            self.tw = TorchWorld(OracleRank(self.job, rank, world))

    def run(self, n_iter):
        if self.tw is not None:
            self.tw.run(n_iter)
        else:
            self.job.run(n_iter)
        self.iterations_done += n_iter

    def sync(self):
        pass


def make(params, data, rank, world):
    return _Engine(params, data, rank, world)
