"""CPU: the synchronisation protocol of the free-running chain master (csrc/htm_flow.hpp) as a model -- tools/flow_protocol_sim.py
runs its rules (prog / done / epoch / anchor, the turn rule, re-prediction from the anchor) as coroutines under a random
scheduler and compares every step's start position, the final states, temperatures and stream position with the serial loop of
the reference (src/hypo_tremor_mcmc.f90:236-284; draws per step as src/cls_mcmc.f90:193, the swap as src/cls_parallel.f90:121-136).
A change of the protocol is tried here before it goes to the GPU."""
import importlib.util
import os

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _sim():
    spec = importlib.util.spec_from_file_location("flow_protocol_sim", os.path.join(ROOT, "tools", "flow_protocol_sim.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("first", [1, 151, 301, 451])
def test_free_running_protocol_equals_the_serial_loop(first):
    sim = _sim()
    for seed in range(first, first + 150):
        assert sim.run_case(seed), "seed %d differs from the serial loop" % seed


@pytest.mark.parametrize("first", [1, 201, 401])
def test_lock_step_ranks_protocol_equals_the_serial_ranks(first):
    """the lock-step ranks' part (flow_post_chain / flow_post_header / flow_lock_swap / flow_lock_finish): 1-4 ranks with their own
    streams, per-chain (T, L) records and per-rank headers in eight-slot inbox rings, the pair drawn by rank 0, the judge draw by
    the pair's first chain's rank (src/cls_parallel.f90:121-213), only the pair's two chains waiting for anybody, the lost bet on
    the draw, a stop request by any rank that ends the job two iterations on, a rank that is scheduled rarely"""
    sim = _sim()
    for seed in range(first, first + 200):
        assert sim.run_case_lock(seed), "lock-step seed %d differs from the serial ranks" % seed
