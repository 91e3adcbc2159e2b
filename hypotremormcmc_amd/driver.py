"""Step-5 driver on the HIP path: same set-up order, RNG consumption, main-loop semantics and output
files as the reference's `program main` (src/hypo_tremor_mcmc.f90).

    python -m hypotremormcmc_amd.driver <parameter file>           # n_procs = 1
    torchrun --nproc-per-node N -m hypotremormcmc_amd.driver <parameter file>   # n_procs = N, 1 rank per GPU

The working directory must hold the station file, selected_win.dat and the opt_data.NNNNNN.dat files, as
for the reference.  Output: hypo.RR.out, t_corr.RR.out, vs.RR.out, a_corr.RR.out, qs.RR.out, likelihoodRR.out
(stream-unformatted records, native endianness) and proposal_count.txt.
"""
from __future__ import annotations

import math
import os
import struct
import sys

import numpy as np

from .chains import LABELS, ChainSet
from .forward import Forward
from .mod_random import Xorshift128
from .model import Model
from .obs_data import ObsData
from .param import Param

EPS = sys.float_info.epsilon  # epsilon(1.d0)
SEEDS = (5551111, 453222, 4444431, 6765)  # src/hypo_tremor_mcmc.f90:72


def _get(params, key):
    if isinstance(params, Param):
        return params.values[key]
    v = params[key]
    if key.startswith("solve_") or key.startswith("use_"):
        return v if isinstance(v, bool) else str(v).strip().upper().lstrip(".").startswith("T")
    if key in ("n_procs", "n_iter", "n_burn", "n_interval", "n_chains", "n_cool"):
        return int(v)
    if isinstance(v, str):
        return float(v.lower().replace("d", "e"))
    return float(v)


def read_selected_win(path="selected_win.dat"):
    if not os.path.exists(path):
        raise SystemExit("cannot open selected_win.dat")
    ids = []
    with open(path) as f:
        for line in f:
            tok = line.split()
            if len(tok) >= 2:
                ids.append(int(tok[0]))
    return ids


def build_initial_models(params, n_sta, n_events, x_mu, y_mu, rank):
    """src/hypo_tremor_mcmc.f90:72,:120-211 -- returns (models per chain, temps, rng) with the rank's RNG
    advanced exactly as the reference's set-up does."""
    g = lambda k: _get(params, k)
    rng = Xorshift128(rank, SEEDS)
    models, temps = [], []
    prior_t_corr = g("prior_t_corr") if _has(params, "prior_t_corr") else 0.0
    prior_a_corr = g("prior_a_corr") if _has(params, "prior_a_corr") else 0.0
    for j in range(1, g("n_chains") + 1):
        t_corr = Model(n_sta)
        if g("solve_t_corr"):
            for i in range(1, n_sta + 1):
                t_corr.set_prior(i, prior_t_corr, g("prior_width_t_corr"))
                t_corr.set_perturb(i, g("step_size_t_corr"))
            t_corr.generate_model(rng)
        else:
            t_corr.x[:] = prior_t_corr
        a_corr = Model(n_sta)
        if g("solve_a_corr"):
            for i in range(1, n_sta + 1):
                a_corr.set_prior(i, prior_a_corr, g("prior_width_a_corr"))
                a_corr.set_perturb(i, g("step_size_a_corr"))
            a_corr.generate_model(rng)
        else:
            a_corr.x[:] = prior_a_corr
        hypo = Model(3 * n_events)
        for i in range(1, n_events + 1):
            hypo.set_prior(3 * i - 2, x_mu[i - 1], g("prior_width_xy"))
            hypo.set_prior(3 * i - 1, y_mu[i - 1], g("prior_width_xy"))
            hypo.set_prior(3 * i, g("prior_z"), g("prior_width_z"), prior_type=1)
            hypo.set_perturb(3 * i - 2, g("step_size_xy"))
            hypo.set_perturb(3 * i - 1, g("step_size_xy"))
            hypo.set_perturb(3 * i, g("step_size_z"))
        hypo.generate_model(rng)
        vs = Model(1)
        vs.set_prior(1, g("prior_vs"), g("prior_width_vs")); vs.set_perturb(1, g("step_size_vs")); vs.set_x(1, g("prior_vs"))
        qs = Model(1)
        qs.set_prior(1, g("prior_qs"), g("prior_width_qs")); qs.set_perturb(1, g("step_size_qs")); qs.set_x(1, g("prior_qs"))
        models.append(dict(hypo=hypo, t_corr=t_corr, vs=vs, a_corr=a_corr, qs=qs))
        if j <= g("n_cool"):
            temps.append(1.0)
        else:  # :205-206
            temps.append(math.exp((rng.rand_u() * (1.0 - EPS) + EPS) * math.log(g("temp_high"))))
    return models, temps, rng


def _has(params, key):
    return key in (params.values if isinstance(params, Param) else params)


def _raw(params, key):
    return (params.values if isinstance(params, Param) else params)[key]


def build_rank(params, sta_x, sta_y, sta_z, obs, rank, n_procs=None, device=0, fwd=None, **caps):
    """Forward + ChainSet of one rank.  `obs` needs get_t_obs().. and make_initial_guess()."""
    n_sta, n_events = obs.n_sta, obs.n_events
    g = lambda k: _get(params, k)
    if fwd is None:
        # optional key (not in the reference's grammar): forward_precision = fp32 | fp64; HTM_FORWARD_PRECISION overrides
        prec = os.environ.get("HTM_FORWARD_PRECISION") or (str(_raw(params, "forward_precision")) if _has(params, "forward_precision") else "fp64")
        fwd = Forward(n_sta=n_sta, n_events=n_events, sta_x=sta_x, sta_y=sta_y, sta_z=sta_z, obs=obs,
                      use_amp=g("use_amp"), use_time=g("use_time"), device=device, forward_precision=prec)
    x_mu, y_mu = obs.make_initial_guess()
    models, temps, rng = build_initial_models(params, n_sta, n_events, x_mu, y_mu, rank)
    cs = ChainSet(fwd, models, temps, rng.state, n_procs=n_procs if n_procs is not None else g("n_procs"),
                  rank=rank, solve_vs=g("solve_vs"), solve_t_corr=g("solve_t_corr"), solve_qs=g("solve_qs"),
                  solve_a_corr=g("solve_a_corr"), n_burn=g("n_burn"), n_interval=g("n_interval"), **caps)
    return fwd, cs


# ---------------------------------------------------------------------------------------------------
# output files, src/hypo_tremor_mcmc.f90:216-233,:270-280 (stream access, unformatted: packed records)
# ---------------------------------------------------------------------------------------------------
def sample_byte_order(endian=None):
    """'<' or '>' for the unformatted output files.  The reference's stock gfortran Makefile builds with
    -fconvert=big-endian (src/Makefile:7-9), so a step 6 built that way expects big-endian sample files; any other
    build reads native little-endian.  HTM_SAMPLE_ENDIAN=big|little (default little) picks the writer's order."""
    e = (endian or os.environ.get("HTM_SAMPLE_ENDIAN") or "little").strip().lower()
    if e in ("big", "big_endian", "big-endian", ">"):
        return ">"
    if e in ("little", "little_endian", "little-endian", "native", "<"):
        return "<"
    raise ValueError("HTM_SAMPLE_ENDIAN must be big or little, got %r" % e)


def write_outputs(directory, rank, cs: ChainSet, endian=None):
    bo = sample_byte_order(endian)
    it, _, lk = cs.likelihood_trace()
    with open(os.path.join(directory, "likelihood%02d.out" % rank), "wb") as f:
        for i, v in zip(it.tolist(), lk.tolist()):
            f.write(struct.pack(bo + "id", i, v))
    smp = cs.samples()
    for name, key in (("vs", "vs"), ("qs", "qs"), ("t_corr", "t_corr"), ("a_corr", "a_corr"), ("hypo", "hypo")):
        with open(os.path.join(directory, "%s.%02d.out" % (name, rank)), "wb") as f:
            for k in range(len(smp["iter"])):
                f.write(struct.pack(bo + "i", int(smp["iter"][k])))
                f.write(np.atleast_1d(smp[key][k]).astype(bo + "f8").tobytes())


def write_proposal_count(path, n_propose, n_accept):
    """src/cls_parallel.f90:270-278: '(A,2I10)' with the label cut to character(5) (src/cls_mcmc.f90:362)."""
    with open(path, "w") as f:
        for lab, a, b in zip(LABELS, n_propose, n_accept):
            f.write('"%-5.5s"%10d%10d\n' % (lab, int(a), int(b)))


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    if len(argv) != 1:
        raise SystemExit("USAGE: hypo_tremor_mcmc [parameter file]")
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    para = Param(argv[0], verb=(rank == 0), from_where="mcmc")
    if para.values["n_procs"] != world:
        if rank == 0:
            print("ERROR: n_procs in parameter file must be equal to that is given in the command line")
        raise SystemExit(1)
    win_id = read_selected_win("selected_win.dat")
    obs = ObsData(win_id, para.n_stations, para.sta_x, para.sta_y, verb=(rank == 0))
    fwd, cs = build_rank(para, para.sta_x, para.sta_y, para.sta_z, obs, rank, n_procs=world, device=local_rank)
    n_iter = para.values["n_iter"]
    print(" start MCMC")
    if world == 1:
        cs.run(n_iter)
        npr, nac = cs.counts()
    else:
        from .parallel import TorchWorld

        tw = TorchWorld(cs)
        tw.run(n_iter)
        npr, nac = tw.reduce_counts()
    write_outputs(".", rank, cs)
    if rank == 0:
        write_proposal_count("proposal_count.txt", npr, nac)


if __name__ == "__main__":
    main()
