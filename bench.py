#!/usr/bin/env python3
"""bench.py -- MCMC proposal steps/s of the HIP likelihood inner loop on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2], the one the metric is quoted on): 1 000 synthetic events x 64 stations
(seed 1), 8 chains per GPU, n_cool = 1, temp_high = 200, all solve_* / use_* = T, priors and step sizes of
sample/hypo_tremor.in.  N > 1: one rank per GPU, 8 chains each (weak scaling, BASELINE configs[3] at N = 8),
temperature swap between ANY two chains of the job every iteration, exchanged with one RCCL all-gather.

A "step" for --steps/--warmup is one MCMC iteration of a rank = n_chains proposal steps (propose ->
forward -> judge for every chain, then swap_temperature).  `value` = proposal steps per second over all
ranks; inputs are resident in HBM before the timed region starts.

Besides the driver contract the JSON line carries
  roofline          the dominant kernel of the timed region (k_mcmc), HIP events on its stream; peak = 8 TB/s
  roofline_batch64  the full-evaluation kernel on its own (64 stacked models per launch)
  stages            the two stages of an iteration timed separately on the two-kernel path
  cpu_baseline  the CPU restatement (oracle/) timed on one host core on a bounded sample of the same workload
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_EVENTS, N_STA, N_CHAINS, SEED = 1000, 64, 8, 1
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def algorithmic_bytes(E, S, w=8):
    """bytes per full evaluation and per single-event partial update."""
    b_full = 4 * w * S * E + 3 * w * E + 5 * w * S + w
    b_part = 4 * w * S + 6 * w + 5 * w * S + 2 * w
    return b_full, b_part


def _port_baseline(params, data, seconds_target):
    from oracle import oracle

    job = oracle.Job(params, data)
    job.run(200)                                  # includes the all-full first iteration; not timed
    t0 = time.perf_counter(); job.run(500); dt = time.perf_counter() - t0
    n_it = max(500, int(500 * seconds_target / max(dt, 1e-3)))
    t0 = time.perf_counter(); job.run(n_it); dt = time.perf_counter() - t0
    return n_it * int(params["n_chains"]) / dt, n_it, dt


def _reference_baseline(params, data, n_iter_long, n_iter_short):
    """The compiled reference itself (oracle/_ref, built in the build container from the unmodified Fortran
    sources with AMD flang -O2 + MPICH), 1 MPI rank: two runs of different length, so that set-up and file
    input cancel and only the main loop is priced."""
    import shutil
    import subprocess
    import tempfile

    from hypotremormcmc_amd import synth

    exe = os.path.join(ROOT, "oracle", "_ref", "hypo_tremor_mcmc_ref")
    mpiexec = "/opt/conda/bin/mpiexec"
    if not (os.path.exists(exe) and os.path.exists(mpiexec)):
        return None
    work = tempfile.mkdtemp(prefix="htm_refbase_")
    try:
        synth.write_dataset(work, data)
        times = []
        for n_it in (n_iter_short, n_iter_long):
            synth.write_param_file(os.path.join(work, "run.in"),
                                   **dict(params, n_iter=n_it, n_burn=n_it, n_interval=1000))
            t0 = time.perf_counter()
            subprocess.run([mpiexec, "-np", "1", exe, "run.in"], cwd=work, check=True, timeout=600,
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            times.append(time.perf_counter() - t0)
        dt = times[1] - times[0]
        if dt <= 0:
            return None
        return (n_iter_long - n_iter_short) * int(params["n_chains"]) / dt, dt
    except Exception:
        return None
    finally:
        shutil.rmtree(work, ignore_errors=True)


def cpu_baseline(params, data, seconds_target=10.0):
    """CPU baseline on ONE host core, same workload, bounded sample: the compiled reference when its
    binary travelled with the snapshot (kind = reference), else the C restatement (kind = port)."""
    n_chains = int(params["n_chains"])
    port, n_it, dt = _port_baseline(params, data, seconds_target)
    out = {"value": port, "unit": "proposal steps/s", "cores": 1, "kind": "port",
           "sample": f"{n_it} iterations x {n_chains} chains of the same {data.n_events}x{data.n_sta} workload "
                     f"on 1 core ({dt:.1f} s), oracle/htm_oracle.c (gcc -O2, no fast-math)"}
    ref = _reference_baseline(params, data, 36300, 300)
    if ref is not None:
        out = {"value": ref[0], "unit": "proposal steps/s", "cores": 1, "kind": "reference",
               "sample": f"36000 iterations x {n_chains} chains of the same workload, 1 MPI rank, main loop only "
                         f"({ref[1]:.1f} s; difference of a 36300- and a 300-iteration run), reference Fortran "
                         f"compiled unmodified with AMD flang -O2",
               "port_value": port, "port_sample": out["sample"]}
    return out


def main():
    # Native libraries (RCCL prints a version banner) write to fd 1; the contract is ONE JSON line on stdout.
    # Everything else goes to stderr; the JSON is written to the saved stdout at the end.
    real_stdout = os.dup(1)
    sys.stdout.flush()
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20000)
    ap.add_argument("--warmup", type=int, default=2000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--chains", type=int, default=N_CHAINS)
    ap.add_argument("--events", type=int, default=N_EVENTS)
    ap.add_argument("--stations", type=int, default=N_STA)
    ap.add_argument("--force-lockstep", action="store_true",
                    help="N = 1 only: drive the multi-rank code path (RCCL all-gather per iteration) with one rank")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N > 1 with torch.distributed.run (one rank per GPU)")
        args.gpus = world

    # a cold `import torch` on a fresh box can take minutes: a sign of life on stderr meanwhile (rank 0)
    if rank == 0:
        import threading

        t_start = time.time()

        def _beat():
            while True:
                time.sleep(45)
                sys.stderr.write("[bench] alive, %d s\n" % (time.time() - t_start))
                sys.stderr.flush()

        threading.Thread(target=_beat, daemon=True).start()
    import torch  # device plumbing + torch.distributed only

    from hypotremormcmc_amd import driver, synth
    from hypotremormcmc_amd.obs_data import ObsData

    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or args.force_lockstep:
        import torch.distributed as dist

        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            import socket

            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ.setdefault("MASTER_PORT", str(sk.getsockname()[1]))
            dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    E, S, nc = args.events, args.stations, args.chains
    data = synth.make_synthetic(E, S, SEED)
    params = dict(synth.DEFAULT_PARAMS, n_procs=world, n_chains=nc, n_cool=1,
                  n_iter=args.steps + args.warmup + 10 ** 6, n_burn=10 ** 9, n_interval=1000)
    obs = ObsData.from_arrays(data.sta_x, data.sta_y, data.t_obs, data.t_stdv, data.a_obs, data.a_stdv)
    fwd, cs = driver.build_rank(params, data.sta_x, data.sta_y, data.sta_z, obs, rank, n_procs=world,
                                device=local_rank)

    def sync_all():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if world == 1 and not args.force_lockstep:
        run = cs.run
    else:
        from hypotremormcmc_amd.parallel import TorchWorld

        tw = TorchWorld(cs)
        run = tw.run

    run(args.warmup)
    sync_all()
    t0 = time.perf_counter()
    run(args.steps)
    cs.sync()
    sync_all()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert cs.iterations_done == args.warmup + args.steps

    value = world * nc * args.steps / dt
    b_full, b_part = algorithmic_bytes(E, S)
    out = {
        "metric": "MCMC proposal steps/sec (whole node), 1k events x 64 stn",
        "value": value, "unit": "proposal steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1000.0 * dt / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{E} events x {S} stations, {nc} chains/GPU x {world} GPU(s), swap every "
                               f"iteration, all solve_*/use_* = T (BASELINE configs[{2 if world == 1 else 3}])",
                   "chains_per_gpu": nc, "n_events": E, "n_sta": S, "seed": SEED,
                   "step_definition": "one MCMC iteration of a rank = n_chains proposal steps",
                   "parallelism": f"chains sharded over {world} rank(s); one all-gather of "
                                  f"{8 * (4 + 2 * nc)} B per rank per iteration" if world > 1 else "single rank"},
    }

    if args.force_lockstep:
        out["config"]["parallelism"] = "lock-step path (one RCCL all-gather per iteration), 1 rank"
    if rank == 0 and world == 1 and not args.force_lockstep:
        # ---- dominant kernel of the timed region: k_mcmc (chain master + resident full-evaluation workers).
        # Its launches ARE the timed region; duration from the HIP events htm_chains_run records on the
        # kernels' stream around them, algorithmic bytes from the evaluations they performed.
        st = cs.last_run_stats()
        persistent = os.environ.get("HTM_PERSIST", "1") != "0"
        n_launch = max(1, st["graph_launches"])
        bytes_region = st["full_evals"] * b_full + st["partial_evals"] * b_part
        achieved = bytes_region / (st["device_us"] * 1e-6) / 1e9
        # HBM-side bytes per launch from the committed PMC passes (profiles/*_traffic.json: FETCH_SIZE doubled as
        # MI355X_MICROARCH.md prescribes for gfx950 + WRITE_SIZE, per iteration) x the iterations of a launch
        traffic, traffic_src = None, None
        try:
            import glob

            tf_files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")))
            if tf_files and persistent and (E, S, nc) == (N_EVENTS, N_STA, N_CHAINS):
                with open(tf_files[-1]) as fh:
                    tj = json.load(fh)
                per_it = 2.0 * tj["fetch_bytes_per_iteration_raw"] + tj["write_bytes_per_iteration"]
                traffic = per_it * args.steps / n_launch
                traffic_src = os.path.relpath(tf_files[-1], ROOT)
        except Exception:
            traffic = None
        out["roofline"] = {
            "bound": "hbm",
            "kernel": "k_mcmc<1> (propose + partial/full log-likelihood + judge + swap, persistent)" if persistent
                      else "k_step<1> + k_full<1,false> (graph of the two-kernel path)",
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic, "traffic_source": traffic_src,
            "bytes_per_launch": bytes_region / n_launch, "avg_launch_us": st["device_us"] / n_launch,
            "launches": n_launch, "full_evals": st["full_evals"], "partial_evals": st["partial_evals"],
            "note": "algorithmic bytes = 2 074 568 B per full evaluation + 4 672 B per single-event partial update "
                    "(SURVEY 8d); at 8 chains/GPU an iteration is a dependent chain of ~8 us that touches ~1.7 MB, "
                    "all of it L2/Infinity-Cache resident (traffic = HBM-side bytes per launch from the committed PMC "
                    "passes, ~1 % of the algorithmic bytes): the workload is latency-bound, not HBM-bound (DESIGN.md 5); "
                    "see roofline_batch64 for the full-evaluation kernel on its own",
        }
        # ---- the two stages timed separately (fallback two-kernel path, same arithmetic), HIP events per launch
        n_prof = min(4000, max(500, args.steps // 5))
        prof = cs.profile(n_prof)
        full_avg_us = prof["full_us"] / max(1, prof["full_launches"])
        step_avg_us = prof["step_us"] / max(1, prof["step_launches"])
        out["stages"] = {
            "k_step": {"avg_launch_us": step_avg_us, "launches": prof["step_launches"],
                       "partial_evals": prof["partial_evals"], "algorithmic_bytes": prof["partial_evals"] * b_part},
            "k_full": {"avg_launch_us": full_avg_us, "launches": prof["full_launches"],
                       "full_evals": prof["full_evals"], "algorithmic_bytes": prof["full_evals"] * b_full,
                       "achieved_GBps": prof["full_evals"] * b_full / max(1e-9, prof["full_us"] * 1e-6) / 1e9},
            "time_share_k_step": prof["step_us"] / max(1e-9, prof["step_us"] + prof["full_us"]),
            "iterations_profiled": n_prof,
            "note": "two-kernel path (k_step exits at every hand-over, k_full is its own launch); event-to-event "
                    "times include the dependent-launch boundary",
        }
        # ---- standalone batched full evaluation: 64 models resident in HBM, the kernel's own ceiling
        nb = 64
        g = torch.Generator(device="cpu").manual_seed(3)
        hyp = torch.tensor(data.ev_xyz.reshape(-1), dtype=torch.float64).repeat(nb, 1)
        hyp = (hyp + torch.randn(hyp.shape, generator=g, dtype=torch.float64)).cuda()
        tc = (0.2 * torch.randn(nb, S, generator=g, dtype=torch.float64)).cuda()
        ac = (0.02 * torch.randn(nb, S, generator=g, dtype=torch.float64)).cuda()
        vs = (3.0 + 0.2 * torch.randn(nb, generator=g, dtype=torch.float64)).cuda()
        qs = (250.0 + 30.0 * torch.randn(nb, generator=g, dtype=torch.float64)).cuda()
        L = torch.empty(nb, dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        us = fwd.time_full_batch_dev(nb, hyp.data_ptr(), tc.data_ptr(), vs.data_ptr(), ac.data_ptr(),
                                     qs.data_ptr(), L.data_ptr(), reps=200)
        gbs = nb * b_full / (us * 1e-6) / 1e9
        out["roofline_batch64"] = {"bound": "hbm", "kernel": "k_full<1,true> (64 stacked models per launch)",
                                   "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                                   "traffic": None, "bytes_per_launch": nb * b_full, "evals_per_launch": nb,
                                   "avg_launch_us": us, "evals_per_s": nb / (us * 1e-6),
                                   "note": "200 back-to-back launches of k_full<1,true> + k_sum_partials bracketed by "
                                           "HIP events; inputs resident in HBM"}
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(dict(params, n_procs=1), data)
    elif rank == 0:
        b_step = 0.1 * b_full + 0.9 * b_part
        ach = value * b_step / 1e9 / world
        out["roofline"] = {"bound": "hbm", "kernel": "whole loop (per GPU)", "achieved": ach, "peak": HBM_PEAK_GBS,
                           "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
                           "note": "N > 1: steps/s x 211 662 B expected algorithmic bytes per step / n_gpus; "
                                   "per-kernel figures are reported by the N = 1 run"}
    if rank == 0:
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
