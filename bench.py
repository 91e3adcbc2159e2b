#!/usr/bin/env python3
"""bench.py -- MCMC proposal steps/s of the HIP likelihood inner loop on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]          # N > 1: spawns its own N ranks (one per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2], the one the metric is quoted on): 1 000 synthetic events x 64 stations
(seed 1), 8 chains per GPU, n_cool = 1, temp_high = 200, all solve_* / use_* = T, priors and step sizes of
sample/hypo_tremor.in.  N > 1: one rank per GPU, 8 chains each (weak scaling, BASELINE configs[3] at N = 8),
temperature swap between ANY two chains of the job every iteration (reference src/cls_parallel.f90:100-216).

A bench "step" (--steps / --warmup) is a fixed BLOCK of main-loop iterations of every rank (reference
src/hypo_tremor_mcmc.f90:236-284), ITERS_PER_STEP = 65 536 by default: one iteration = n_chains proposal steps
(propose -> forward -> judge for every chain, then swap_temperature).  So the driver's `--steps 20 --warmup 5`
times 1 310 720 iterations = 10.5 M proposal steps per GPU: several seconds of the persistent kernel, not a
launch prologue.  `value` = proposal steps per second over all ranks; inputs are resident in HBM before
the timed region starts; the region is bracketed by a barrier + device synchronisation on both sides.

Besides the driver contract the JSON line carries
  roofline          the dominant kernel of the timed region (k_mcmc), HIP events on its stream; peak = 8 TB/s
  roofline_batch64  the full-evaluation kernel on its own (64 stacked models per launch)            [N = 1]
  stages            the two stages of an iteration timed separately on the two-kernel path           [N = 1]
  cpu_baseline      the compiled reference (oracle/_ref; else the C restatement) timed on the host cores of
                    this box on a bounded sample of the same workload: all chains of the job spread over
                    min(chains, cores) MPI ranks, plus the one-core figure
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_EVENTS, N_STA, N_CHAINS, SEED = 1000, 64, 8, 1
ITERS_PER_STEP = 65536     # 20 timed steps = 1.3 M iterations: several seconds of the persistent kernel (the driver's GPU sampler sees it)
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def algorithmic_bytes(E, S, w=8):
    """bytes per full evaluation and per single-event partial update; w = bytes of one element of the
    four observation streams (8, or 4 with the fp32 forward -- everything else stays fp64)."""
    b_full = 4 * w * S * E + 3 * 8 * E + 5 * 8 * S + 8
    b_part = 4 * w * S + 6 * 8 + 5 * 8 * S + 2 * 8
    return b_full, b_part


# ---------------------------------------------------------------------------------------------------
# CPU baseline (the only part of this file that touches oracle/)
# ---------------------------------------------------------------------------------------------------
def _port_baseline(params, data, seconds_target):
    from oracle import oracle

    job = oracle.Job(params, data)
    job.run(200)                                  # includes the all-full first iteration; not timed
    t0 = time.perf_counter(); job.run(300); dt = time.perf_counter() - t0
    n_it = max(300, int(300 * seconds_target / max(dt, 1e-3)))
    t0 = time.perf_counter(); job.run(n_it); dt = time.perf_counter() - t0
    return n_it * int(params["n_chains"]) / dt, n_it, dt


def _host_cores():
    try:
        return len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        return os.cpu_count() or 1


def _reference_run(work, exe, params, ranks, n_it, timeout):
    from hypotremormcmc_amd import synth

    ref_params = {k: v for k, v in params.items() if k != "forward_precision"}      # the reference stops on a key it does not know
    synth.write_param_file(os.path.join(work, "run.in"), **dict(ref_params, n_iter=n_it, n_burn=n_it, n_interval=1000))
    t0 = time.perf_counter()
    subprocess.run(["/opt/conda/bin/mpiexec", "-np", str(ranks), exe, "run.in"], cwd=work, check=True, timeout=timeout,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return time.perf_counter() - t0


def _reference_baseline(params, data, ranks, chains_per_rank, seconds_target, port_steps_per_s):
    """The compiled reference itself (oracle/_ref: the unmodified Fortran sources, AMD flang -O2 + MPICH) on
    `ranks` MPI ranks x `chains_per_rank` chains: runs of different length, so that set-up and file input
    cancel and only the main loop (src/hypo_tremor_mcmc.f90:236-284) is priced.  The long run's length comes from
    a calibration pair, capped by what the C port's one-core rate allows (a noisy calibration must not turn a
    bounded sample into minutes of CPU work); every run has its own timeout.
    Returns (proposal steps/s, iterations, seconds) or None."""
    import shutil
    import tempfile

    from hypotremormcmc_amd import synth

    exe = os.path.join(ROOT, "oracle", "_ref", "hypo_tremor_mcmc_ref")
    if not (os.path.exists(exe) and os.path.exists("/opt/conda/bin/mpiexec")):
        return None
    work = tempfile.mkdtemp(prefix="htm_refbase_")
    try:
        synth.write_dataset(work, data)
        p = dict(params, n_procs=ranks, n_chains=chains_per_rank)
        n0, n1 = 200, 2200
        _reference_run(work, exe, p, ranks, n0, 120)                   # warms the file cache; not timed
        t_short = _reference_run(work, exe, p, ranks, n0, 120)
        t_mid = _reference_run(work, exe, p, ranks, n1, 240)
        rate_cap = 3.0 * port_steps_per_s / chains_per_rank           # iterations / s a rank can plausibly reach
        rate = min((n1 - n0) / max(t_mid - t_short, 1e-3), rate_cap)
        n2 = n0 + int(max(rate * seconds_target, 2000))
        t_long = _reference_run(work, exe, p, ranks, n2, 60 + 6 * seconds_target)
        dt = t_long - t_short
        if dt <= 0:
            return None
        return (n2 - n0) * ranks * chains_per_rank / dt, n2 - n0, dt
    except Exception as exc:                                          # report the port instead
        sys.stderr.write(f"[bench] reference baseline failed: {exc}\n")
        return None
    finally:
        shutil.rmtree(work, ignore_errors=True)


def cpu_baseline(params, data, total_chains, seconds_target=10.0):
    """CPU baseline of the same job on this box's host cores (bounded samples).  `value` = the reference on
    min(total_chains, cores) MPI ranks (the reference's own parallelism: chains over ranks,
    README.md:62-66), `one_core` = the same chains of ONE GPU's share on one rank."""
    nc = int(params["n_chains"])
    cores = _host_cores()
    ranks = 1
    for r in range(min(total_chains, cores), 0, -1):
        if total_chains % r == 0:
            ranks = r
            break
    port, n_it, dt = _port_baseline(dict(params, n_procs=1), data, min(5.0, seconds_target))
    out = {"value": port, "unit": "proposal steps/s", "cores": 1, "kind": "port",
           "sample": f"{n_it} iterations x {nc} chains of the same {data.n_events}x{data.n_sta} workload "
                     f"on 1 core ({dt:.1f} s), oracle/htm_oracle.c (gcc -O2, no fast-math)"}
    one = _reference_baseline(params, data, 1, nc, seconds_target, port)
    multi = _reference_baseline(params, data, ranks, total_chains // ranks, seconds_target, port) if ranks > 1 else one
    if multi is not None:
        out = {"value": multi[0], "unit": "proposal steps/s", "cores": ranks, "kind": "reference",
               "host_cores_available": cores,
               "sample": f"{multi[1]} iterations x {total_chains} chains ({ranks} MPI ranks x {total_chains // ranks} "
                         f"chain(s), swap every iteration) of the same workload, main loop only ({multi[2]:.1f} s; "
                         f"difference of two run lengths), reference Fortran compiled unmodified with AMD flang -O2 + MPICH",
               "port_value": port, "port_sample": out["sample"]}
        if one is not None:
            out["one_core"] = {"value": one[0], "cores": 1,
                               "sample": f"{one[1]} iterations x {nc} chains, 1 MPI rank ({one[2]:.1f} s)"}
    return out


# ---------------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` (N > 1) starts its own ranks -- the reference is launched as
# `mpirun -np N` (README.md:62-66); here one rank per GPU under torch.distributed.run
# ---------------------------------------------------------------------------------------------------
def free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_ranks(n, argv):
    """Runs before anything in this process has touched the GPU (no torch import, no HIP call): the ranks are
    child processes, this process only forwards their output and exit code."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--iters-per-step", type=int, default=ITERS_PER_STEP,
                    help="main-loop iterations per bench step (default %(default)s)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--chains", type=int, default=N_CHAINS)
    ap.add_argument("--events", type=int, default=N_EVENTS)
    ap.add_argument("--stations", type=int, default=N_STA)
    ap.add_argument("--forward-precision", choices=["fp64", "fp32"], default="fp64",
                    help="fp32: single-precision forward model, fp64 sums and accept (BASELINE configs[4]; statistical tolerance)")
    ap.add_argument("--force-lockstep", action="store_true",
                    help="N = 1 only: drive the multi-rank code path (swap records exchanged every iteration) with one rank")
    return ap.parse_args(argv)


def main(argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, argv))

    # Native libraries (RCCL prints a version banner) write to fd 1; the contract is ONE JSON line on stdout.
    # Everything else goes to stderr; the JSON is written to the saved stdout at the end.
    real_stdout = os.dup(1)
    sys.stdout.flush()
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
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

    from hypotremormcmc_amd import synth

    E, S, nc = args.events, args.stations, args.chains
    ips = max(1, args.iters_per_step)
    n_warm, n_timed = args.warmup * ips, args.steps * ips
    data = synth.make_synthetic(E, S, SEED)
    params = dict(synth.DEFAULT_PARAMS, n_procs=world, n_chains=nc, n_cool=1,
                  n_iter=n_warm + n_timed + 10 ** 6, n_burn=2 * 10 ** 9, n_interval=1000,
                  forward_precision=args.forward_precision)

    # Test hook (tests/test_bench_launcher.py): "module:function" supplying the per-rank engine, so that launcher,
    # rendezvous, timing protocol and the JSON contract run on a box without a GPU (gloo).  Never a measurement:
    # the line is marked.  The product engine below has no fallback -- without the HIP library it raises.
    test_engine = os.environ.get("HTM_BENCH_TEST_ENGINE")
    dist = None
    lockstep = world > 1 or args.force_lockstep
    if test_engine:
        import importlib

        mod, fn = test_engine.split(":")
        if world > 1:
            import torch.distributed as dist

            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        eng =getattr(importlib.import_module(mod), fn)(params, data, rank, world)
        run, sync, device_sync = eng.run, eng.sync, (lambda: None)
        fwd = cs = None
    else:
        from hypotremormcmc_amd import driver
        from hypotremormcmc_amd.obs_data import ObsData

        # Rehearsal on a box with fewer GPUs than ranks (HTM_BENCH_ONE_GPU=1): every rank on device 0, rendezvous over gloo
        # (RCCL refuses two ranks on one device) -- the launcher, the inbox exchange between processes and the timing
        # protocol are the real ones, the number is not a scaling measurement and the line says so.
        one_gpu = os.environ.get("HTM_BENCH_ONE_GPU") == "1"
        if one_gpu:
            local_rank = 0
            os.environ.setdefault("HTM_RANKS_PER_GPU", str(world))   # the ranks' kernels must co-reside: each takes its share of the CUs
        backend = "gloo" if one_gpu else "nccl"
        torch.cuda.set_device(local_rank)
        if lockstep:
            import torch.distributed as dist

            if world == 1:
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                os.environ.setdefault("MASTER_PORT", str(free_port()))
                dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", local_rank))
            elif one_gpu:
                dist.init_process_group(backend="gloo")
            else:
                dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        obs = ObsData.from_arrays(data.sta_x, data.sta_y, data.t_obs, data.t_stdv, data.a_obs, data.a_stdv)
        fwd, cs = driver.build_rank(params, data.sta_x, data.sta_y, data.sta_z, obs, rank, n_procs=world,
                                    device=local_rank)
        if not lockstep:
            run = cs.run
        else:
            from hypotremormcmc_amd.parallel import TorchWorld

            tw = TorchWorld(cs)
            run = tw.run
        sync, device_sync = cs.sync, torch.cuda.synchronize

    def fence():
        sync()                  # the chains' own stream: every iteration asked for has completed
        if dist is not None:
            dist.barrier()
        device_sync()

    run(n_warm)
    fence()
    t0 = time.perf_counter()
    run(n_timed)
    fence()
    dt = time.perf_counter() - t0
    rank_dts = [dt]
    if dist is not None:
        # every rank's own time of the region: the line's time is the MAX over ranks, config carries them all
        tdev = "cpu" if (test_engine or dist.get_backend() == "gloo") else "cuda"
        t = torch.tensor([dt], dtype=torch.float64, device=tdev)
        tall = torch.zeros(world, dtype=torch.float64, device=tdev)
        dist.all_gather_into_tensor(tall, t)
        rank_dts = [float(x) for x in tall.cpu().tolist()]
        dt = max(rank_dts)
    done = eng.iterations_done if test_engine else cs.iterations_done
    assert done == n_warm + n_timed, (done, n_warm + n_timed)

    value = world * nc * n_timed / dt
    b_full, b_part = algorithmic_bytes(E, S, 4 if args.forward_precision == "fp32" else 8)
    out = {
        "metric": "MCMC proposal steps/sec (whole node), 1k events x 64 stn",
        "value": value, "unit": "proposal steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1000.0 * dt / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64" if args.forward_precision == "fp64" else "f32 forward / f64 sums + accept", "data": "synthetic",
        "config": {"workload": f"{E} events x {S} stations, {nc} chains/GPU x {world} GPU(s), swap every "
                               f"iteration, all solve_*/use_* = T (BASELINE configs[{2 if world == 1 else 3}])",
                   "chains_per_gpu": nc, "n_events": E, "n_sta": S, "seed": SEED,
                   "step_definition": f"one bench step = {ips} main-loop iterations of every rank = {ips * nc} proposal "
                                      f"steps per GPU (one iteration = n_chains x [propose -> forward -> judge] + swap_temperature)",
                   "iterations_per_step": ips, "iterations_timed": n_timed, "us_per_iteration": 1e6 * dt / n_timed,
                   "parallelism": f"chains sharded over {world} rank(s), one per GPU; swap records of "
                                  f"{8 * (4 + 2 * nc)} B per rank exchanged every iteration" if world > 1 else "single rank"},
    }
    if args.force_lockstep:
        out["config"]["parallelism"] = "lock-step path (swap records exchanged every iteration), 1 rank"
    if world > 1:
        out["config"]["us_per_iteration_by_rank"] = [1e6 * x / n_timed for x in rank_dts]
        out["config"]["max_skew_us_per_iteration"] = 1e6 * (max(rank_dts) - min(rank_dts)) / n_timed
    if lockstep and not test_engine:
        out["config"]["swap_transport"] = tw.transport_name()      # the transport the timed region really ran on
        if tw.fell_back:
            out["config"]["swap_transport_note"] = tw.fell_back
        if os.environ.get("HTM_BENCH_ONE_GPU") == "1" and world > 1:
            out["data"] = "synthetic (rehearsal: all ranks share ONE GPU, not a scaling measurement)"
    if test_engine:
        out["engine"] = f"{test_engine} (test double on CPU: launcher/protocol check, NOT a measurement)"
        out["data"] = "synthetic (test double)"

    if rank == 0 and not test_engine:
        # ---- dominant kernel of the timed region: k_mcmc (chain master + resident full-evaluation workers).
        # Its launches ARE the timed region; duration from the HIP events the library records on the kernels'
        # stream around them, algorithmic bytes from the evaluations they performed.
        st = cs.last_run_stats()
        try:
            out["config"]["orders_put_aside_by_worker_0"] = cs.handoff_stats()["orders_put_aside"]   # health of the hand-off: 0
        except Exception:      # an older build of the library (tools/ab.sh)
            pass
        persistent = os.environ.get("HTM_PERSIST", "1") != "0"
        n_launch = max(1, st["graph_launches"])
        bytes_region = st["full_evals"] * b_full + st["partial_evals"] * b_part
        achieved = bytes_region / max(st["device_us"] * 1e-6, 1e-9) / 1e9
        # HBM-side bytes per launch: PMC passes are separate rocprofv3 runs of this same command (tools/
        # make_profiles.sh), so the figure comes from the newest committed profiles/*_traffic.json -- FETCH_SIZE
        # doubled as MI355X_MICROARCH.md prescribes for gfx950 + WRITE_SIZE, per iteration -- x the iterations of a
        # launch; `traffic_source` says which file (null when the shape differs from the profiled one)
        traffic, traffic_src = None, None
        try:
            import glob

            tf_files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")))
            if tf_files and persistent and not lockstep and (E, S, nc) == (N_EVENTS, N_STA, N_CHAINS):
                with open(tf_files[-1]) as fh:
                    tj = json.load(fh)
                per_it = 2.0 * tj["fetch_bytes_per_iteration_raw"] + tj["write_bytes_per_iteration"]
                traffic = per_it * n_timed / n_launch
                traffic_src = os.path.relpath(tf_files[-1], ROOT) + " (separate PMC passes of this command, not this run)"
        except Exception:
            traffic = None
        nch = 1 if S <= 64 else 2 if S <= 128 else 4 if S <= 256 else 0
        try:
            loops = cs.master_stats()        # which main loop the library selected for this chain set
        except Exception:
            loops = {}
        if lockstep:
            lk = loops.get("lockstep_loop", 4)
            kname = (f"k_mcmc<{nch}, *, {lk}> (persistent lock-step rank: " +
                     ("free-running chain master, per-chain swap records and per-rank headers written into the peers' inboxes from inside the kernel, "
                      "only the pair's two chains wait" if lk == 4 else "one iteration per hand-shake, swap records exchanged inside the kernel") + ")")
        elif persistent and loops.get("single_rank_loop", 3) == 7:
            kname = (f"k_mcmc<{nch}, *, 7> (free-running chain masters, one workgroup per eight chains, + resident full-evaluation workers: "
                     "one persistent launch)")
        elif persistent and os.environ.get("HTM_FLOW", "1") != "0":
            kname = (f"k_mcmc<{nch}, *, 3> (free-running chain master + resident full-evaluation workers: propose + partial/full "
                     "log-likelihood + judge + swap, one persistent launch)")
        elif persistent:
            kname = f"k_mcmc<{nch}, *, 0> (chain master with barriers + resident workers, persistent)"
        else:
            kname = f"k_step<{nch}> + k_full<{nch},false> (graph of the two-kernel path)"
        out["roofline"] = {
            "bound": "hbm", "kernel": kname,
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic, "traffic_source": traffic_src,
            "bytes_per_launch": bytes_region / n_launch, "avg_launch_us": st["device_us"] / n_launch,
            "launches": n_launch, "full_evals": st["full_evals"], "partial_evals": st["partial_evals"],
            "device_us_region": st["device_us"],
            "note": f"algorithmic bytes = {b_full} B per full evaluation + {b_part} B per single-event partial update "
                    "(SURVEY 8d) x the evaluations counted on the device in the timed region (rank 0's GPU), / the HIP-event "
                    "time of that region on the kernels' stream.  At a few chains per GPU an iteration is a dependent chain "
                    "of a few us whose data (observations, chain state) stays in L2 / Infinity Cache: the workload is bound by "
                    "instruction issue and latency on the master's CU, not by HBM (DESIGN.md 5); see roofline_batch64 for the "
                    "full-evaluation kernel on its own",
        }
    if rank == 0 and world == 1 and not args.force_lockstep and not test_engine:
        # ---- the two stages timed separately (fallback two-kernel path, same arithmetic), HIP events per launch
        n_prof = 2000
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
    # The process group is taken down BEFORE the CPU baseline: the other ranks leave (no collective waits on a watchdog
    # while rank 0 runs mpiexec for a minute, no idle rank processes competing with the baseline's MPI ranks for cores);
    # the measurement is in `out` by now.
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0 and not args.no_cpu_baseline and not test_engine:
        try:
            out["cpu_baseline"] = cpu_baseline(params, data, world * nc)
        except Exception as exc:           # the baseline is a side figure: never lose the measurement over it
            out["cpu_baseline"] = {"value": None, "unit": "proposal steps/s", "cores": 0, "kind": "port",
                                   "sample": f"not measured: {exc}"}
    if rank == 0:
        os.write(real_stdout, (json.dumps(out) + "\n").encode())


if __name__ == "__main__":
    main()
