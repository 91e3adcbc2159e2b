#!/usr/bin/env python3
"""Run ON THE GPU BOX: wall time of the main loop of the Fortran product programs (hypo_tremor_mcmc_hip and
hypo_tremor_mcmc_hip_mpi with one rank) on BASELINE configs[2] -- 1 000 events x 64 stations, 8 chains -- for
N iterations (default 1 000 000), main loop only (reference src/hypo_tremor_mcmc.f90:236-284): the programs print it
themselves with HTM_TIME_MAIN_LOOP=1.  Slices of 1 000 iterations (the reference's progress cadence,
src/cls_mcmc.f90:230: one library call + one progress report each) against slices of 20 000.

    python tools/time_fortran.py [n_iter] > gpurun_out/fortran_timing.txt
"""
import os
import re
import subprocess
import sys
import tempfile
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from hypotremormcmc_amd import synth  # noqa: E402

n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
B = os.path.join(ROOT, "hypotremormcmc_amd", "fortran", "build")
data = synth.make_synthetic(1000, 64, 1)
with tempfile.TemporaryDirectory(prefix="htm_ftime_") as work:
    synth.write_dataset(work, data)
    synth.write_param_file(os.path.join(work, "run.in"), n_procs=1, n_chains=8, n_cool=1, n_iter=n_iter, n_burn=n_iter,
                           n_interval=1000)
    runs = [("hypo_tremor_mcmc_hip", [os.path.join(B, "hypo_tremor_mcmc_hip"), "run.in"], {}),
            ("hypo_tremor_mcmc_hip_mpi, 1 rank, in-kernel exchange",
             ["/opt/conda/bin/mpiexec", "-np", "1", os.path.join(B, "hypo_tremor_mcmc_hip_mpi"), "run.in"], {})]
    print(f"# Fortran product programs, configs[2] (1000 events x 64 stations, 8 chains), {n_iter} iterations, main loop only")
    for name, cmd, extra in runs:
        if not os.path.exists(cmd[0]) or not os.path.exists(cmd[-2] if len(cmd) > 2 else cmd[0]):
            print(f"{name}: not built / no mpiexec")
            continue
        for slice_ in (1000, 20000):
            env = dict(os.environ, HTM_TIME_MAIN_LOOP="1", HTM_SLICE=str(slice_), **extra)
            t0 = time.time()
            r = subprocess.run(cmd, cwd=work, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, timeout=900)
            m = re.search(r"main loop:\s*([0-9.]+) s.*=\s*([0-9.]+) proposal steps/s", r.stderr)
            if r.returncode != 0 or not m:
                print(f"{name}, slices of {slice_}: FAILED rc {r.returncode}: {r.stderr[-400:]}")
                continue
            print(f"{name}, slices of {slice_:5d}: main loop {float(m.group(1)):9.4f} s = {float(m.group(2)):12.0f} proposal steps/s "
                  f"({1e6 * float(m.group(1)) / n_iter:.3f} us/iteration; whole program {time.time() - t0:.1f} s)")
            sys.stdout.flush()
