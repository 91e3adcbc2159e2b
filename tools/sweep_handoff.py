"""Sweep the hand-off geometry knobs of k_mcmc (run on the GPU box): prints steps/s per configuration.
    python tools/sweep_handoff.py [replicas,slot_stride,pgran_stride,npoll ...]      (SWEEP_BENCH_ARGS: the shape)"""
import itertools
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
configs = []
for rep, stride, pgs, npoll in [
    (1, 1024, 16, 1), (4, 1024, 16, 1), (4, 4096, 16, 1), (8, 4096, 16, 1), (8, 65536, 16, 1), (16, 4096, 16, 1),
    (16, 65536, 16, 1), (8, 4352, 16, 1), (8, 4096, 128, 1), (8, 4096, 256, 1), (8, 4096, 16, 2), (16, 65536, 256, 1),
    (16, 65536, 256, 2),
]:
    configs.append(dict(HTM_SLOT_REPLICAS=rep, HTM_SLOT_STRIDE=stride, HTM_PGRAN_STRIDE=pgs, HTM_NPOLL=npoll))
if len(sys.argv) > 1:
    configs = [dict(zip(("HTM_SLOT_REPLICAS", "HTM_SLOT_STRIDE", "HTM_PGRAN_STRIDE", "HTM_NPOLL"), map(int, a.split(","))))
               for a in sys.argv[1:]]
for cfg in configs:
    env = dict(os.environ, **{k: str(v) for k, v in cfg.items()})
    # (a bench step is a block of iterations: a few short blocks per configuration; SWEEP_BENCH_ARGS picks the shape, e.g.
    # "--events 10000 --stations 128 --chains 16 --forward-precision fp32 --iters-per-step 1024")
    extra = os.environ.get("SWEEP_BENCH_ARGS", "--iters-per-step 16384").split()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--steps", "4",
                          "--warmup", "1"] + extra, env=env, capture_output=True, text=True)
    try:
        v = json.loads(out.stdout.strip().splitlines()[-1])["value"]
    except Exception:
        v = float("nan")
        print(out.stderr[-500:])
    print(cfg, "%.0f" % v, flush=True)
