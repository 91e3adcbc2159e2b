"""Throughput of the step-6 order-statistics kernel (k_select) on device-resident samples."""
import ctypes as C
import sys
import time

sys.path.insert(0, ".")
import numpy as np
import torch

from hypotremormcmc_amd import _lib, statistics as st

lib = _lib.load()
for n_mod, n_par in ((4000, 3130), (40000, 3130), (80000, 3130)):
    x = torch.randn(n_mod, n_par, dtype=torch.float64, device="cuda")
    out = torch.empty(n_par, 3, dtype=torch.float64, device="cuda")
    rk = (C.c_int * 3)(*st.ranks(n_mod))
    s = torch.cuda.current_stream().cuda_stream
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        _lib.check(lib.htm_quantiles_dev(0, x.data_ptr(), n_mod, n_par, n_par, rk, out.data_ptr(), s))
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    ref = torch.sort(x, dim=0).values[[r - 1 for r in st.ranks(n_mod)]].T.contiguous()
    ok = bool(torch.equal(ref, out))
    b = n_mod * n_par * 8
    print(f"n_mod {n_mod} x n_par {n_par}: {1e3 * dt:.2f} ms, {b / dt / 1e9:.1f} GB/s of samples "
          f"({16 * b / dt / 1e9:.0f} GB/s read over 16 passes), equal to torch.sort: {ok}", flush=True)
