"""CPU: the C-ABI library builds, loads and exports every symbol include/htm_hip.h declares; without a GPU
its entry points fail loudly (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _declared():
    src = open(os.path.join(ROOT, "include", "htm_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(htm_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from hypotremormcmc_amd import _lib

    lib = _lib.load()
    names = _declared()
    assert len(names) >= 35
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/htm_hip.h but not exported"
    assert set(names) == set(_lib.SIGNATURES), set(names) ^ set(_lib.SIGNATURES)
    assert lib.htm_abi_version() == 1


def test_no_cpu_fallback_without_device():
    from hypotremormcmc_amd import _lib

    lib = _lib.load()
    n = C.c_int(-1)
    rc = lib.htm_device_count(C.byref(n))
    if rc == 0 and n.value > 0:
        pytest.skip("a GPU is present; the no-device behaviour is exercised on the CPU-only container")
    from hypotremormcmc_amd.forward import Forward
    from hypotremormcmc_amd.obs_data import ObsData

    z = np.zeros((2, 3))
    obs = ObsData.from_arrays(np.zeros(3), np.zeros(3), z, z + 1, z, z + 1)
    with pytest.raises(_lib.HtmError, match="no HIP device|CPU fallback"):
        Forward(n_sta=3, n_events=2, sta_x=np.zeros(3), sta_y=np.zeros(3), sta_z=np.zeros(3), obs=obs)
    assert lib.htm_selftest(0) == -2       # HTM_ENODEVICE


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "hypotremormcmc_amd")
    for dp_, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".f90", ".cpp")) or f == "Makefile":
                txt = open(os.path.join(dp_, f), errors="ignore").read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f
                assert "liboracle" not in txt and "htm_oracle" not in txt, f
