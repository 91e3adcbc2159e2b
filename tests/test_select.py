"""Step 4 (`hypo_tremor_select`): the regressions of src/cls_selector.f90:75-132 / src/mod_regress.f90.

CPU: the oracle restatement against what the COMPILED REFERENCE step 4 wrote (regress.dat, selected_win.dat of
tests/golden/select*.npz, produced under mpiexec by tests/golden/make_golden.py): bit-identical.
GPU: `htm_select_regress` through the C ABI against the same files (1e-10 relative: wave-tree sums instead of serial
ones), the same windows selected, and the drop-in program `python -m hypotremormcmc_amd.select` on the reference's input
files."""
import os
import subprocess
import sys

import numpy as np
import pytest

from hypotremormcmc_amd import synth

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
GOLDEN = os.path.join(ROOT, "tests", "golden")
CASES = ["select", "select_wide"]


def _load(name):
    fx = np.load(os.path.join(GOLDEN, name + ".npz"))
    E, S = (int(v) for v in fx["in_shape"])
    data = synth.make_synthetic(E, S, int(fx["in_seed"]), 0)
    pv = dict(zip(fx["param_keys"].tolist(), (float(v) for v in fx["param_vals"])))
    return fx, data, pv


@pytest.mark.parametrize("name", CASES)
def test_oracle_equals_reference_regress_file(name):
    from oracle import oracle

    fx, data, pv = _load(name)
    out = oracle.select_regress(data.sta_x, data.sta_y, data.sta_z, pv["z_guess"], data.t_obs, data.t_stdv, data.a_obs,
                                data.a_stdv)
    assert np.array_equal(fx["regress"][:, 0].astype(int), np.arange(1, data.n_events + 1))
    assert np.array_equal(out, fx["regress"][:, 1:])           # bit-exact: same operation order, same libm
    from hypotremormcmc_amd.select import select

    keep = select(out, pv["vs_min"], pv["vs_max"], pv["b_min"], pv["b_max"])
    assert np.array_equal(np.flatnonzero(keep) + 1, fx["selected"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_device_regressions_equal_reference(name):
    from hypotremormcmc_amd.select import regress, select

    fx, data, pv = _load(name)
    out = regress(data.sta_x, data.sta_y, data.sta_z, pv["z_guess"], data.t_obs, data.t_stdv, data.a_obs, data.a_stdv)
    np.testing.assert_allclose(out, fx["regress"][:, 1:], rtol=1e-10, atol=0)
    keep = select(out, pv["vs_min"], pv["vs_max"], pv["b_min"], pv["b_max"])
    assert np.array_equal(np.flatnonzero(keep) + 1, fx["selected"])


@pytest.mark.gpu
def test_select_program_writes_the_reference_files(tmp_path):
    fx, data, pv = _load("select")
    synth.write_dataset(str(tmp_path), data)
    os.rename(tmp_path / "selected_win.dat", tmp_path / "detected_win.dat")
    with open(tmp_path / "select.in", "w") as fh:
        fh.write("n_procs = 1\nstation_file = station_xy.list\n")
        for k, v in pv.items():
            fh.write("%s = %r\n" % (k, v))
    r = subprocess.run([sys.executable, "-m", "hypotremormcmc_amd.select", "select.in"], cwd=tmp_path, timeout=300,
                       capture_output=True, text=True, env=dict(os.environ, PYTHONPATH=ROOT))
    assert r.returncode == 0, r.stderr[-2000:]
    reg = np.array([float(x) for x in open(tmp_path / "regress.dat").read().split()]).reshape(-1, 7)
    np.testing.assert_allclose(reg, fx["regress"], rtol=1e-10, atol=0)
    sel = [int(ln.split()[0]) for ln in open(tmp_path / "selected_win.dat") if ln.strip()]
    assert sel == fx["selected"].tolist()


def test_select_parameter_keys_are_required(tmp_path):
    from hypotremormcmc_amd.param import Param, ParamError

    (tmp_path / "station_xy.list").write_text("S001 0.0 0.0 0.0 1.0 1.0\n")
    (tmp_path / "p.in").write_text("n_procs = 1\nstation_file = station_xy.list\nz_guess = 7.0\nvs_min = 2.0\nvs_max = 4.0\nb_min = 0.0\n")
    with pytest.raises(ParamError):
        Param(str(tmp_path / "p.in"), from_where="select")     # b_max missing (src/cls_param.f90:123-126, :294-346)
