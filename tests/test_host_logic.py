"""CPU: host-side mirrors of the reference interfaces -- RNG, model set-up order, parameter-file grammar,
opt_data reader, output writers -- against the oracle / golden fixtures."""
import os
import struct

import numpy as np
import pytest

from hypotremormcmc_amd import driver, synth
from hypotremormcmc_amd.mod_random import Xorshift128
from hypotremormcmc_amd.obs_data import ObsData
from hypotremormcmc_amd.param import Param, ParamError, parse_line
from tests.helpers import load_case


def test_host_rng_matches_reference_vectors():
    fx, _, _ = load_case("c1")
    for r in range(4):
        g = Xorshift128(r)
        v = [g.rand_u() for _ in range(8)] + [g.rand_u2(), g.rand_g(), g.rand_r(), g.rand_g()]
        assert v == fx["probe_rng"][r].tolist()


@pytest.mark.parametrize("name", ["c1", "fixedcorr", "timeonly"])
def test_initial_models_match_oracle_setup(name):
    """same draws in the same order as src/hypo_tremor_mcmc.f90:120-211 => identical initial state + RNG"""
    from oracle import oracle

    fx, data, params = load_case(name)
    job = oracle.Job(params, data)
    obs = ObsData.from_arrays(data.sta_x, data.sta_y, data.t_obs, data.t_stdv, data.a_obs, data.a_stdv)
    x_mu, y_mu = obs.make_initial_guess()
    for rank in range(int(params["n_procs"])):
        models, temps, rng = driver.build_initial_models(params, data.n_sta, data.n_events, x_mu, y_mu, rank)
        assert rng.state == job.rng_state(rank)
        for c, m in enumerate(models):
            st = job.chain(rank, c)
            assert np.array_equal(m["hypo"].x, st["hypo"])
            assert np.array_equal(m["t_corr"].x, st["t_corr"]) and np.array_equal(m["a_corr"].x, st["a_corr"])
            assert m["vs"].x[0] == st["vs"] and m["qs"].x[0] == st["qs"] and temps[c] == st["temp"]
            mu, sg, stp, pt = job.hypo_priors(rank, c)
            assert np.array_equal(m["hypo"].mu, mu) and np.array_equal(m["hypo"].sigma, sg)
            assert np.array_equal(m["hypo"].step_size, stp) and np.array_equal(m["hypo"].prior_type, pt)


def test_param_grammar(tmp_path):
    assert parse_line("n_iter = 4000000   # comment") == ("n_iter", "4000000")
    assert parse_line("filename_format = $STA + / + $ID + . + $CMP ") == ("filename_format", "$STA+/+$ID+.+$CMP")
    assert parse_line("# only a comment") is None and parse_line("novalue =") is None and parse_line("= 3") is None
    d = synth.make_synthetic(3, 4, 0)
    synth.write_dataset(str(tmp_path), d)
    synth.write_param_file(str(tmp_path / "p.in"), temp_high="200.d0", solve_vs=".true.", n_chains=5)
    p = Param(str(tmp_path / "p.in"))
    assert p.get_temp_high() == 200.0 and p.get_solve_vs() is True and p.get_n_chains() == 5
    assert p.get_prior_t_corr() == 0.0 and p.get_n_stations() == 4            # defaults, station file
    assert np.array_equal(p.get_sta_x(), d.sta_x)
    with open(tmp_path / "bad.in", "w") as f:
        f.write(open(tmp_path / "p.in").read() + "no_such_key = 1\n")
    with pytest.raises(ParamError, match="Invalid parameter name"):
        Param(str(tmp_path / "bad.in"))
    with open(tmp_path / "short.in", "w") as f:
        f.write("station_file = station_xy.list\nn_procs = 1\n")
    with pytest.raises(ParamError, match="is not given"):
        Param(str(tmp_path / "short.in"))


def test_reference_sample_parameter_file_is_accepted(tmp_path):
    """the reference's own sample file parses unchanged (only the station file it names must exist)"""
    ref = "/root/reference/sample/hypo_tremor.in"
    if not os.path.exists(ref):
        pytest.skip("reference tree not present (GPU box)")
    d = synth.make_synthetic(2, 3, 0)
    synth.write_dataset(str(tmp_path), d)
    dst = tmp_path / "hypo_tremor.in"
    dst.write_text(open(ref).read())
    p = Param(str(dst))
    assert p.get_n_procs() == 20 and p.get_n_chains() == 5 and p.get_n_iter() == 4000000
    assert p.get_temp_high() == 200.0 and p.get_prior_qs() == 250.0 and p.get_use_amp() is True


def test_obs_reader_roundtrip(tmp_path):
    d = synth.make_synthetic(6, 5, 9, n_missing=2)
    synth.write_dataset(str(tmp_path), d)
    ids = driver.read_selected_win(str(tmp_path / "selected_win.dat"))
    assert ids == list(range(1, 7))
    obs = ObsData(ids, 5, d.sta_x, d.sta_y, directory=str(tmp_path))
    assert np.array_equal(obs.t_obs, d.t_obs) and np.array_equal(obs.t_stdv, d.t_stdv)
    assert np.array_equal(obs.a_obs, d.a_obs) and np.array_equal(obs.a_stdv, d.a_stdv)
    with pytest.raises(SystemExit):
        ObsData([99], 5, d.sta_x, d.sta_y, directory=str(tmp_path))


def test_proposal_count_format(tmp_path):
    p = tmp_path / "proposal_count.txt"
    driver.write_proposal_count(str(p), [984, 985, 1001, 1021, 12041, 12124, 11844], [50, 736, 599, 906, 9326, 1358, 1852])
    lines = p.read_text().splitlines()
    assert lines[0] == '"vs   "       984        50' and lines[1] == '"t_cor"       985       736'
    assert lines[4] == '"x    "     12041      9326'


def test_fortran_host_layer_builds_and_fails_loudly_without_gpu(tmp_path):
    """the Fortran shim + driver link against the C ABI; without a device they stop with the library's message"""
    import ctypes as C
    import subprocess

    from hypotremormcmc_amd import _lib

    fdir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "hypotremormcmc_amd", "fortran")
    subprocess.run(["make", "-C", fdir], check=True, stdout=subprocess.DEVNULL)
    exe = os.path.join(fdir, "build", "hypo_tremor_mcmc_hip")
    assert os.path.exists(exe) and os.path.exists(os.path.join(fdir, "build", "forward_probe"))
    n = C.c_int(0)
    if _lib.load().htm_device_count(C.byref(n)) == 0 and n.value > 0:
        pytest.skip("a GPU is present")
    d = synth.make_synthetic(3, 4, 0)
    synth.write_dataset(str(tmp_path), d)
    synth.write_param_file(str(tmp_path / "p.in"), n_procs=1, n_chains=2, n_iter=10)
    r = subprocess.run([exe, "p.in"], cwd=tmp_path, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0
    assert "no HIP device" in (r.stderr + r.stdout) or "libhtm_hip" in (r.stderr + r.stdout)


def test_rng_jump_ahead_equals_serial_stream():
    """htm_rng_jump (host GF(2) powers of the xorshift128 step, the tables the device generator k_rawgen starts its
    segments from) against the serial generator of src/mod_random.f90:60-74, for jump lengths around the table's
    block sizes (64 * 2^b) and for all three golden seeds of SURVEY 8a."""
    import ctypes as C

    from hypotremormcmc_amd import _lib
    from hypotremormcmc_amd.mod_random import Xorshift128

    lib = _lib.load()
    for rank in (0, 1, 2):
        g = Xorshift128(rank)
        s0 = (C.c_uint32 * 4)(*g.state)
        done = 0
        for n in (0, 1, 2, 63, 64, 65, 127, 128, 64 * 37 + 5, 4095, 4096, 4097, 1 << 16, (1 << 18) + 64 * 3):
            while done < n:
                g._next(); done += 1
            out = (C.c_uint32 * 4)()
            _lib.check(lib.htm_rng_jump(s0, n, out))
            assert tuple(out) == g.state, (rank, n)
    # composition: jump(a) then jump(b) == jump(a + b), far beyond what a serial loop could check
    a, b = 10 ** 15 + 7, 3 * 10 ** 12 + 11
    s0 = (C.c_uint32 * 4)(*Xorshift128(0).state)
    sa, sab, sd = (C.c_uint32 * 4)(), (C.c_uint32 * 4)(), (C.c_uint32 * 4)()
    _lib.check(lib.htm_rng_jump(s0, a, sa)); _lib.check(lib.htm_rng_jump(sa, b, sab)); _lib.check(lib.htm_rng_jump(s0, a + b, sd))
    assert tuple(sab) == tuple(sd)


def test_sample_files_in_either_byte_order(tmp_path, monkeypatch):
    """HTM_SAMPLE_ENDIAN: records [int32 iteration][n float64] (src/hypo_tremor_mcmc.f90:216-233) as a stock-gfortran
    build of the reference writes them (-fconvert=big-endian, src/Makefile:7-9) or native"""
    from hypotremormcmc_amd.statistics import read_sample_file

    class FakeChains:                       # what write_outputs reads from a ChainSet
        def likelihood_trace(self):
            return np.array([1, 2, 3]), None, np.array([-1.5, -2.25, 1e300])

        def samples(self):
            return {"iter": np.array([10, 20]), "vs": np.array([3.5, 3.25]), "qs": np.array([100.0, 101.0]),
                    "t_corr": np.arange(8.0).reshape(2, 4), "a_corr": -np.arange(8.0).reshape(2, 4),
                    "hypo": np.linspace(0, 1, 12).reshape(2, 6)}

    for endian, bo in (("little", "<"), ("big", ">")):
        d = tmp_path / endian
        d.mkdir()
        monkeypatch.setenv("HTM_SAMPLE_ENDIAN", endian)
        driver.write_outputs(str(d), 0, FakeChains())
        raw = open(d / "likelihood00.out", "rb").read()
        assert raw[:12] == struct.pack(bo + "id", 1, -1.5) and len(raw) == 36
        it, v = read_sample_file(str(d / "t_corr.00.out"), 4)
        assert it.tolist() == [10, 20] and np.array_equal(v, np.arange(8.0).reshape(2, 4))
        it, v = read_sample_file(str(d / "hypo.00.out"), 6)
        assert np.array_equal(v, np.linspace(0, 1, 12).reshape(2, 6))
    # the two orders differ on disk, and reading with the wrong one does not silently give the same numbers
    assert open(tmp_path / "big" / "vs.00.out", "rb").read() != open(tmp_path / "little" / "vs.00.out", "rb").read()
    assert read_sample_file(str(tmp_path / "little" / "vs.00.out"), 1, endian="big")[0].tolist() != [10, 20]
    monkeypatch.setenv("HTM_SAMPLE_ENDIAN", "middle")
    with pytest.raises(ValueError):
        driver.sample_byte_order()
