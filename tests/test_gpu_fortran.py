"""GPU parity of the Fortran host layer (hypotremormcmc_amd/fortran): the drop-in `module cls_forward`
shim and the Fortran step-5 driver, both through ISO_C_BINDING onto the same C ABI.  Expected values are
the reference's own outputs (golden fixtures)."""
import os
import subprocess

import numpy as np
import pytest

from hypotremormcmc_amd import synth
from tests.helpers import load_case, tf

pytestmark = pytest.mark.gpu

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
BUILD = os.path.join(ROOT, "hypotremormcmc_amd", "fortran", "build")


def _need(binary):
    path = os.path.join(BUILD, binary)
    if not os.path.exists(path):
        pytest.fail(f"{path} missing: run `make -C hypotremormcmc_amd/fortran` (build() does)")
    return path


@pytest.mark.parametrize("name", ["c1", "missing", "timeonly"])
def test_fortran_forward_shim_known_answers(name, tmp_path):
    fx, data, params = load_case(name)
    synth.write_dataset(str(tmp_path), data)
    E, S = data.n_events, data.n_sta
    n_cases = len(fx["probe_L"])
    fl = lambda v: "T" if tf(v) else "F"
    with open(tmp_path / "probe_in.txt", "w") as f:
        f.write(f"{S} {E} {fl(params.get('use_time', 'T'))} {fl(params.get('use_amp', 'T'))} {n_cases}\n")
        for arr in (data.sta_x, data.sta_y, data.sta_z):
            f.write(" ".join("%.17g" % v for v in arr) + "\n")
        for k in range(n_cases):
            f.write(" ".join("%.17g" % v for v in fx["probe_in_hypo"][k]) + "\n")
            f.write(" ".join("%.17g" % v for v in fx["probe_in_t_corr"][k]) + "\n")
            f.write("%.17g\n" % fx["probe_in_vs"][k])
            f.write(" ".join("%.17g" % v for v in fx["probe_in_a_corr"][k]) + "\n")
            f.write("%.17g\n" % fx["probe_in_qs"][k])
            f.write("%d\n" % int(fx["probe_in_evt_id"][k]))
            f.write(" ".join("%.17g" % v for v in fx["probe_in_xyz"][k]) + "\n")
    subprocess.run([_need("forward_probe")], cwd=tmp_path, check=True, timeout=300)
    tok = [float(t) for t in open(tmp_path / "probe_out_hip.txt").read().split()]
    L0 = np.array(tok[:3]); single = np.array(tok[3:3 + 2 * S]).reshape(S, 2); rest = np.array(tok[3 + 2 * S:]).reshape(-1, 3)
    got = np.vstack([L0, rest])
    np.testing.assert_allclose(got, fx["probe_L"], rtol=1e-12)
    np.testing.assert_allclose(single[:, 0], fx["probe_t_syn_single"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(single[:, 1], fx["probe_a_syn_single"], rtol=0, atol=1e-12)


def _records(path, n_val, bo="<"):
    dt = np.dtype([("iter", bo + "i4"), ("val", bo + "f8", (n_val,))])
    if os.path.getsize(path) == 0:
        return np.zeros(0, np.int32), np.zeros((0, n_val))
    a = np.fromfile(path, dtype=dt)
    return a["iter"].astype(np.int32), a["val"].reshape(-1, n_val).astype(np.float64)


@pytest.mark.parametrize("name,endian", [("c2", "little"), ("missing", "little"), ("c2", "big")])
def test_fortran_driver_reproduces_reference_outputs(name, endian, tmp_path):
    """same CLI, parameter file, inputs and output files as the reference's hypo_tremor_mcmc; endian = big: the
    records of a stock-gfortran build of the reference (-fconvert=big-endian, src/Makefile:7-9), HTM_SAMPLE_ENDIAN"""
    fx, data, params = load_case(name)
    synth.write_dataset(str(tmp_path), data)
    synth.write_param_file(str(tmp_path / "run.in"), **{k: v for k, v in params.items()})
    subprocess.run([_need("hypo_tremor_mcmc_hip"), "run.in"], cwd=tmp_path, check=True, timeout=600,
                   stdout=subprocess.DEVNULL, env=dict(os.environ, HTM_SAMPLE_ENDIAN=endian))
    E, S = data.n_events, data.n_sta
    bo = ">" if endian == "big" else "<"
    it, v = _records(tmp_path / "likelihood00.out", 1, bo)
    assert np.array_equal(it, fx["lik_iter_0"])
    np.testing.assert_allclose(v[:, 0], fx["lik_0"], rtol=1e-9, atol=0)
    for nm, nv in (("vs", 1), ("qs", 1), ("t_corr", S), ("a_corr", S), ("hypo", 3 * E)):
        it, v = _records(tmp_path / f"{nm}.00.out", nv, bo)
        assert np.array_equal(it, fx[f"{nm}_iter_0"])
        np.testing.assert_allclose(v, fx[f"{nm}_0"], rtol=1e-11, atol=1e-12)
    rows = [ln.split('"') for ln in open(tmp_path / "proposal_count.txt")]
    assert [r[1] for r in rows] == fx["count_labels"].tolist()
    assert [int(r[2].split()[0]) for r in rows] == fx["n_propose"].tolist()
    assert [int(r[2].split()[1]) for r in rows] == fx["n_accept"].tolist()


def test_unmodified_reference_driver_on_hip_forward_shim(tmp_path):
    """The drop-in check proper: oracle/_ref/hypo_tremor_mcmc_ref_hipfwd is the reference's own step-5 driver
    and modules, compiled UNMODIFIED in the build container, with only src/cls_forward.f90 replaced by the
    build's shim (hypotremormcmc_amd/fortran/cls_forward_hip.f90 -> libhtm_hip.so).  Run under real MPI
    (2 ranks sharing the GPU) it must reproduce the traces the pure reference produced (fixture c1)."""
    exe = os.path.join(ROOT, "oracle", "_ref", "hypo_tremor_mcmc_ref_hipfwd")
    mpiexec = "/opt/conda/bin/mpiexec"
    if not (os.path.exists(exe) and os.path.exists(mpiexec)):
        pytest.skip("reference+shim binary or MPICH not present on this box")
    fx, data, params = load_case("c1")
    params = dict(params, n_iter="4000", n_burn="2000")
    synth.write_dataset(str(tmp_path), data)
    synth.write_param_file(str(tmp_path / "run.in"), **params)
    r = subprocess.run([mpiexec, "-np", "2", exe, "run.in"], cwd=tmp_path, timeout=900, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    for rank in range(2):
        it, v = _records(tmp_path / ("likelihood%02d.out" % rank), 1)
        n = len(it)
        assert n > 0 and np.array_equal(it, fx[f"lik_iter_{rank}"][:n])
        np.testing.assert_allclose(v[:, 0], fx[f"lik_{rank}"][:n], rtol=1e-9, atol=0)


@pytest.mark.parametrize("transport", ["direct", "staged"])
@pytest.mark.parametrize("name", ["c1", "timeonly", "fixedcorr", "rejects"])
def test_fortran_mpi_driver_reproduces_reference_outputs(name, transport, tmp_path):
    """hypo_tremor_mcmc_hip_mpi under real MPI (2-3 processes sharing the GPU): an MPI program like the reference,
    every rank's chains device-resident.  "direct": persistent lock-step, the kernels exchange the swap records
    through IPC-mapped inboxes (handles all-gathered over MPI once); "staged": one MPI_Allgather of host-staged
    records per iteration.  All output files of all ranks against what the reference wrote under the same mpiexec."""
    mpiexec = "/opt/conda/bin/mpiexec"
    exe = os.path.join(BUILD, "hypo_tremor_mcmc_hip_mpi")
    if not (os.path.exists(exe) and os.path.exists(mpiexec)):
        pytest.skip("MPI driver or MPICH not present on this box")
    fx, data, params = load_case(name)
    n_procs = int(params["n_procs"])
    synth.write_dataset(str(tmp_path), data)
    synth.write_param_file(str(tmp_path / "run.in"), **{k: v for k, v in params.items()})
    env = dict(os.environ, HTM_XCHG="1" if transport == "direct" else "0")
    r = subprocess.run([mpiexec, "-np", str(n_procs), exe, "run.in"], cwd=tmp_path, timeout=300, capture_output=True,
                       text=True, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    assert ("swap exchange: in-kernel" in r.stdout) == (transport == "direct"), r.stdout[-800:]
    E, S = data.n_events, data.n_sta
    for rank in range(n_procs):
        it, v = _records(tmp_path / ("likelihood%02d.out" % rank), 1)
        assert np.array_equal(it, fx[f"lik_iter_{rank}"])
        np.testing.assert_allclose(v[:, 0], fx[f"lik_{rank}"], rtol=1e-9, atol=0)
        for nm, nv in (("vs", 1), ("qs", 1), ("t_corr", S), ("a_corr", S), ("hypo", 3 * E)):
            it, v = _records(tmp_path / ("%s.%02d.out" % (nm, rank)), nv)
            assert np.array_equal(it, fx[f"{nm}_iter_{rank}"])
            np.testing.assert_allclose(v, fx[f"{nm}_{rank}"], rtol=1e-11, atol=1e-12)
    rows = [ln.split('"') for ln in open(tmp_path / "proposal_count.txt")]
    assert [int(r_[2].split()[0]) for r_ in rows] == fx["n_propose"].tolist()
    assert [int(r_[2].split()[1]) for r_ in rows] == fx["n_accept"].tolist()


def test_fortran_mpi_driver_over_rccl(tmp_path):
    """HTM_XCHG=rccl: the Fortran MPI program reaches RCCL through the C ABI (htm_comm_*: unique id from rank 0 over
    MPI_Bcast, ncclCommInitRank, one ncclAllGather per iteration enqueued from C).  One rank here -- RCCL refuses two
    ranks on one device, and this box has one GPU; the multi-rank protocol itself is covered by the transports above."""
    mpiexec = "/opt/conda/bin/mpiexec"
    exe = os.path.join(BUILD, "hypo_tremor_mcmc_hip_mpi")
    if not (os.path.exists(exe) and os.path.exists(mpiexec)):
        pytest.skip("MPI driver or MPICH not present on this box")
    fx, data, params = load_case("c2")
    synth.write_dataset(str(tmp_path), data)
    synth.write_param_file(str(tmp_path / "run.in"), **{k: v for k, v in params.items()})
    r = subprocess.run([mpiexec, "-np", "1", exe, "run.in"], cwd=tmp_path, timeout=300, capture_output=True, text=True,
                       env=dict(os.environ, HTM_XCHG="rccl"))
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    assert "swap exchange: RCCL" in r.stdout, r.stdout[-800:]
    it, v = _records(tmp_path / "likelihood00.out", 1)
    assert np.array_equal(it, fx["lik_iter_0"])
    np.testing.assert_allclose(v[:, 0], fx["lik_0"], rtol=1e-9, atol=0)
    rows = [ln.split('"') for ln in open(tmp_path / "proposal_count.txt")]
    assert [int(r_[2].split()[0]) for r_ in rows] == fx["n_propose"].tolist()
    assert [int(r_[2].split()[1]) for r_ in rows] == fx["n_accept"].tolist()


@pytest.mark.parametrize("name,endian", [("c1", "little"), ("timeonly", "little"), ("fixedcorr", "little"), ("c1", "big")])
def test_fortran_statistics_program_writes_the_reference_stat_files(name, endian, tmp_path):
    """hypo_tremor_statistics_hip on the sample files the reference's step 5 wrote (rebuilt from the fixture):
    the four .stat files must be the reference step 6's, character for character.  endian = big: sample files as a
    stock-gfortran build of the reference's step 5 writes them (HTM_SAMPLE_ENDIAN=big)."""
    bo = ">" if endian == "big" else "<"
    fx, data, params = load_case(name)
    n_procs = int(params["n_procs"])
    synth.write_dataset(str(tmp_path), data)
    synth.write_param_file(str(tmp_path / "run.in"), **{k: v for k, v in params.items()})
    E, S = data.n_events, data.n_sta
    for r in range(n_procs):
        for nm, nv in (("vs", 1), ("qs", 1), ("t_corr", S), ("a_corr", S), ("hypo", 3 * E)):
            it, v = fx[f"{nm}_iter_{r}"], fx[f"{nm}_{r}"].reshape(len(fx[f"{nm}_iter_{r}"]), nv)
            rec = np.zeros(len(it), dtype=np.dtype([("iter", bo + "i4"), ("val", bo + "f8", (nv,))]))
            rec["iter"] = it; rec["val"] = v
            rec.tofile(tmp_path / ("%s.%02d.out" % (nm, r)))
    res = subprocess.run([_need("hypo_tremor_statistics_hip"), "run.in"], cwd=tmp_path, timeout=600, capture_output=True,
                         text=True, env=dict(os.environ, HTM_SAMPLE_ENDIAN=endian))
    assert res.returncode == 0, (res.stdout[-1500:], res.stderr[-1500:])
    for fn in ("uniform_structure.stat", "station_corrections.stat", "hypo.stat", "hypo.stat.removed"):
        assert open(tmp_path / fn).read() == str(fx["stat_" + fn.replace(".", "_")]), fn
