"""Step-6 order statistics: oracle restatement pinned on the reference's own .stat files (CPU),
device radix select against the restatement (GPU), end-to-end files from a GPU step-5 run (GPU)."""
import os

import numpy as np
import pytest

from tests.helpers import load_case

STAT_CASES = ["c1", "c2", "missing", "timeonly", "fixedcorr"]


def _samples_from_fixture(fx, n_procs):
    cat = lambda nm: np.concatenate([fx[f"{nm}_{r}"] for r in range(n_procs)], axis=0)
    return {nm: cat(nm) for nm in ("vs", "qs", "t_corr", "a_corr", "hypo")}


def _names(n_sta):
    return ["S%03d" % (k + 1) for k in range(n_sta)]


@pytest.mark.parametrize("name", STAT_CASES)
def test_oracle_reproduces_reference_stat_files(name):
    """the reference's step 6 (compiled unmodified) wrote these files from the same sample records"""
    from oracle import stats_oracle as so

    fx, data, params = load_case(name)
    n_procs = int(params["n_procs"])
    s = _samples_from_fixture(fx, n_procs)
    assert len(s["vs"]) == int(fx["stat_n_mod"])
    assert so.uniform_structure_text(s["vs"], s["qs"]) == str(fx["stat_uniform_structure_stat"])
    assert so.station_corrections_text(_names(data.n_sta), s["t_corr"], s["a_corr"]) == str(fx["stat_station_corrections_stat"])
    win_id = list(range(1, data.n_events + 1))
    assert so.hypo_text(win_id, s["hypo"]) == str(fx["stat_hypo_stat"])


def test_rank_rule_is_single_precision():
    from hypotremormcmc_amd import statistics as st
    from oracle import stats_oracle as so

    for n in (40, 41, 79, 80, 200, 225, 1000, 4000, 39999, 40000, 123457):
        assert st.ranks(n) == so.ranks(n)
    assert st.ranks(40) == (1, 20, 39) and st.ranks(200) == (5, 100, 195)
    assert st.expected_n_mod(20000, 10000, 2, 1, 100) == 200


def test_remove_double_counts_rule():
    """two consecutive windows whose medians both lie inside the overlap of their 95 % boxes collapse into the
    first; a gap in the ids or a median outside the overlap keeps both (src/cls_statistics.f90:150-185)"""
    from hypotremormcmc_amd.statistics import remove_double_counts

    box = lambda m, w: [m, m - w, m + w] * 3
    q = [box(1.0, 1.0), box(1.2, 1.0), box(1.1, 1.0), box(9.0, 1.0), box(9.1, 1.0)]
    assert remove_double_counts([1, 2, 3, 4, 5], q) == [0, 3]          # 2, 3 fold into 1; 5 folds into 4
    assert remove_double_counts([1, 2, 3, 5, 6], q) == [0, 3]
    assert remove_double_counts([1, 3, 5, 7, 9], q) == [0, 1, 2, 3, 4]  # ids not consecutive
    q2 = [box(1.0, 0.05), box(1.2, 0.05)]                               # medians outside the overlap
    assert remove_double_counts([1, 2], q2) == [0, 1]


@pytest.mark.gpu
@pytest.mark.parametrize("slabs", [None, "1", "7", "64"])
def test_device_select_equals_sorted_column(slabs, monkeypatch):
    """None: the library's own choice (one launch per column group for these sizes, row slabs + one launch per digit for
    the last, 4 M-sample one); 1 / 7 / 64: the slab path forced with that many slabs (ragged last slab, slabs of a
    few rows, more slabs than 256-row pieces)"""
    from hypotremormcmc_amd import statistics as st
    from oracle import stats_oracle as so

    if slabs is not None:
        monkeypatch.setenv("HTM_SELECT_SLABS", slabs)
    rng = np.random.default_rng(11)
    for n_mod, n_par in ((40, 1), (200, 3), (1000, 130), (4097, 67), (20000, 257)):
        x = rng.normal(size=(n_mod, n_par)) * rng.choice([1e-3, 1.0, 1e6], size=n_par)
        x[:, 0] = np.round(x[:, 0])                 # many duplicates
        if n_par > 2:
            x[:, 1] = -np.abs(x[:, 1])              # all negative
            x[:, 2] = 0.0                           # constant column
        got = st.quantiles(x)
        assert np.array_equal(got, so.quantiles(x)), (n_mod, n_par)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["c1", "fixedcorr"])
def test_gpu_step5_plus_step6_writes_the_reference_stat_files(name, tmp_path):
    """step 5 on the GPU (all ranks of the job on one device), then step 6 on the GPU: the four .stat files
    against what the reference's step 5 + step 6 wrote"""
    from hypotremormcmc_amd import driver, statistics as st
    from hypotremormcmc_amd.obs_data import ObsData
    from hypotremormcmc_amd.parallel import LocalWorld

    fx, data, params = load_case(name)
    n_procs = int(params["n_procs"])
    obs = ObsData.from_arrays(data.sta_x, data.sta_y, data.t_obs, data.t_stdv, data.a_obs, data.a_stdv)
    fwd, sets = None, []
    for r in range(n_procs):
        fwd, cs = driver.build_rank(params, data.sta_x, data.sta_y, data.sta_z, obs, r, n_procs=n_procs, fwd=fwd)
        sets.append(cs)
    LocalWorld(sets).run(int(params["n_iter"]))
    smp = [cs.samples() for cs in sets]
    cat = lambda k: np.concatenate([s[k] for s in smp], axis=0)
    S = st.Statistics(n_procs, params["n_iter"], params["n_burn"], params["n_interval"], params["n_cool"],
                      _names(data.n_sta), range(1, data.n_events + 1))
    S.estimate_vs_qs(cat("vs"), cat("qs"), tmp_path)
    S.estimate_corr_factors(cat("t_corr"), cat("a_corr"), tmp_path)
    S.estimate_hypo(cat("hypo"), tmp_path)
    for fn in ("uniform_structure.stat", "station_corrections.stat", "hypo.stat", "hypo.stat.removed"):
        got = open(os.path.join(tmp_path, fn)).read().split("\n")
        ref = str(fx["stat_" + fn.replace(".", "_")]).split("\n")
        assert got[0] == ref[0] and len(got) == len(ref), fn
        for g, r in zip(got[1:], ref[1:]):
            if not r:
                continue
            gv, rv = g.split(), r.split()
            assert gv[0] == rv[0]
            np.testing.assert_allclose([float(v) for v in gv[1:]], [float(v) for v in rv[1:]], atol=2e-6, rtol=0)
