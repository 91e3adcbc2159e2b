"""Parameter-file and station-file reader with the reference's grammar
(src/cls_line_text.f90:88-148, src/cls_param.f90:212-255,:294-346,:350-390,:429-541):

  * `#` starts a comment, ALL spaces are removed, `name=value`; lines without `=` (or with an empty side)
    are skipped;
  * an unknown name is an error; every key of the per-program required list must be present;
  * prior_t_corr / prior_a_corr default to 0 (src/cls_param.f90:89,:91).
"""
from __future__ import annotations

import os

INT_KEYS = {"n_procs", "n_pair_thred", "n_iter", "n_burn", "n_interval", "n_chains", "n_cool"}
STR_KEYS = {"station_file", "time_id_file", "cmp1", "cmp2", "data_dir", "filename_format",
            # optional key of this build, not in the reference's grammar (its files never contain it):
            # forward_precision = fp64 | fp32  (fp32 forward model with fp64 sums and accept, BASELINE configs[4])
            "forward_precision"}
BOOL_KEYS = {"solve_vs", "solve_qs", "solve_t_corr", "solve_a_corr", "use_amp", "use_time"}
REAL_KEYS = {
    "t_win_conv", "t_win_corr", "t_step_corr", "alpha", "vs_min", "vs_max", "b_min", "b_max", "z_guess",
    "temp_high", "prior_width_xy", "prior_width_z", "prior_z", "prior_vs", "prior_width_vs", "prior_qs",
    "prior_width_qs", "prior_t_corr", "prior_width_t_corr", "prior_a_corr", "prior_width_a_corr",
    "step_size_xy", "step_size_z", "step_size_vs", "step_size_t_corr", "step_size_qs", "step_size_a_corr",
}
# src/cls_param.f90:127-137
REQUIRED_MCMC = [
    "n_procs", "station_file", "n_iter", "n_burn", "n_interval", "n_chains", "n_cool", "temp_high", "prior_z",
    "prior_width_z", "prior_width_xy", "prior_vs", "prior_width_vs", "prior_qs", "prior_width_qs",
    "prior_width_t_corr", "prior_width_a_corr", "step_size_z", "step_size_xy", "step_size_vs", "step_size_qs",
    "step_size_t_corr", "step_size_a_corr", "solve_vs", "solve_t_corr", "solve_qs", "solve_a_corr",
    "use_time", "use_amp",
]


# src/cls_param.f90:123-126
REQUIRED_SELECT = ["n_procs", "station_file", "z_guess", "vs_min", "vs_max", "b_min", "b_max"]


class ParamError(SystemExit):
    """The reference `stop`s on a bad parameter file; mirrored as SystemExit with its message."""


def _fortran_real(s: str) -> float:
    t = s.strip().lower().replace("d", "e")
    return float(t)


def _fortran_logical(s: str) -> bool:
    t = s.strip().upper().lstrip(".")
    if t.startswith("T"):
        return True
    if t.startswith("F"):
        return False
    raise ValueError(f"bad logical value {s!r}")


def parse_line(line: str):
    """-> (name, value) or None, src/cls_line_text.f90:88-148."""
    line = line.rstrip("\n")[:200]
    k = line.find("#")
    if k >= 0:
        line = line[:k]
    line = line.replace(" ", "")
    j = line.find("=")
    if j <= 0 or j == len(line) - 1:
        return None
    return line[:j], line[j + 1:]


class Param:
    def __init__(self, param_file: str, verb: bool = False, from_where: str = "mcmc"):
        self.param_file = param_file
        self.values = {"prior_t_corr": 0.0, "prior_a_corr": 0.0}
        self.given = []
        if not os.path.exists(param_file):
            raise ParamError(f"ERROR: cannot open {param_file}")
        with open(param_file) as f:
            for line in f:
                nv = parse_line(line)
                if nv is None:
                    continue
                self.set_value(*nv)
        for key in {"mcmc": REQUIRED_MCMC, "select": REQUIRED_SELECT}.get(from_where, []):
            if key not in self.given:
                raise ParamError(f"ERROR: {key} is not given.")
        base = os.path.dirname(os.path.abspath(param_file))
        sf = self.values["station_file"]
        self.read_station_file(sf if os.path.isabs(sf) or os.path.exists(sf) else os.path.join(base, sf))

    def set_value(self, name: str, val: str):
        if name in STR_KEYS:
            self.values[name] = val
        elif name in INT_KEYS:
            self.values[name] = int(val)
        elif name in REAL_KEYS:
            self.values[name] = _fortran_real(val)
        elif name in BOOL_KEYS:
            self.values[name] = _fortran_logical(val)
        else:
            raise ParamError(f"ERROR: Invalid parameter name\n        : {name}  (?)")
        self.given.append(name)

    def read_station_file(self, path: str):  # src/cls_param.f90:350-390
        if not os.path.exists(path):
            raise ParamError(f"ERROR: cannot open {path}")
        self.stations, xs, ys, zs, fac = [], [], [], [], []
        with open(path) as f:
            for line in f:
                tok = line.replace(",", " ").split()
                if not tok:
                    continue
                self.stations.append(tok[0])
                xs.append(_fortran_real(tok[1])); ys.append(_fortran_real(tok[2])); zs.append(_fortran_real(tok[3]))
                fac.append((_fortran_real(tok[4]), _fortran_real(tok[5])))
        import numpy as np

        self.sta_x, self.sta_y, self.sta_z = np.array(xs), np.array(ys), np.array(zs)
        self.sta_amp_fac = np.array(fac)
        self.n_stations = len(xs)

    def __getattr__(self, name):
        # get_<key>() accessors like the reference's ~60 getters
        if name.startswith("get_"):
            key = name[4:]
            if key == "n_stations":
                return lambda: self.n_stations
            if key in ("sta_x", "sta_y", "sta_z"):
                return lambda: getattr(self, key)
            if key in self.values:
                return lambda: self.values[key]
        raise AttributeError(name)
