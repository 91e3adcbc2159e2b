"""Host mirror of `type obs_data` (reference src/cls_obs_data.f90): reads opt_data.NNNNNN.dat
(n_sta rows x 7 list-directed columns, x y z t t_stdv amp amp_stdv; columns 1-3 ignored, :84-93) and makes
the initial x,y guess (station with the largest amplitude, :120-134).  The reference's dt_obs / dt_stdv
cubes (:101-109) are dead data (no reader) and are not built."""
from __future__ import annotations

import os

import numpy as np


class ObsData:
    def __init__(self, win_id, n_sta, sta_x, sta_y, verb=False, directory="."):
        self.win_id = [int(w) for w in win_id]
        self.n_events = len(self.win_id)
        self.n_sta = int(n_sta)
        self.sta_x = np.asarray(sta_x, dtype=np.float64)
        self.sta_y = np.asarray(sta_y, dtype=np.float64)
        self.directory = directory
        E, S = self.n_events, self.n_sta
        self.t_obs = np.empty((E, S)); self.t_stdv = np.empty((E, S))
        self.a_obs = np.empty((E, S)); self.a_stdv = np.empty((E, S))
        self.read_obs_files()

    @classmethod
    def from_arrays(cls, sta_x, sta_y, t_obs, t_stdv, a_obs, a_stdv):
        self = cls.__new__(cls)
        self.t_obs = np.ascontiguousarray(t_obs, dtype=np.float64)
        self.n_events, self.n_sta = self.t_obs.shape
        self.t_stdv = np.ascontiguousarray(t_stdv, dtype=np.float64)
        self.a_obs = np.ascontiguousarray(a_obs, dtype=np.float64)
        self.a_stdv = np.ascontiguousarray(a_stdv, dtype=np.float64)
        self.sta_x = np.asarray(sta_x, dtype=np.float64)
        self.sta_y = np.asarray(sta_y, dtype=np.float64)
        self.win_id = list(range(1, self.n_events + 1))
        return self

    def read_obs_files(self):
        for i, w in enumerate(self.win_id):
            path = os.path.join(self.directory, "opt_data.%06d.dat" % w)
            if not os.path.exists(path):
                print(path)
                raise SystemExit("ERROR: obs_file is not found")
            with open(path) as f:
                rows = [ln.replace(",", " ").split() for ln in f if ln.strip()]
            if len(rows) < self.n_sta:
                raise SystemExit(f"ERROR: {path} has {len(rows)} rows, expected {self.n_sta}")
            for j in range(self.n_sta):
                v = [float(t.lower().replace("d", "e")) for t in rows[j][:7]]
                self.t_obs[i, j], self.t_stdv[i, j], self.a_obs[i, j], self.a_stdv[i, j] = v[3], v[4], v[5], v[6]

    def make_initial_guess(self):
        ista = np.argmax(self.a_obs, axis=1)  # first maximum, like maxloc
        return self.sta_x[ista].copy(), self.sta_y[ista].copy()

    def get_t_obs(self): return self.t_obs
    def get_t_stdv(self): return self.t_stdv
    def get_a_obs(self): return self.a_obs
    def get_a_stdv(self): return self.a_stdv
