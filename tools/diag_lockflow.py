"""Run ON THE GPU BOX: two processes sharing the GPU through the in-kernel exchange on a golden case; every rank's likelihood
records against the fixture (first differing record).  python tools/diag_lockflow.py [case] [world]"""
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)


def worker(rank, world, port, name, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), HTM_XCHG="1")
    import numpy as np
    import torch.distributed as dist

    from hypotremormcmc_amd import driver
    from hypotremormcmc_amd.obs_data import ObsData
    from hypotremormcmc_amd.parallel import TorchWorld
    from tests.helpers import load_case

    dist.init_process_group("gloo", rank=rank, world_size=world)
    fx, data, params = load_case(name)
    obs = ObsData.from_arrays(data.sta_x, data.sta_y, data.t_obs, data.t_stdv, data.a_obs, data.a_stdv)
    fwd, cs = driver.build_rank(params, data.sta_x, data.sta_y, data.sta_z, obs, rank, n_procs=world, device=0)
    tw = TorchWorld(cs)
    n_iter = int(params["n_iter"])
    tw.run(n_iter)
    it, ch, lk = cs.likelihood_trace()
    fi, fl = fx[f"lik_iter_{rank}"], fx[f"lik_{rank}"]
    n = min(len(it), len(fi))
    bad = [k for k in range(n) if it[k] != fi[k] or abs(lk[k] - fl[k]) > 1e-9 * abs(fl[k])]
    q.put((rank, len(it), len(fi), bad[:3], [(int(it[k]), float(lk[k]), int(fi[k]), float(fl[k])) for k in bad[:3]], tw.direct))
    dist.destroy_process_group()


if __name__ == "__main__":
    import socket

    import torch.multiprocessing as mp

    name = sys.argv[1] if len(sys.argv) > 1 else "c1"
    world = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, world, port, name, q)) for r in range(world)]
    for p in procs:
        p.start()
    for _ in range(world):
        print(q.get(timeout=300), flush=True)
    for p in procs:
        p.join(timeout=60)
