"""Multi-rank lock-step execution: the build's counterpart of `parallel%swap_temperature`
(reference src/cls_parallel.f90:100-216) and `parallel%output_proposal` (:244-281).

One process per GPU.  The reference exchanges a 4-int broadcast plus two 2-double messages per iteration
over MPI; here every rank contributes ONE small record {pair chosen by rank 0, its pending judge_swap
draw, (T, L) of all its chains} per iteration and every rank evaluates the identical swap decision from the
n_procs records on its device.  Transports, fastest first: (1) persistent lock-step -- the kernels write the
records straight into each other's memory (peer-mapped inboxes over xGMI, set up with one all-gather of IPC
handles) and never leave the GPU; (2) one RCCL all-gather per iteration enqueued from C (torch.distributed's
"nccl" backend = RCCL over xGMI); (3) torch.distributed calls per iteration (gloo on CPU for tests).

`LocalWorld` runs all ranks of a job inside one process on one device (record exchange = device copies);
it exists for parity tests on a single GPU.
"""
from __future__ import annotations

import numpy as np

from .chains import ChainSet


class _DevBuf:
    """Zero-copy view of a raw device pointer for torch.as_tensor (CUDA array interface, version 2)."""

    def __init__(self, ptr: int, n_f64: int):
        self.__cuda_array_interface__ = {"shape": (n_f64,), "typestr": "<f8", "data": (ptr, False), "version": 2}


def record_tensor(cs: ChainSet):
    import torch

    ptr, nbytes = cs.swap_record()
    return torch.as_tensor(_DevBuf(ptr, nbytes // 8), device=f"cuda:{cs.fwd.device}")


class DeviceRank:
    """Adapter: one rank's ChainSet on its GPU, in the shape TorchWorld drives."""

    def __init__(self, cs: ChainSet):
        import torch

        self.cs = cs
        self.n_procs, self.rank = cs.n_procs, cs.rank
        torch.cuda.set_device(cs.fwd.device)
        # our kernels and the collective must share one stream order
        cs.fwd.set_stream(torch.cuda.current_stream().cuda_stream)
        self.record = record_tensor(cs)

    def step_begin(self):
        self.cs.step_begin()

    def step_end(self, gathered):
        self.cs.step_end(gathered.data_ptr())

    def drain(self):
        self.cs.drain()

    def counts(self):
        return self.cs.counts()


class TorchWorld:
    """One rank of a torch.distributed job (already initialised; backend "nccl" = RCCL over xGMI on the GPUs,
    "gloo" for the CPU tests).  `rank_obj` is a DeviceRank (or a ChainSet, wrapped here); anything with
    .n_procs .rank .record (a float64 tensor viewing the rank's swap record) .step_begin() .step_end(gathered)
    .drain() .counts() works, which is how the CPU tests drive the protocol without a GPU."""

    def __init__(self, rank_obj, group=None):
        import torch
        import torch.distributed as dist

        if isinstance(rank_obj, ChainSet):
            rank_obj = DeviceRank(rank_obj)
        self.torch, self.dist, self.r, self.group = torch, dist, rank_obj, group
        self.world = dist.get_world_size(group)
        if self.world != rank_obj.n_procs or dist.get_rank(group) != rank_obj.rank:
            raise ValueError("torch.distributed world/rank do not match the chain set's n_procs/rank")
        self.rec = rank_obj.record
        # A backend without device collectives (gloo) and a rank on a GPU: stage the 160-byte records through
        # host memory (htm_chains_swap_record_host / _step_end_host) -- the exchange an MPI program would do.
        self.host_staged = isinstance(rank_obj, DeviceRank) and dist.get_backend(group) != "nccl"
        if self.host_staged:
            self.rec = torch.zeros(self.rec.numel(), dtype=torch.float64)
        self.gathered = torch.zeros(self.world * self.rec.numel(), dtype=torch.float64, device=self.rec.device)
        # records a rank may hold between drains: n_chains per iteration at most
        self.drain_every = 4096
        self._share_gpus()
        self.direct = self._try_direct_exchange()
        self.fast = None if self.direct else self._try_direct_rccl()
        # The set-up probe proves one token per peer, not the loop.  The FIRST direct run is therefore guarded: the rank's
        # state is saved before it, every rank reports how it ended, and if any rank failed (error -10: a peer's records
        # did not arrive; -11: a peer reported a failure) ALL ranks reload their state and repeat the iterations on the
        # per-iteration all-gather path.  `fell_back` says so (bench.py prints it).
        self._direct_proven = False
        self._probe_token = 1
        self.fell_back = None

    def _share_gpus(self):
        """Ranks whose chains sit on the same GPU of the same host (a test arrangement, or deliberately more masters per GPU,
        DESIGN.md 6) must all be resident at once: count them and let every rank take its share of the CUs."""
        if not isinstance(self.r, DeviceRank) or self.world == 1:
            return
        import socket

        # (the GPU by its PCI address, not by its ordinal: with a visible-devices mask per rank every rank's ordinal is 0)
        import ctypes
        from . import _lib

        pid = ctypes.c_int(-1)
        dev = int(self.r.cs.fwd.device)
        if _lib.load().htm_device_physical_id(dev, ctypes.byref(pid)) != 0:
            pid.value = -1 - dev
        mine = (socket.gethostname(), int(pid.value))
        where = [None] * self.world
        self.dist.all_gather_object(where, mine, group=self.group)
        k = sum(1 for w in where if tuple(w) == mine)
        if k > 1:
            self.r.cs.share_gpu(k)

    def transport_name(self):
        if self.direct:
            return ("persistent lock-step: swap records written into the peers' inboxes from inside the kernel "
                    "(xGMI peer memory)")
        if self.fast is not None:
            return "one k_mcmc launch + one RCCL all-gather per iteration, enqueued from C"
        return "torch.distributed all-gather per iteration"

    def _all_ok(self, ok: bool) -> bool:
        dev = self.gathered.device if self.dist.get_backend(self.group) == "nccl" else self.torch.device("cpu")
        flag = self.torch.tensor([1 if ok else 0], dtype=self.torch.int32, device=dev)
        self.dist.all_reduce(flag, op=self.dist.ReduceOp.MIN, group=self.group)
        return bool(int(flag.item()))

    def _try_direct_exchange(self):
        """Persistent lock-step (htm_chains_run_lockstep_direct): every rank's kernel writes its swap record straight
        into the other ranks' inboxes (peer-mapped device memory, xGMI) and stays resident over the iterations.  Set-up
        = one all-gather of the inboxes' IPC handles over this group (RCCL, or gloo in the CPU-launched tests); used
        only if EVERY rank could map every peer, else the per-iteration all-gather paths below take over.
        HTM_XCHG=0 disables it."""
        import os

        if os.environ.get("HTM_XCHG", "1") == "0" or not isinstance(self.r, DeviceRank):
            return False
        torch, dist, cs = self.torch, self.dist, self.r.cs
        dev = self.gathered.device if dist.get_backend(self.group) == "nccl" else torch.device("cpu")
        nb = cs.XCHG_HANDLE_BYTES
        ok, mine = 1, bytes(nb)
        try:
            mine = cs.xchg_handle()
        except Exception:
            ok = 0
        t_in = torch.tensor(list(mine), dtype=torch.uint8, device=dev)
        t_out = torch.empty(self.world * nb, dtype=torch.uint8, device=dev)
        dist.all_gather_into_tensor(t_out, t_in, group=self.group)
        if ok:
            try:
                cs.xchg_connect(bytes(t_out.cpu().numpy().tobytes()) if self.world > 1 else None)
            except Exception:
                ok = 0
        flag = torch.tensor([ok], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
        if not int(flag.item()):
            return False
        # every rank mapped every peer: prove that writes into the inboxes reach a polling kernel (all ranks probe together)
        dist.barrier(group=self.group)
        try:
            cs.xchg_probe(token=1, seconds=10.0)       # (the library adds the set's probe count to the token)
        except Exception:
            ok = 0
        flag = torch.tensor([ok], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
        return bool(int(flag.item()))

    def _try_direct_rccl(self):
        """(function pointer, communicator) of RCCL's ncclAllGather on torch's own communicator, so that the
        iteration loop can run in C (one kernel launch + one all-gather enqueue per iteration, no Python in
        between).  None -> the torch.distributed call per iteration is used (also the gloo / CPU-test path)."""
        import ctypes
        import os

        if os.environ.get("HTM_DIRECT_RCCL", "1") == "0" or not isinstance(self.r, DeviceRank):
            return None
        try:
            if self.dist.get_backend(self.group) != "nccl":
                return None
            # make sure the communicator exists, then fetch it
            self.dist.all_gather_into_tensor(self.gathered, self.rec, group=self.group)
            self.torch.cuda.synchronize()
            pg = self.group if self.group is not None else self.dist.distributed_c10d._get_default_group()
            backend = pg._get_backend(self.torch.device("cuda", self.r.cs.fwd.device))
            comm = int(backend._comm_ptr())
            lib = ctypes.CDLL(os.path.join(os.path.dirname(self.torch.__file__), "lib", "librccl.so"))
            fn = ctypes.cast(lib.ncclAllGather, ctypes.c_void_p).value
            if not comm or not fn:
                return None
            return fn, comm
        except Exception:
            return None

    def step(self):
        self.r.step_begin()
        if self.host_staged:
            self.r.cs.swap_record_host(self.rec.numpy())
            self.dist.all_gather_into_tensor(self.gathered, self.rec, group=self.group)
            self.r.cs.step_end_host(self.gathered.numpy())
            return
        self.dist.all_gather_into_tensor(self.gathered, self.rec, group=self.group)
        self.r.step_end(self.gathered)

    def run(self, n_iter: int):
        if self.direct:
            cs = self.r.cs
            blob = None if self._direct_proven else cs.checkpoint()
            self.dist.barrier(group=self.group)      # the ranks' kernels start together (the exchange waits 20 s for a peer)
            err = None
            try:
                cs.run_lockstep_direct(n_iter)
            except Exception as exc:
                if self._direct_proven:
                    raise
                err = exc
            if self._direct_proven:
                return
            if self._all_ok(err is None):
                self._direct_proven = True
                return
            # some rank's exchange failed inside the loop: everybody goes back to the saved state and takes the
            # all-gather path from here on (restore clears this rank's inbox; nobody posts again: direct is off)
            cs.restore(blob)
            self.dist.barrier(group=self.group)
            self.direct = False
            self.fast = self._try_direct_rccl()
            self.fell_back = ("the in-kernel exchange failed in its first run (%s); state reloaded, iterations repeated on: %s"
                              % (err if err is not None else "on another rank", self.transport_name()))
        if self.fast is not None:
            self.r.cs.run_lockstep(n_iter, self.fast[0], self.fast[1], self.gathered.data_ptr())
            self.r.drain()
            return
        for k in range(n_iter):
            self.step()
            if (k + 1) % self.drain_every == 0:
                self.r.drain()
        self.r.drain()

    def reduce_counts(self):
        npr, nac = self.r.counts()
        t = self.torch.tensor(np.concatenate([npr, nac]).astype(np.int64), device=self.gathered.device)
        self.dist.all_reduce(t, group=self.group)
        t = t.cpu().numpy()
        return t[:7], t[7:]


class LocalWorld:
    """All ranks of a job in one process, on one device (test harness for the lock-step path)."""

    def __init__(self, chain_sets):
        import torch

        self.torch = torch
        self.sets = list(chain_sets)
        self.world = len(self.sets)
        dev = self.sets[0].fwd.device
        torch.cuda.set_device(dev)
        stream = torch.cuda.current_stream().cuda_stream
        seen = set()
        for cs in self.sets:
            if id(cs.fwd) not in seen:
                cs.fwd.set_stream(stream)
                seen.add(id(cs.fwd))
        self.recs = [record_tensor(cs) for cs in self.sets]
        n = self.recs[0].numel()
        self.gathered = torch.zeros(self.world * n, dtype=torch.float64, device=self.recs[0].device)
        self.n = n

    def step(self):
        for cs in self.sets:
            cs.step_begin()
        for r, rec in enumerate(self.recs):
            self.gathered[r * self.n:(r + 1) * self.n].copy_(rec)
        for cs in self.sets:
            cs.step_end(self.gathered.data_ptr())

    def run(self, n_iter: int, drain_every: int = 1024):
        for k in range(n_iter):
            self.step()
            if (k + 1) % drain_every == 0:
                for cs in self.sets:
                    cs.drain()
        for cs in self.sets:
            cs.drain()
