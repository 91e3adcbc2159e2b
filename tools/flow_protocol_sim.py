#!/usr/bin/env python3
"""Model check of the synchronisation protocol of the free-running chain master (csrc/htm_flow.hpp), on the CPU.

The master's waves run their chains' steps without workgroup barriers; what keeps the run identical to the serial
loop of the reference (src/hypo_tremor_mcmc.f90:236-284) is a handful of rules on shared words in LDS:
  prog[c]   (epoch, key, reject flag) published by chain c's wave when the proposal of its step is known
  done[c]   key of the chain's last committed step;  L4 / T4: log-likelihood and temperature by iteration & 3
  epoch, anchor[epoch & 1] = (key, pos): after a Rayleigh-prior rejection (a step one draw shorter than the hop tables
            assume) every later step starts elsewhere: "step `key` starts at `pos`"
This script runs those rules as coroutines (one per wave) under a random scheduler that may switch between any two
shared-memory accesses, with state-dependent rejections, swaps and a stop request, and compares everything -- the
position every step started at, final chain states, temperatures, final stream position -- with the serial loop.

The second half models the LOCK-STEP ranks (k_mcmc<.., 4>): several ranks with their own streams and NO meeting of a rank's
waves: every chain posts its own (T, L) when it commits, the wave of the rank's last chain posts the rank's header (pair on
rank 0, the judge draw the rank would take, its stop request) -- into eight-slot rings in every rank's inbox; a chain wave
looks at rank 0's header for the pair and waits for the partner's record only if its own chain is one of the two; ranks != 0
bet that the judge draw is not theirs (lost: whoever learns the pair first anchors the next iteration one position later under
a new epoch, before any wave of the rank commits a step of it); a stop asked for in the headers of iteration j ends the job
after iteration j + 2 on every rank; after the loop the swap of the last iteration is applied to the rank's own chains --
against the serial ranks of cls_parallel.f90:121-213.  (With four-slot rings the model deadlocks: a reader of the headers of
iteration j - 4 can still be at work when a peer posts those of j.)

    python tools/flow_protocol_sim.py [n_cases]          # exits non-zero on the first difference
"""
import random
import sys


def h32(*a):
    x = 0x9E3779B9
    for v in a:
        x = (x ^ (int(v) & 0xFFFFFFFF)) * 0x85EBCA6B & 0xFFFFFFFF
        x ^= x >> 13
        x = x * 0xC2B2AE35 & 0xFFFFFFFF
        x ^= x >> 16
    return x


class Stream:
    """what the stream rings hold, as functions of the position"""

    def __init__(self, seed, nc, p_rej):
        self.seed, self.nc, self.p_rej = seed, nc, p_rej

    def w(self, p):                      # draws of a step starting at p if its prior is ok (dec.w)
        return 4 + h32(self.seed, 1, p) % 3

    def rayleigh(self, p):               # a step that CAN be rejected by its prior (z component)
        return h32(self.seed, 2, p) % 3 == 0

    def hop(self, p, n):                 # start of the step n steps after the one starting at p (optimistic)
        for _ in range(n):
            p += self.w(p)
        return p

    def sw(self, p):                     # select_pair at p: (i1, i2, draws incl. the judge draw)
        if self.nc < 2:
            return 0, 0, 0
        i1 = h32(self.seed, 3, p) % self.nc
        i2 = (i1 + 1 + h32(self.seed, 4, p) % (self.nc - 1)) % self.nc
        return i1, i2, 2 + h32(self.seed, 5, p) % 3 + 1

    def next_from(self, p, n):           # (end of this iteration's chain steps, base of the next) when n steps remain from p
        e = self.hop(p, n)
        return e, e + self.sw(e)[2]


def step_outcome(st, p, x, L, T):
    """proposal at position p on chain state (x, L) at temperature T -> (ok, cnt, accepted, x', L')"""
    ok = not (st.rayleigh(p) and h32(st.seed, 6, p, x) % 1000 < int(1000 * st.p_rej))
    cnt = st.w(p) - (0 if ok else 1)
    acc = ok and h32(st.seed, 7, p, L, T) % 100 < 45
    return ok, cnt, acc, (h32(x, p) if acc else x), (h32(L, p, 1) % 100000 if acc else L)


def swap_outcome(st, e, L1, L2, T1, T2):
    return h32(st.seed, 8, e, L1, L2, T1, T2) % 100 < 30


def serial(st, nc, n_iter, x0, L0, T0):
    x, L, T = list(x0), list(L0), list(T0)
    p, trace = 0, {}
    for i in range(1, n_iter + 1):
        for c in range(nc):
            trace[(i, c)] = p
            ok, cnt, acc, x[c], L[c] = step_outcome(st, p, x[c], L[c], T[c])
            p += cnt
        if nc > 1:
            i1, i2, d = st.sw(p)
            if swap_outcome(st, p, L[i1], L[i2], T[i1], T[i2]):
                T[i1], T[i2] = T[i2], T[i1]
            p += d
    return x, L, T, p, trace


class Shared:
    pass


def wave(sh, st, w, NW, nc, i0, target, rnd, trace):
    """one wave of the master: chains w, w + NW, ...; a generator that yields between shared-memory accesses"""
    key = lambda i, c: (i - i0) * nc + c
    nextB = lambda pos, n: st.next_from(pos, n)[1]
    my_epoch, my_akey = 0, 0             # the epoch this wave predicts in, and that epoch's anchor key
    i = i0 + 1
    # "chain rc's step of this iteration starts at rpos" (rc = 0: the iteration's base); the same for the next iteration;
    # B2 = base of the iteration after that (orders sent ahead only)
    rc, rpos = 0, 0
    rc1, rpos1 = 0, nextB(0, nc)
    B2 = nextB(rpos1, nc)
    chains = list(range(w, nc, NW))
    if not chains:
        return
    ci = 0

    def from_anchor(a_key, a_pos):
        """(iteration, chain, position) of the anchor step.  A rejected LAST step of an iteration leaves a_pos = the end
        of that iteration's chain steps: the anchor step (chain 0 of the next) starts after the swap's draws"""
        ia, ca = i0 + a_key // nc, a_key % nc
        return (ia, 0, a_pos + st.sw(a_pos)[2]) if ca == 0 else (ia, ca, a_pos)

    while True:
        c = chains[ci]
        # ---- top of a step: the epoch this step's position is predicted in
        e = sh.epoch; yield
        if e != my_epoch:
            while True:
                a_key, a_pos = sh.anchor[e & 1]; yield
                e2 = sh.epoch; yield
                if e2 == e:
                    break
                e = e2
            my_epoch, my_akey = e, a_key
            ia, ca, ap = from_anchor(a_key, a_pos)
            # every step this wave has still to run lies at or after the anchor (the rejected step right before it could
            # only be committed after all earlier steps had passed their checks -- this one has not started)
            assert key(i, c) >= a_key, (i, c, a_key)
            if ia == i:
                rc, rpos = ca, ap
            else:
                assert ia == i - 1 and ca > 0, (ia, ca, i)
                rc, rpos = 0, nextB(ap, nc - ca)
            rc1, rpos1 = 0, nextB(rpos, nc - rc)
            B2 = nextB(rpos1, nc)
        last = sh.last_iter; yield
        if i > last:
            break
        assert c >= rc
        P = st.hop(rpos, c - rc)
        if c == 0:                       # chain 0's wave: stop request of this launch (records / stream nearly used up)
            if sh.stop_at == i:
                sh.last_iter = i; yield
        # ---- front: proposal (own chain state only)
        for _ in range(rnd.randint(0, 3)):
            yield
        x, L = sh.x[c], sh.L[c]
        ok_pre = not (st.rayleigh(P) and h32(st.seed, 6, P, x) % 1000 < int(1000 * st.p_rej))
        cnt_pre = st.w(P) - (0 if ok_pre else 1)
        if c == nc - 1:
            sh.Eof[i & 3] = P + cnt_pre; yield       # where this iteration's swap starts (valid if this step is)
        sh.prog[c] = (my_epoch, key(i, c), 0 if ok_pre else 1); yield
        for _ in range(rnd.randint(0, 6)):     # evaluation
            yield
        # ---- turn: every earlier step has passed its check in this epoch (or lies before the anchor)
        restart = False
        while True:
            pr = list(sh.prog); yield
            e = sh.epoch; yield
            if e != my_epoch:
                a_key, a_pos = sh.anchor[e & 1]; yield
                e2 = sh.epoch; yield
                if e2 != e:
                    continue
                if key(i, c) >= a_key:
                    restart = True
                    break
                # this step stands; everything this wave runs after it starts at or after the anchor
                my_epoch, my_akey = e, a_key
                ia, ca, ap = from_anchor(a_key, a_pos)
                if ia == i:
                    assert ca > c
                    rc, rpos = ca, ap                 # (the wave's later chains of this iteration)
                    rc1, rpos1 = 0, nextB(ap, nc - ca)
                else:
                    assert ia == i + 1 and ca <= c, (ia, i, ca, c)
                    rc1, rpos1 = ca, ap
                B2 = nextB(rpos1, nc - rc1)
                continue
            good = True
            for c2 in range(nc):
                if c2 == c:
                    continue
                need = key(i, c2) if c2 < c else key(i - 1, c2)
                pe, pk, pf = pr[c2]
                if pk > need:
                    continue
                # the step itself: final if it lies before the anchor (checked in an earlier epoch; the one right before
                # the anchor is the committed rejection itself), else it must have passed its check in THIS epoch
                if pk == need and (pk < my_akey or (pe == my_epoch and pf == 0)):
                    continue
                good = False
            if good:
                break
            yield
        if restart:
            continue
        # ---- the swap of the iteration before, if this chain was in its pair
        T_now = sh.T4[(i - 1) & 3][c] if i - 1 > i0 else sh.T4[i & 3][c]
        if i - 1 > i0 and nc > 1:
            E_prev = sh.Eof[(i - 1) & 3]; yield      # written by the last chain's wave before it published its check
            i1, i2, d = st.sw(E_prev)
            if c in (i1, i2):
                o = i2 if c == i1 else i1
                while True:
                    dn = sh.done[o]; yield
                    if dn >= key(i - 1, o):
                        break
                L1, L2 = sh.L4[(i - 1) & 3][i1], sh.L4[(i - 1) & 3][i2]; yield
                T1, T2 = sh.T4[(i - 1) & 3][i1], sh.T4[(i - 1) & 3][i2]; yield
                if swap_outcome(st, E_prev, L1, L2, T1, T2):
                    T_now = T2 if c == i1 else T1
        sh.T4[i & 3][c] = T_now; yield
        ok, cnt, acc, x2, L2_ = step_outcome(st, P, x, L, T_now)
        assert ok == ok_pre
        trace[(i, c)] = P
        sh.x[c], sh.L[c] = x2, L2_
        sh.L4[i & 3][c] = L2_; yield
        sh.done[c] = key(i, c); yield
        if not ok:
            # a step one draw shorter than predicted: everything after it starts elsewhere
            a = (key(i, c) + 1, P + cnt)
            sh.anchor[(my_epoch + 1) & 1] = a; yield
            sh.epoch = my_epoch + 1; yield
            my_epoch, my_akey = my_epoch + 1, a[0]
            ia, ca, ap = from_anchor(*a)
            if ia == i:
                rc, rpos = ca, ap                     # (the wave's later chains of this iteration)
                rc1, rpos1 = 0, nextB(ap, nc - ca)
            else:
                rc1, rpos1 = 0, ap
            B2 = nextB(rpos1, nc - rc1)
        # ---- next step of this wave
        ci += 1
        if ci == len(chains):
            ci = 0
            i += 1
            rc, rpos = rc1, rpos1
            rc1, rpos1 = 0, B2
            B2 = nextB(B2, nc)
    return


def run_case(seed, verbose=False):
    rnd = random.Random(seed)
    nc = rnd.choice([1, 2, 3, 5, 8, 8, 11, 16, 19])
    NW = 8
    n_iter = rnd.randint(3, 40)
    p_rej = rnd.choice([0.0, 0.02, 0.2, 0.6, 1.0])
    st = Stream(seed, nc, p_rej)
    x0 = [h32(seed, 10, c) for c in range(nc)]
    L0 = [h32(seed, 11, c) % 100000 for c in range(nc)]
    T0 = [1 + c for c in range(nc)]
    stop_at = rnd.choice([None, None, rnd.randint(1, n_iter)])
    n_run = stop_at if stop_at else n_iter
    xs, Ls, Ts, ps, tr_s = serial(st, nc, n_run, x0, L0, T0)

    sh = Shared()
    sh.epoch, sh.anchor = 0, [(0, 0), (0, 0)]
    i0 = 0
    sh.prog = [(0, c, 0) for c in range(nc)]                # key(i0, c): the steps of the iteration before the launch
    sh.done = [c for c in range(nc)]
    sh.x, sh.L = list(x0), list(L0)
    sh.L4 = [[0] * nc for _ in range(4)]
    sh.T4 = [[0] * nc for _ in range(4)]
    for c in range(nc):
        sh.T4[(i0 + 1) & 3][c] = T0[c]
    sh.last_iter, sh.stop_at = n_iter, stop_at
    sh.Eof = [0, 0, 0, 0]
    trace = {}
    gens = [wave(sh, st, w, NW, nc, i0, n_iter, rnd, trace) for w in range(NW)]
    alive = list(range(NW))
    steps = 0
    bias = rnd.choice([None, None, rnd.randrange(NW)])        # one wave that is scheduled rarely (a straggler)
    while alive:
        k = rnd.choice(alive)
        if bias is not None and k == bias and rnd.random() < 0.85:
            continue
        try:
            next(gens[k])
        except StopIteration:
            alive.remove(k)
        steps += 1
        if steps > 5_000_000:
            raise RuntimeError("no progress (deadlock or livelock): seed %d" % seed)
    # after the final barrier: the swap of the last iteration, by one thread
    T = [sh.T4[n_run & 3][c] for c in range(nc)]
    L = [sh.L4[n_run & 3][c] for c in range(nc)]
    p_end = sh.Eof[n_run & 3]
    if nc > 1:
        i1, i2, d = st.sw(p_end)
        if swap_outcome(st, p_end, L[i1], L[i2], T[i1], T[i2]):
            T[i1], T[i2] = T[i2], T[i1]
        p_end += d
    ok = (sh.x == xs and sh.L == Ls and T == Ts and p_end == ps and trace == tr_s)
    if verbose or not ok:
        print("seed %d: nc %d, %d iterations (stop at %s), p_rej %.2f, epochs %d: %s" %
              (seed, nc, n_iter, stop_at, p_rej, sh.epoch, "equal" if ok else "MISMATCH"))
        if not ok:
            for k in sorted(tr_s):
                if trace.get(k) != tr_s[k]:
                    print("  first differing step", k, "serial", tr_s[k], "flow", trace.get(k))
                    break
            print("  x", sh.x == xs, "L", sh.L == Ls, "T", T == Ts, "pos", p_end, ps)
    return ok



# ======================================================================================================================
# Lock-step ranks (k_mcmc<.., 4>: flow_post_chain / flow_post_header / flow_lock_swap / flow_lock_finish of csrc/htm_flow.hpp).
# R ranks, each with its own stream and nc chains; the pair of an iteration's swap is drawn by rank 0 from ITS stream
# (select_pair), the judge draw by the rank of the pair's first chain from its own (cls_parallel.f90:121-213).  Chains post their
# (T, L), the last chain's wave the rank's header; nobody collects: each wave reads what its own chain needs.
# ======================================================================================================================
class LockStream(Stream):
    def __init__(self, seed, nc, p_rej, rank, R):
        super().__init__(seed * 131 + rank, nc, p_rej)
        self.rank, self.R = rank, R

    def pair(self, p):                   # select_pair at p on rank 0's stream: (i1, i2, draws of the pair)
        n_all = self.R * self.nc
        if n_all < 2:
            return -1, -1, 0
        i1 = h32(self.seed, 3, p) % n_all
        i2 = (i1 + 1 + h32(self.seed, 4, p) % (n_all - 1)) % n_all
        return i1, i2, 2 + h32(self.seed, 5, p) % 3

    def u(self, p):                      # the uniform draw at position p (judge_swap's number when it is taken there)
        return h32(self.seed, 9, p)

    def nd_pred(self, p):                # what THIS rank predicts the swap at p to draw from its stream (flow_swap_at's lock rule)
        if self.R * self.nc < 2 or self.rank != 0:
            return 0
        i1, i2, z = self.pair(p)
        return z + (1 if i1 // self.nc == 0 else 0)

    def next_from(self, p, n):
        e = self.hop(p, n)
        return e, e + self.nd_pred(e)


def swap_outcome_lock(u, L1, L2, T1, T2):
    return h32(u, L1, L2, T1, T2) % 100 < 30


def serial_lock(sts, nc, R, n_iter, x0, L0, T0):
    x = [list(v) for v in x0]; L = [list(v) for v in L0]; T = [list(v) for v in T0]
    p = [0] * R
    trace = {}
    for i in range(1, n_iter + 1):
        for r in range(R):
            for c in range(nc):
                trace[(r, i, c)] = p[r]
                ok, cnt, acc, x[r][c], L[r][c] = step_outcome(sts[r], p[r], x[r][c], L[r][c], T[r][c])
                p[r] += cnt
        if R * nc > 1:
            i1, i2, z = sts[0].pair(p[0])
            p[0] += z
            r1, c1, r2, c2 = i1 // nc, i1 % nc, i2 // nc, i2 % nc
            u = sts[r1].u(p[r1]); p[r1] += 1
            if swap_outcome_lock(u, L[r1][c1], L[r2][c2], T[r1][c1], T[r2][c2]):
                T[r1][c1], T[r2][c2] = T[r2][c2], T[r1][c1]
    return x, L, T, p, trace


def wave_lock(sh, st, net, w, NW, nc, R, rank, i0, rnd, trace):
    """one chain wave of a lock-step rank; `net[q]` = rank q's inbox: "H"[j & 7][r] = (j, header of rank r), "C"[j & 7][(r, c)] =
    (j, T, L) of chain c of rank r after iteration j"""
    key = lambda i, c: (i - i0) * nc + c
    nextB = lambda pos, n: st.next_from(pos, n)[1]
    my_epoch, my_akey = 0, 0
    i = i0 + 1
    rc, rpos = 0, 0
    rc1, rpos1 = 0, nextB(0, nc)
    B2 = nextB(rpos1, nc)
    chains = list(range(w, nc, NW))
    if not chains:
        return
    ci = 0

    def from_anchor(a_key, a_pos):
        ia, ca = i0 + a_key // nc, a_key % nc
        return (ia, 0, a_pos + st.nd_pred(a_pos)) if ca == 0 else (ia, ca, a_pos)

    def adopt(e, a_key, a_pos, it, c, in_turn):
        """flow_adopt: returns `stands`"""
        nonlocal my_epoch, my_akey, rc, rpos, rc1, rpos1, B2
        stands = in_turn and key(it, c) < a_key
        my_epoch, my_akey = e, a_key
        ia, ca, ap = from_anchor(a_key, a_pos)
        if not stands:
            if ia == it:
                rc, rpos = ca, ap
            else:
                assert ia == it - 1 and ca > 0, (ia, ca, it)
                rc, rpos = 0, nextB(ap, nc - ca)
            rc1, rpos1 = 0, nextB(rpos, nc - rc)
            B2 = nextB(rpos1, nc)
        elif ia == it:
            rc, rpos = ca, ap
            rc1, rpos1 = 0, nextB(ap, nc - ca)
            B2 = nextB(rpos1, nc)
        else:
            rc1, rpos1 = ca, ap
            B2 = nextB(ap, nc - ca)
        return stands

    def read_anchor():
        """generator: (e, a_key, a_pos) read consistently"""
        e = sh.epoch; yield
        while True:
            a_key, a_pos = sh.anchor[e & 1]; yield
            e2 = sh.epoch; yield
            if e2 == e:
                return e, a_key, a_pos
            e = e2

    while True:
        c = chains[ci]
        e = sh.epoch; yield
        if e != my_epoch:
            e, a_key, a_pos = yield from read_anchor()
            assert key(i, c) >= a_key, (i, c, a_key)
            adopt(e, a_key, a_pos, i, c, False)
        last = sh.last_iter; yield
        if i > last:
            break
        assert c >= rc
        P = st.hop(rpos, c - rc)
        for _ in range(rnd.randint(0, 3)):
            yield
        x, L = sh.x[c], sh.L[c]
        ok_pre = not (st.rayleigh(P) and h32(st.seed, 6, P, x) % 1000 < int(1000 * st.p_rej))
        cnt_pre = st.w(P) - (0 if ok_pre else 1)
        if c == nc - 1:
            sh.Eof[i & 3] = P + cnt_pre; yield
        sh.prog[c] = (my_epoch, key(i, c), 0 if ok_pre else 1); yield
        for _ in range(rnd.randint(0, 6)):
            yield
        restart = stop = False
        while True:
            pr = list(sh.prog); yield
            e = sh.epoch; yield
            if e != my_epoch:
                e, a_key, a_pos = yield from read_anchor()
                if not adopt(e, a_key, a_pos, i, c, True):
                    restart = True
                    break
                continue
            good = True
            for c2 in range(nc):
                if c2 == c:
                    continue
                need = key(i, c2) if c2 < c else key(i - 1, c2)
                pe, pk, pf = pr[c2]
                if pk > need:
                    continue
                if pk == need and (pk < my_akey or (pe == my_epoch and pf == 0)):
                    continue
                good = False
            if good:
                break
            last = sh.last_iter; yield
            if i > last:
                stop = True
                break
            yield
        if restart:
            continue
        if stop:
            break
        # ---- the swap of the iteration before (flow_step, LOCK): the pair is rank 0's; only the two chains it names look at
        # ---- anybody's (T, L); a stop asked for in the headers of iteration i - 2 ends the job after THIS iteration
        T_now = sh.temp[c]; yield
        if i - 1 > i0 and R * nc > 1:
            if i - 2 > i0:
                while True:
                    box = dict(net[rank]["H"][(i - 2) & 7]); yield
                    if all(box.get(q, (None,))[0] == i - 2 for q in range(R)):
                        break
                if any(box[q][1]["stop"] for q in range(R)):
                    sh.last_iter = min(sh.last_iter, i); yield          # (atomic min)
            if rank == 0:
                E1 = sh.Eof[(i - 1) & 3]; yield
                i1, i2, z1 = st.pair(E1)
            else:
                while True:
                    h = net[rank]["H"][(i - 1) & 7].get(0); yield
                    if h is not None and h[0] == i - 1:
                        break
                i1, i2 = h[1]["i1"], h[1]["i2"]
                E1 = sh.Eof[(i - 1) & 3]; yield
                z1 = 0
            r1 = i1 // nc
            if rank != 0:
                # the bet that the judge draw is not this rank's: whoever learns the pair first settles it for the rank
                xa = sh.xanch; yield
                if xa < i - 1:
                    mine = False
                    if sh.xanch_claim < i - 1:                         # (atomic max)
                        sh.xanch_claim = i - 1; mine = True
                    yield
                    if mine:
                        if r1 == rank:
                            e1 = sh.epoch + 1
                            sh.anchor[e1 & 1] = (key(i, 0), E1 + 1); yield
                            sh.epoch = e1; yield
                        sh.xanch = i - 1; yield
                    else:
                        while True:
                            xa = sh.xanch; yield
                            if xa >= i - 1:
                                break
            e = sh.epoch; yield
            if e != my_epoch:
                e, a_key, a_pos = yield from read_anchor()
                if not adopt(e, a_key, a_pos, i, c, True):
                    continue                               # the step starts elsewhere: again
            g = rank * nc + c
            if g == i1 or g == i2:
                gp = i2 if g == i1 else i1
                rp, cp = gp // nc, gp % nc
                if rp == rank:
                    while True:
                        d = sh.done[cp]; yield
                        if d >= key(i - 1, cp):
                            break
                    Tp, Lp = sh.T4[(i - 1) & 3][cp], sh.L4[(i - 1) & 3][cp]; yield
                else:
                    while True:
                        cr = net[rank]["C"][(i - 1) & 7].get((rp, cp)); yield
                        if cr is not None and cr[0] == i - 1:
                            break
                    Tp, Lp = cr[1], cr[2]
                if r1 == rank:
                    u = st.u(E1 + z1)
                else:
                    while True:
                        h1 = net[rank]["H"][(i - 1) & 7].get(r1); yield
                        if h1 is not None and h1[0] == i - 1:
                            break
                    u = h1[1]["u"]
                if g == i1:
                    acc_s = swap_outcome_lock(u, L, Lp, T_now, Tp)
                else:
                    acc_s = swap_outcome_lock(u, Lp, L, Tp, T_now)
                if acc_s:
                    T_now = Tp
                sh.temp[c] = T_now; yield
        last = sh.last_iter; yield
        if i > last:
            break
        ok, cnt, acc, x2, L2_ = step_outcome(st, P, x, L, T_now)
        assert ok == ok_pre
        trace[(rank, i, c)] = P
        sh.x[c], sh.L[c] = x2, L2_; yield
        sh.T4[i & 3][c], sh.L4[i & 3][c] = T_now, L2_; yield
        sh.done[c] = key(i, c); yield
        order = list(range(R)); rnd.shuffle(order)
        for q in order:                               # this chain's record: peer writes land one rank after the other
            net[q]["C"][i & 7][(rank, c)] = (i, T_now, L2_); yield
        if c == nc - 1:
            # the rank's header: pair (rank 0), the judge draw this rank would take, its stop request
            E = P + cnt
            i1h, i2h, zh = st.pair(E) if rank == 0 else (-1, -1, 0)
            rec = {"i1": i1h, "i2": i2h, "u": st.u(E + zh), "stop": sh.stop_at == i}
            rnd.shuffle(order)
            for q in order:
                net[q]["H"][i & 7][rank] = (i, rec); yield
        if not ok:
            a = (key(i, c) + 1, P + cnt)
            sh.anchor[(my_epoch + 1) & 1] = a; yield
            sh.epoch = my_epoch + 1; yield
            e, a_key, a_pos = yield from read_anchor()
            adopt(e, a_key, a_pos, i, c, True)
        ci += 1
        if ci == len(chains):
            ci = 0
            i += 1
            rc, rpos = rc1, rpos1
            rc1, rpos1 = 0, B2
            B2 = nextB(B2, nc)
    return


def finish_lock(sh, st, net, nc, R, rank, i0, waves_alive):
    """flow_body's epilogue on a lock-step rank: when all its waves have left, the swap of the last iteration is applied to the
    rank's own chains and the stream position settled"""
    while waves_alive[rank]:
        yield
    last = min(sh.last_iter, sh.target)
    if last > i0 and R * nc > 1:
        E = sh.Eof[last & 3]; yield
        if rank == 0:
            i1, i2, z = st.pair(E)
        else:
            while True:
                h = net[rank]["H"][last & 7].get(0); yield
                if h is not None and h[0] == last:
                    break
            i1, i2, z = h[1]["i1"], h[1]["i2"], 0
        r1 = i1 // nc
        for g, gp in ((i1, i2), (i2, i1)):
            if g // nc != rank:
                continue
            c, rp, cp = g % nc, gp // nc, gp % nc
            if rp == rank:
                Tp, Lp = sh.T4[last & 3][cp], sh.L4[last & 3][cp]
            else:
                while True:
                    cr = net[rank]["C"][last & 7].get((rp, cp)); yield
                    if cr is not None and cr[0] == last:
                        break
                Tp, Lp = cr[1], cr[2]
            if r1 == rank:
                u = st.u(E + z)
            else:
                while True:
                    h1 = net[rank]["H"][last & 7].get(r1); yield
                    if h1 is not None and h1[0] == last:
                        break
                u = h1[1]["u"]
            Tm, Lm = sh.T4[last & 3][c], sh.L4[last & 3][c]
            acc_s = swap_outcome_lock(u, Lm, Lp, Tm, Tp) if g == i1 else swap_outcome_lock(u, Lp, Lm, Tp, Tm)
            sh.newT[c] = Tp if acc_s else Tm
        for c in range(nc):
            if sh.newT[c] is not None:
                sh.temp[c] = sh.newT[c]
        sh.spos = E + z + (1 if r1 == rank else 0)
    elif last > i0:
        sh.spos = sh.Eof[last & 3]
    sh.iter_done = last
    yield


def run_case_lock(seed, verbose=False):
    rnd = random.Random(seed * 7919 + 1)
    R = rnd.choice([1, 2, 2, 3, 4])
    nc = rnd.choice([1, 2, 3, 5, 8, 8, 11, 16])
    NW = 8
    n_iter = rnd.randint(3, 30)
    p_rej = rnd.choice([0.0, 0.02, 0.2, 0.6, 1.0])
    sts = [LockStream(seed, nc, p_rej, r, R) for r in range(R)]
    x0 = [[h32(seed, 10, r, c) for c in range(nc)] for r in range(R)]
    L0 = [[h32(seed, 11, r, c) % 100000 for c in range(nc)] for r in range(R)]
    T0 = [[1 + r * nc + c for c in range(nc)] for r in range(R)]
    stop = rnd.choice([None, None, (rnd.randrange(R), rnd.randint(1, n_iter))])      # (rank, iteration) that asks everybody to stop
    # a stop asked for in the headers of iteration j ends the job after iteration j + 2 (every rank reads the same headers)
    n_run = min(n_iter, stop[1] + 2) if (stop and R * nc > 1) else n_iter
    xs, Ls, Ts, ps, tr_s = serial_lock(sts, nc, R, n_run, x0, L0, T0)
    net = [{"H": [{} for _ in range(8)], "C": [{} for _ in range(8)]} for _ in range(R)]
    shs, gens, trace = [], [], {}
    i0 = 0
    waves_alive = [0] * R
    for r in range(R):
        sh = Shared()
        sh.epoch, sh.anchor = 0, [(0, 0), (0, 0)]
        sh.prog = [(0, c, 0) for c in range(nc)]
        sh.done = [c for c in range(nc)]
        sh.x, sh.L, sh.temp = list(x0[r]), list(L0[r]), list(T0[r])
        sh.T4 = [[None] * nc for _ in range(4)]; sh.L4 = [[None] * nc for _ in range(4)]
        sh.newT = [None] * nc
        sh.last_iter = n_iter; sh.target = n_iter
        sh.stop_at = stop[1] if (stop and stop[0] == r) else None
        sh.Eof = [0, 0, 0, 0]
        sh.xanch, sh.xanch_claim, sh.spos, sh.iter_done = i0, i0, 0, i0
        shs.append(sh)
        for w in range(NW):
            if w < nc:
                waves_alive[r] += 1
            gens.append((r, True, wave_lock(sh, sts[r], net, w, NW, nc, R, r, i0, rnd, trace)))
        gens.append((r, False, finish_lock(sh, sts[r], net, nc, R, r, i0, waves_alive)))
    alive = list(range(len(gens)))
    slow_rank = rnd.choice([None, None, rnd.randrange(R)])      # a rank that is scheduled rarely
    steps = 0
    while alive:
        k = rnd.choice(alive)
        if slow_rank is not None and gens[k][0] == slow_rank and rnd.random() < 0.8:
            continue
        try:
            next(gens[k][2])
        except StopIteration:
            alive.remove(k)
            if gens[k][1] and k % (NW + 1) < nc:
                waves_alive[gens[k][0]] -= 1
        steps += 1
        if steps > 3_000_000:
            for k2 in alive:                              # where every coroutine stands
                fr = gens[k2][2].gi_frame
                print("  rank %d %s line %d  i=%s c=%s my_epoch=%s" % (gens[k2][0], "wave" if gens[k2][1] else "finish", fr.f_lineno,
                      fr.f_locals.get("i"), fr.f_locals.get("c"), fr.f_locals.get("my_epoch")))
            for r in range(R):
                print("  rank %d epoch %d anchor %s prog %s xanch %d last %d" % (r, shs[r].epoch, shs[r].anchor, shs[r].prog, shs[r].xanch, shs[r].last_iter))
            raise RuntimeError("no progress (deadlock or livelock): lock-step seed %d" % seed)
    ok = all(shs[r].x == xs[r] and shs[r].L == Ls[r] and shs[r].temp == Ts[r] and shs[r].spos == ps[r] and
             shs[r].iter_done == n_run for r in range(R)) and trace == tr_s
    if verbose or not ok:
        print("lock-step seed %d: %d ranks x %d chains, %d iterations (stop %s), p_rej %.2f: %s" %
              (seed, R, nc, n_iter, stop, p_rej, "equal" if ok else "MISMATCH"))
        if not ok:
            for k in sorted(tr_s):
                if trace.get(k) != tr_s[k]:
                    print("  first differing step (rank, iteration, chain)", k, "serial", tr_s[k], "flow", trace.get(k))
                    break
            extra = sorted(k for k in trace if k not in tr_s)
            if extra:
                print("  steps the serial ranks did not take:", extra[:6])
            for r in range(R):
                print("  rank", r, "x", shs[r].x == xs[r], "L", shs[r].L == Ls[r], "T", shs[r].temp == Ts[r], "pos", shs[r].spos, ps[r],
                      "done", shs[r].iter_done, n_run)
    return ok


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    bad = 0
    for s in range(1, n + 1):
        if not run_case(s):
            bad += 1
            break
    print("%d cases: %s" % (n, "all equal to the serial loop" if not bad else "FAILED"))
    if not bad:
        for s in range(1, n + 1):
            if not run_case_lock(s):
                bad += 1
                break
        print("%d lock-step cases: %s" % (n, "all equal to the serial ranks" if not bad else "FAILED"))
    sys.exit(1 if bad else 0)
