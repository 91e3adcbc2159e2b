// htm_pipe.hpp -- the PIPELINED chain master of k_mcmc (MK 5: single rank; MK 6: a lock-step rank): the main loop of a rank
// (hypo_tremor_mcmc.f90:236-284) as a three-stage pipeline over ITERATIONS inside block 0.
//
// htm_flow.hpp gives every chain a wave that runs the chain's steps back to back: ~1 300 instructions per step of which ~230
// are the evaluation -- the rest is control (positions, look-ups, orders, turns) executed wave-wide for ONE chain, and a step
// is one dependent chain of ~9.5 k cycles.  But what a step really depends on in the step before it is tiny: the running
// log-likelihood (additive), the temperature, and -- rarely -- the very element or event it touches.  So here
//   F  (wave 0, lanes <-> chains)   the FRONT runs up to kPipeAhead iterations ahead of the decisions: stream positions of every
//        chain step (prefix scan of the draw counts, Rayleigh rejections included: cls_model.f90:178-181, cls_mcmc.f90:193),
//        decoded proposals, perturbed values, prior ratios (cls_model.f90:162-190), the iteration's swap pair
//        (cls_parallel.f90:226-230) -- one PROPOSAL RECORD per chain and iteration in an LDS ring of slots; it also sends the
//        work orders of the coming full evaluations to the worker blocks (as early as the chain's state allows) and keeps the
//        LDS window of the stream rings ahead (it is its only reader);
//   E  (waves 2..7, lanes <-> stations)   EVALUATORS take (iteration, chain) tasks from an in-order queue: the two-position
//        single-event update (cls_forward.f90:307-362) from the record alone (event coordinates travel in the record; station
//        corrections, 1/vs, pi f/(qs vs) from an LDS mirror under a sequence lock), or -- for a step that needs the full
//        evaluation (cls_forward.f90:268-303) -- the collection of the workers' partial sums plus the events the workers left
//        out;
//   D  (wave 1, lanes <-> chains)   the DECIDER takes the iterations in order: validates what F and E assumed, applies the swap
//        of the iteration before (cls_parallel.f90:121-136, :285-302), Metropolis (cls_mcmc.f90:193-203), commit, counters,
//        records (hypo_tremor_mcmc.f90:270-280) -- ~250 instructions for ALL chains of an iteration.
// Everything F and E do ahead of D is SPECULATION on the chain state, checked by D before it decides:
//   * F reads the element it perturbs (and the event's coordinates) from memory at a known commit count of the chain (`cland`:
//     commits whose stores have landed); D keeps a ring of the chain's last commits and flushes if any commit since then touched
//     that element or event;
//   * an evaluation records the parameter version (`pver`, bumped by every accepted vs / qs / correction commit) it was made
//     under.  An evaluator that sees ONE undecided full-evaluation step of its chain ahead of its own step evaluates twice --
//     under the current parameters and under that step's proposal -- and D picks by that step's outcome; D flushes if the
//     version it finds is neither.
// A FLUSH (new epoch) throws away every record and result from the current iteration on; F starts again from the final state.
// It is the one recovery path: rare at production sizes (a conflict needs the same event twice within ~3 steps of a chain),
// frequent and merely slow on toy sizes.  Nothing D has decided is ever undone.
// A full evaluation's value does not depend on when its order went out: the events of the (up to two) hypocentre steps right
// before it are ALWAYS left out by the workers and added by the evaluator -- at both candidate positions, D picks -- in a fixed
// association, so the bits are a function of the stream and the state alone.
#pragma once
#include "htm_flow.hpp"

namespace htm {

constexpr int kPipeSlots = 8;     // iterations in the slot ring
constexpr int kPipeAhead = 6;     // F produces iteration it only when it <= decided + kPipeAhead   (kPipeSlots >= kPipeAhead + 2)
constexpr int kPipeLog = 8;       // commits per chain D remembers

enum { PI_POS, PI_TYPE, PI_IDX, PI_EVT, PI_O, PI_CNT, PI_SEQ, PI_KIND, PI_LO1, PI_LO2, PI_N };
enum { PD_XNEW, PD_LPR, PD_R, PD_LOGR, PD_HX, PD_HY, PD_HZ, PD_N };
enum { EI_TAG, EI_PV, EI_DEP, EI_N };
enum { ED_N = 5 };
// PI_KIND: bits 0..3 kind (0 prior rejected: no evaluation, 1 single-event update, 2 full evaluation), bits 4..7 how many steps
// ahead of the chain's decisions the order may go (0..2), bit 8 / 9: the event of the step one / two before is left out

struct PipeHdr {
    int tag;                      // (epoch & 0xff) << 24 | iteration & 0xffffff: the slot is complete for that iteration
    int base, E;                  // where the iteration's chain steps start and end
    int i1, i2, nd;               // its swap: pair, draws
    double sr, slr;               // judge_swap's draw and its log
};

struct PipeShared : StepShared {
    PipeHdr hdr[kPipeSlots];
    unsigned long long q;         // {epoch : 32, next task : 32}: the evaluators' queue (task t = (iteration - i0 - 1) * n_chains + chain)
    int epoch, fl_it, fl_base;    // a flush: everything from iteration fl_it (which starts at fl_base) on is produced again
    int d_done;                   // iterations <= d_done are decided
    int landed_it;                // the commits of iterations <= landed_it have landed in memory
    int f_it;                     // F has produced iterations <= f_it (in the current epoch)
    int f_stop_it;                // F cannot produce this iteration (end of the produced stream); 0 = none
    int win_fill, win_want, win_lo;   // the LDS window of the stream rings holds [.., win_fill); F asks for [.., win_want); positions < win_lo are dead (C extends it)
    int win_seq, win_ack;             // F went BACK (a flush): it reads the window again only when C has seen the new win_lo
    int quit;
    int i0;
    int pver[kMaxChains];         // parameter version of the chain (odd: a commit is being written)
    int cland[kMaxChains];        // commits of the chain whose stores have landed
    int clog[kMaxChains][kPipeLog];      // elements of the chain's latest commits
    double rbeta[kMaxChains], katt[kMaxChains];      // 1 / vs and pi f / (qs vs) (cls_forward.f90:118, :204)
    int col_w[kMaxChains];        // the order out for the chain, as pipe_tag(epoch it was sent in, its iteration); 0 = none (F posts, C clears)
    unsigned col_tag[kMaxChains]; // its granule tag
    unsigned long long n_flush;
    unsigned trace_n;
};

struct PipeRings {                // the slot ring, carved from dynamic LDS: [slot][field][chain]
    int *pi; double *pd;
    int *ei; double *ed;          // [epoch & 1][slot][field][chain]
    int nc;
    __device__ __forceinline__ int &I(int s, int f, int c) const { return pi[(s * PI_N + f) * nc + c]; }
    __device__ __forceinline__ double &D(int s, int f, int c) const { return pd[(s * PD_N + f) * nc + c]; }
    __device__ __forceinline__ int &EI(int b, int s, int f, int c) const { return ei[((b * kPipeSlots + s) * EI_N + f) * nc + c]; }
    __device__ __forceinline__ double &ED(int b, int s, int f, int c) const { return ed[((b * kPipeSlots + s) * ED_N + f) * nc + c]; }
};
__host__ __device__ inline size_t pipe_ring_bytes(int nc)
{
    return (size_t)kPipeSlots * nc * (PI_N * 4 + PD_N * 8) + (size_t)2 * kPipeSlots * nc * (EI_N * 4 + ED_N * 8) + 64;
}

__device__ __forceinline__ int pipe_tag(int epoch, int it) { return ((epoch & 0xff) << 24) | (it & 0xffffff); }
__device__ __forceinline__ int rl_i32(int v, int l) { return __builtin_amdgcn_readlane(v, l); }

#ifdef HTM_STAMPS
// event trace of a few iterations (tools/pipe_trace.py): {time, code << 48 | iteration << 8 | chain} appended to ChainsDev::stamps + 128
#define PTRACE(code, it_, c_) do { if (lane == 0 && (it_) - sh.i0 >= 3000 && (it_) - sh.i0 < 3040 && sh.i0 > 0) { \
        const unsigned k_ = atomicAdd(&sh.trace_n, 1u); if (k_ < 8192u) { cs.stamps[128 + 2 * k_] = __builtin_amdgcn_s_memtime(); \
        cs.stamps[129 + 2 * k_] = ((unsigned long long)(code) << 48) | ((unsigned long long)(unsigned)(it_) << 8) | (unsigned)(c_); } } } while (0)
#else
#define PTRACE(code, it_, c_) do { } while (0)
#endif
#ifdef HTM_STAMPS
#define PSTAMP(k) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); st_acc[k] += n_ - t_last; t_last = n_; } while (0)
#define PCOUNT(k) do { st_acc[k] += 1; } while (0)
#else
#define PSTAMP(k) do { } while (0)
#define PCOUNT(k) do { } while (0)
#endif

constexpr unsigned long long kPipeWaitTicks = 500000000ull;      // 5 s of the 100 MHz clock: every wait is a fail-stop

// ------------------------------------------------------------------------------------------------------------------
// F: the front
// ------------------------------------------------------------------------------------------------------------------
struct PipeFront {
    int it, B, epoch;             // next iteration to produce, where it starts, the epoch it is produced in
    int fill, lo;                 // the LDS window holds stream positions [lo, fill)
    int k1, e1, k2, e2;           // per lane: kind and event of the chain's step one / two iterations before `it` (2 = unknown)
    int p0, p1, p2;               // per lane: the chain's full-evaluation steps whose orders have not been sent (iterations, oldest first; 0 = none)
    unsigned long long jobs;      // orders ever sent by this chain set (F is the only sender in a launch): the granule tags
    // the swap looked up where the stream predicts the end of iteration `it`'s chain steps (pipe_front): used if that is where they end
    int sE, si1, si2, snd;
    double ssr, sslr;
};

// start of the chain step n steps after the one that starts at p0 (per lane n; every lane runs the same number of rounds)
__device__ __forceinline__ int pipe_hops(const Ring &rg, int p0, int n, int rounds)
{
    int p = p0;
    for (int r = 0; r < rounds; ++r) {
        if (n > kHops) { p += rg.hop[(p & rg.mask) * kHops + kHops - 1]; n -= kHops; }
    }
    if (n > 0) p += rg.hop[(p & rg.mask) * kHops + n - 1];
    return p;
}

// the LDS window of the stream rings covers [.., target) (F only; 64 positions per round)
__device__ __forceinline__ void pipe_window(CsRef cs, PipeShared &sh, const Ring &rg, PipeFront &F, int target, int lane)
{
    if (target > sh.avail) target = sh.avail;
    while (F.fill < target) {
        const int to = min(F.fill + 64, target);
        PfRegs pf;
        pf_load(pf, cs, sh, F.fill + lane, to);
        pf_store(pf, rg);
        F.fill = to;
    }
    F.lo = max(F.lo, F.fill - (rg.mask + 1));
}

// the orders of the coming full evaluations (lanes <-> chains): the chain's oldest unsent one goes out when the chain's state
// is final and in memory up to `ahead` steps before it -> eight tagged granules into the chain's order slot (htm_step.hpp,
// worker_body), and the request to the collector wave (col_w)
__device__ __forceinline__ void pipe_orders(CsRef cs_, PipeShared &sh, const PipeRings &pr, PipeFront &F, int lane, unsigned long long launch)
{
    CsRef cs = rebase(cs_);
    const int nc = cs.n_chains;
    const bool in = lane < nc;
    const int c = in ? lane : 0;
    if (__ballot(in && F.p0 != 0) == 0ull) return;
    const int landed = lds_ld(&sh.landed_it);
    bool send = false;
    if (in && F.p0 != 0) {
        const int kd = pr.I(F.p0 & (kPipeSlots - 1), PI_KIND, c);
        send = landed >= F.p0 - 1 - ((kd >> 4) & 15) && lds_ld(&sh.col_w[c]) == 0;
    }
    unsigned long long m = __ballot(send);
    while (m) {
        const int cc = __ffsll((long long)m) - 1;
        m &= m - 1;
        const int jit = rl_i32(F.p0, cc), s = jit & (kPipeSlots - 1);
        const int type = pr.I(s, PI_TYPE, cc), idx = pr.I(s, PI_IDX, cc), kdc = pr.I(s, PI_KIND, cc);
        const double xn = pr.D(s, PD_XNEW, cc);
        // the hypocentre steps before it whose events the workers leave out: element and proposed value of each
        const int s1 = (jit - 1) & (kPipeSlots - 1), s2 = (jit - 2) & (kPipeSlots - 1);
        const int o1 = pr.I(s1, PI_O, cc), o2 = pr.I(s2, PI_O, cc);
        const double x1 = pr.D(s1, PD_XNEW, cc), x2 = pr.D(s2, PD_XNEW, cc);
        F.jobs += 1ull;
        unsigned long long tk = F.jobs & 0x7fffffffull;
        if (tk == 0) tk = 0x7fffffffull;
        if (lane == 0) {
            sh.col_tag[cc] = (unsigned)tk;
            lds_st(&sh.col_w[cc], pipe_tag(F.epoch, jit));
        }
        PTRACE(2, jit, cc);
        const unsigned tag = (unsigned)tk;
        if (lane < cs.slot_rep * kGranPerSlot) {
            const int gi = lane & 7;
            const unsigned long long xb = (unsigned long long)__double_as_longlong(xn);
            const unsigned pay = gi == 0 ? (unsigned)launch
                               : gi == 1 ? (0x80000000u | (unsigned)type | ((unsigned)idx << 3))
                               : gi == 2 ? (unsigned)(xb >> 32) : gi == 3 ? (unsigned)xb
                               : gi == 4 ? 0xffffffffu                                              // no commit to wait for: the state has landed
                               : gi == 5 ? (((kdc >> 9) & 1) ? (unsigned)o2 + 1u : 0u)              // element of the step two before (+1; 0 = none): its event is left out
                               : gi == 6 ? 0u
                               : (((kdc >> 8) & 1) ? (unsigned)o1 + 1u : 0u);                       // element of the step before
            st_gran(cs.slots + (size_t)(lane >> 3) * cs.slot_stride + cc * kGranPerSlot + gi, tag, pay);
        } else if (lane >= 32 && lane < 36) {
            // (the values those steps propose: tagged granules beside the order; the wave that holds the event waits for the tag)
            const unsigned long long b1 = (unsigned long long)__double_as_longlong(x1), b2 = (unsigned long long)__double_as_longlong(x2);
            const int gi = lane - 32;
            st_gran(cs.lo_gran + (size_t)cc * 16 + gi, tag, gi == 0 ? (unsigned)(b1 >> 32) : gi == 1 ? (unsigned)b1 : gi == 2 ? (unsigned)(b2 >> 32) : (unsigned)b2);
        }
    }
    if (send) { F.p0 = F.p1; F.p1 = F.p2; F.p2 = 0; }
}

// F adopts a new epoch: production restarts at the flushed iteration, every order on the books is taken back
__device__ __forceinline__ void pipe_front_adopt(CsRef cs_, PipeShared &sh, const PipeRings &pr, PipeFront &F, int e, int lane)
{
    CsRef cs = rebase(cs_);
    const int nc = cs.n_chains;
    F.epoch = e;
    F.it = lds_ld(&sh.fl_it); F.B = lds_ld(&sh.fl_base);
    // The restart position lies BEHIND where F was: the collector, which extends the window, may be overwriting exactly those
    // ring positions (it took them for dead).  It is told, and F reads the window again only when it has answered -- nothing
    // checks a decoded proposal against the stream, so F must never read a position that is being replaced.
    if (lane == 0) { lds_st(&sh.win_lo, F.B); lds_st(&sh.win_seq, lds_ld(&sh.win_seq) + 1); }
    while (lds_ld(&sh.win_ack) != lds_ld(&sh.win_seq)) {
        if (lds_ld(&sh.quit) != 0 || sh.c.err != 0) break;
        __builtin_amdgcn_s_sleep(2);
    }
    const bool in = lane < nc;
    const int c = in ? lane : 0;
    // kinds and events of the two steps before: decided iterations, their slots are intact
    F.k1 = 2; F.k2 = 2; F.e1 = -1; F.e2 = -1;
    if (in && F.it - 1 > sh.i0) { const int s = (F.it - 1) & (kPipeSlots - 1); F.k1 = pr.I(s, PI_KIND, c) & 15; F.e1 = pr.I(s, PI_EVT, c) - 1; }
    if (in && F.it - 2 > sh.i0) { const int s = (F.it - 2) & (kPipeSlots - 1); F.k2 = pr.I(s, PI_KIND, c) & 15; F.e2 = pr.I(s, PI_EVT, c) - 1; }
    F.p0 = 0; F.p1 = 0; F.p2 = 0; F.sE = -1;
    // (an order of the old epoch: the workers are told; the collector drops it when it sees the epoch)
    if (in && lds_ld(&sh.col_w[c]) != 0) { void_slot(cs, c); atomicExch(&sh.col_w[c], 0); }
    if (lane == 0) lds_st(&sh.f_stop_it, 0);           // (positions are predicted afresh)
}

// What F holds of one chain step between requesting its inputs and using them (per lane <-> chain): the decoded proposal found
// at the step's start position and the loads in flight -- the element it perturbs, the event's coordinates, the prior record.
struct PipeLoads {
    int P, type, idx, evt, o, gnx, snap, decw;
    double g, r, logr, x_old, hx, hy, hz;
    f64x4 pv;
};

// decode the step that starts at L.P and request its inputs; nothing is waited for here
__device__ __forceinline__ void pipe_issue(CsRef cs_, PipeShared &sh, const Ring &rg, PipeLoads &L, int c, bool act)
{
    CsRef cs = rebase(cs_);
    if (act) {
        const int nc = cs.n_chains, S_ = cs.S, nh = 3 * cs.E, M = rg.mask;
        const int off_tc = nc, off_qs = nc + nc * S_, off_ac = 2 * nc + nc * S_, off_hy = 2 * nc + 2 * nc * S_;
        const i32x4 dec = reinterpret_cast<const i32x4 *>(rg.dec)[L.P & M];
        L.g = rg.pg[L.P & M]; L.r = rg.pr[L.P & M]; L.logr = rg.plogr[L.P & M];
        L.type = dec.x; L.idx = dec.y; L.evt = dec.z; L.decw = dec.w;
        const int goff = L.type == 1 ? 0 : L.type == 2 ? off_tc : L.type == 3 ? off_qs : L.type == 4 ? off_ac : off_hy;
        L.gnx = (L.type == 1 || L.type == 3) ? 1 : (L.type == 2 || L.type == 4) ? S_ : nh;
        L.o = goff + c * L.gnx + L.idx;
        // the chain's commits that have landed: what the loads below are sure to see (D checks everything after them)
        L.snap = lds_ld(&sh.cland[c]);
        const double *xall = cs.xall;
        L.x_old = ld_agent(xall + L.o);
        L.hx = 0.0; L.hy = 0.0; L.hz = 0.0;
        if (L.type >= 5) {
            const int o_h = off_hy + c * nh + 3 * (L.evt - 1);
            L.hx = ld_agent(xall + o_h); L.hy = ld_agent(xall + o_h + 1); L.hz = ld_agent(xall + o_h + 2);
        }
        L.pv = *reinterpret_cast<const f64x4 *>(cs.prior + (cs.prior_same ? L.o - c * L.gnx : L.o));
    }
}

// judge_swap's draw of the swap found at E (cls_parallel.f90:163, :292-299) as this rank's stream has it: a single rank draws it
// right after the pair; a lock-step rank PEEKS at the draw it would take if the pair's first chain were its own (rank 0
// knows: then it is the last of the nd draws) -- what goes into the rank's swap record (htm_step.hpp, exchange_post)
template <bool LOCK>
__device__ __forceinline__ void pipe_judge_draw(CsRef cs, const Ring &rg, int E, int i1, int nd, double &sr, double &slr)
{
    sr = 0.0; slr = 0.0;
    if constexpr (LOCK) {
        const int own = (cs.rank == 0 && i1 >= 0 && i1 / cs.n_chains == 0) ? 1 : 0;
        const int jp = E + nd - own;
        sr = rg.U[jp & rg.mask]; slr = rg.LOGU[jp & rg.mask];
    } else if (nd > 0) {
        sr = rg.U[(E + nd - 1) & rg.mask]; slr = rg.LOGU[(E + nd - 1) & rg.mask];
    }
}

// one iteration's proposal records from the inputs requested for it (lanes <-> chains); returns where the next iteration starts
template <bool LOCK>
__device__ __forceinline__ int pipe_finish(CsRef cs_, PipeShared &sh, const Ring &rg, const PipeRings &pr, PipeFront &F, PipeLoads &L, int lane)
{
    CsRef cs = rebase(cs_);
    const int nc = cs.n_chains;
    const bool in = lane < nc;
    const int c = in ? lane : 0;
    const int it = F.it, B = F.B;
    const int rounds = (nc - 1) / kHops;
    int cnt = 0, ok = 1;
    double x_new = 0.0, lpr = 0.0;
    int from = 0, E_end = B;
    for (;;) {
        if (in && lane >= from) {
            const double mu = L.pv[0], rs2 = L.pv[1], step = L.pv[2];
            const int ptype = __double2loint(L.pv[3]);
            x_new = L.x_old + L.g * step;                               // cls_model.f90:172
            const double da = x_new - mu, db = L.x_old - mu;
            lpr = -(da * da - db * db) * rs2;                           // :175-177
            ok = 1;
            if (ptype == 1) {                                           // :178-187
                if (x_new <= mu) { lpr = (double)-1.0e+30f; ok = 0; }
                else lpr = lpr + htm_log(x_new - mu) - htm_log(L.x_old - mu);
            }
            cnt = L.decw - 1 + ok;                                      // the judge draw happens only if prior_ok (cls_mcmc.f90:193)
        }
        const int incl = wave_incl_scan(in ? cnt : 0);
        const int Pexp = B + incl - cnt;
        E_end = rl_i32(incl, 63) + B;
        const unsigned long long bad = __ballot(in && Pexp != L.P);
        if (bad == 0ull) break;
        // a rejected prior made a step one draw shorter (or the inputs were requested at a predicted start that did not hold):
        // the steps from the first one that starts elsewhere on are looked up again
        const int cb = __ffsll((long long)bad) - 1;
        const int Pcb = rl_i32(Pexp, cb);
        if (lane >= cb) L.P = pipe_hops(rg, Pcb, lane - cb, rounds);
        pipe_issue(cs, sh, rg, L, c, in && lane >= cb);
        from = cb;
    }
    // the swap that follows the chain steps (cls_parallel.f90:226-230, :163): looked up ahead if they end where the stream predicted
    int i1 = F.si1, i2 = F.si2, nd = F.snd;
    double sr = F.ssr, slr = F.sslr;
    if (__builtin_expect(E_end != F.sE, 0)) {
        flow_swap_at(cs, sh, rg, E_end, 1 << 30, i1, i2, nd);
        pipe_judge_draw<LOCK>(cs, rg, E_end, i1, nd, sr, slr);
    }
    const int s = it & (kPipeSlots - 1);
    const int kind = !ok ? 0 : (L.evt > 0 && it > 1) ? 1 : 2;           // hypo_tremor_mcmc.f90:246
    // how early the order of a full evaluation may go out, and which events the workers leave out (a function of the
    // stream and the state alone: see the header)
    int ahead = F.k1 == 2 ? 0 : F.k2 == 2 ? 1 : 2;
    int lo1 = (ahead >= 1 && F.k1 == 1) ? F.e1 : -1;
    int lo2 = (ahead >= 2 && F.k2 == 1) ? F.e2 : -1;
    if (lo1 >= 0 && lo1 == lo2) { ahead = 1; lo2 = -1; }
    if (in) {
        pr.I(s, PI_POS, c) = L.P; pr.I(s, PI_TYPE, c) = L.type; pr.I(s, PI_IDX, c) = L.idx; pr.I(s, PI_EVT, c) = L.evt;
        pr.I(s, PI_O, c) = L.o; pr.I(s, PI_CNT, c) = cnt; pr.I(s, PI_SEQ, c) = L.snap;
        pr.I(s, PI_KIND, c) = kind | (ahead << 4) | (lo1 >= 0 ? 256 : 0) | (lo2 >= 0 ? 512 : 0);
        pr.I(s, PI_LO1, c) = lo1; pr.I(s, PI_LO2, c) = lo2;
        pr.D(s, PD_XNEW, c) = x_new; pr.D(s, PD_LPR, c) = lpr; pr.D(s, PD_R, c) = L.r; pr.D(s, PD_LOGR, c) = L.logr;
        pr.D(s, PD_HX, c) = L.hx; pr.D(s, PD_HY, c) = L.hy; pr.D(s, PD_HZ, c) = L.hz;
    }
    if (lane == 0) {
        PipeHdr &h = sh.hdr[s];
        h.base = B; h.E = E_end; h.i1 = i1; h.i2 = i2; h.nd = nd;
        h.sr = sr; h.slr = slr;
        lds_st(&h.tag, pipe_tag(F.epoch, it));
        lds_st(&sh.f_it, it);
    }
    PTRACE(1, it, 0);
    // (a step whose prior rejected has nothing to evaluate: its result tag comes from here, after the slot's own)
    if (in && kind == 0) lds_st(&pr.EI(F.epoch & 1, s, EI_TAG, c), pipe_tag(F.epoch, it));
    if (in && kind == 2) { if (F.p0 == 0) F.p0 = it; else if (F.p1 == 0) F.p1 = it; else F.p2 = it; }
    F.k2 = F.k1; F.e2 = F.e1; F.k1 = kind; F.e1 = L.evt - 1;
    F.it = it + 1; F.B = E_end + nd;
    return F.B;
}

// F's loop is software-pipelined over iterations: the inputs of iteration it + 1 are requested -- at the positions the stream
// predicts when every prior of iteration it is ok -- BEFORE the inputs of iteration it are used, so the memory round trip of
// an iteration (~1 us: agent-scope loads) runs under the arithmetic of the one before; the LDS window of the stream rings is
// extended by a round of loads that fly under the same work.  A prediction that does not hold (a Rayleigh rejection) costs
// the look-ups of the steps behind it once more.
template <bool LOCK>
__device__ __forceinline__ void pipe_front(CsRef cs_, PipeShared &sh, const Ring &rg, const PipeRings &pr, int lane, unsigned long long launch)
{
    CsRef cs = rebase(cs_);
    const int nc = cs.n_chains, wd = 6 * nc + 16, rounds = (nc - 1) / kHops;
    const bool in = lane < nc;
    const int c = in ? lane : 0;
    PipeFront F;
    F.it = sh.i0 + 1; F.B = 0; F.epoch = 0; F.fill = sh.fill; F.lo = 0;
    F.k1 = 2; F.k2 = 2; F.e1 = -1; F.e2 = -1; F.p0 = 0; F.p1 = 0; F.p2 = 0;
    F.jobs = sh.c.jobs_total;
    F.sE = -1; F.si1 = -1; F.si2 = -1; F.snd = 0; F.ssr = 0.0; F.sslr = 0.0;
    int nE = -1, ni1 = -1, ni2 = -1, nnd = 0;        // the same for the iteration after (whose inputs are in L1)
    double nsr = 0.0, nslr = 0.0;
    PipeLoads L0, L1;
    L0.P = 0; L0.type = 5; L0.idx = 0; L0.evt = 1; L0.o = 0; L0.gnx = 1; L0.snap = 0; L0.decw = 6;
    L0.g = 0.0; L0.r = 0.0; L0.logr = 0.0; L0.x_old = 0.0; L0.hx = 0.0; L0.hy = 0.0; L0.hz = 0.0; L0.pv = f64x4{0.0, 1.0, 0.0, 0.0};
    L1 = L0;
    bool have = false;               // the current buffer holds the inputs of iteration F.it, requested at F.B
#ifdef HTM_STAMPS
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long t_last = __builtin_amdgcn_s_memtime();
#endif
    __builtin_amdgcn_s_setprio(1);
    // one pass of the loop with La = the current iteration's inputs, Lb = the buffer the next iteration's are requested into.
    // The two buffers swap roles after every published iteration (the loop below is unrolled by two): copying loaded values from
    // one to the other would wait for the loads just issued.  Returns 0: leave, 1: nothing published, 2: published.
    auto pass = [&](PipeLoads &La, PipeLoads &Lb) __attribute__((always_inline)) -> int {
        if (lds_ld(&sh.quit) != 0) return 0;
        const int e = lds_ld(&sh.epoch);
        if (e != F.epoch) { pipe_front_adopt(cs, sh, pr, F, e, lane); have = false; }
        // the stream window: kept three iterations ahead of F.B by the collector wave (positions before F.B are dead)
        F.fill = lds_ld(&sh.win_fill);
        if (lane == 0) { lds_st(&sh.win_lo, F.B); lds_st(&sh.win_want, min(F.B + 3 * wd + 96, sh.avail)); }
        const int dd = lds_ld(&sh.d_done);
        const int target = sh.c.iter_target;
        if (!(F.it <= dd + kPipeAhead && F.it <= target && lds_ld(&sh.f_stop_it) == 0 && __ballot(in && F.p2 != 0) == 0ull)) {
            pipe_orders(cs, sh, pr, F, lane, launch);
            // (not allowed to publish yet: the iteration's inputs are requested meanwhile)
            if (!have && F.it <= target && F.B + wd + 16 <= min(F.fill, sh.avail) && F.B >= F.fill - (rg.mask + 1)) {
                La.P = pipe_hops(rg, F.B, c, rounds);
                pipe_issue(cs, sh, rg, La, c, in);
                have = true; F.sE = -1;
            }
            __builtin_amdgcn_s_sleep(2);
            PSTAMP(1);
            return 1;
        }
        if (!have) {
            if (F.B + wd + 16 > sh.avail) { if (lane == 0) lds_st(&sh.f_stop_it, F.it); return 1; }      // the produced stream ends here: D stops the launch before it
            if (F.B + wd + 16 > F.fill || F.B < F.fill - (rg.mask + 1)) { __builtin_amdgcn_s_sleep(2); return 1; }      // (the window is on its way)
            La.P = pipe_hops(rg, F.B, c, rounds);
            pipe_issue(cs, sh, rg, La, c, in);
            have = true; F.sE = -1;
            PCOUNT(4);
        }
        // the next iteration, where the stream predicts it
        bool have1 = false;
        int Bp = -1;
        {
            // where this iteration's chain steps end if every prior is ok, and the swap found there
            if (F.sE < 0) {
                F.sE = pipe_hops(rg, F.B, nc, rounds);
                flow_swap_at(cs, sh, rg, F.sE, 1 << 30, F.si1, F.si2, F.snd);
                pipe_judge_draw<LOCK>(cs, rg, F.sE, F.si1, F.snd, F.ssr, F.sslr);
            }
            Bp = F.sE + F.snd;
            if (F.it + 1 <= target && Bp + wd + 16 <= min(F.fill, sh.avail) && F.B >= F.fill - (rg.mask + 1)) {
                Lb.P = pipe_hops(rg, Bp, c, rounds);
                pipe_issue(cs, sh, rg, Lb, c, in);
                have1 = true;
                // (and the same for the iteration after: where ITS steps end if every prior is ok)
                nE = pipe_hops(rg, Bp, nc, rounds);
                flow_swap_at(cs, sh, rg, nE, 1 << 30, ni1, ni2, nnd);
                pipe_judge_draw<LOCK>(cs, rg, nE, ni1, nnd, nsr, nslr);
            }
        }
        PSTAMP(5);
        pipe_orders(cs, sh, pr, F, lane, launch);
        PSTAMP(3);
        const int Bn = pipe_finish<LOCK>(cs, sh, rg, pr, F, La, lane);
        PSTAMP(6);
        if (have1 && Bn == Bp) { have = true; F.sE = nE; F.si1 = ni1; F.si2 = ni2; F.snd = nnd; F.ssr = nsr; F.sslr = nslr; }
        else { have = false; F.sE = -1; }
        PSTAMP(7); PCOUNT(2);
        pipe_orders(cs, sh, pr, F, lane, launch);
        PSTAMP(3);
        return 2;
    };
    for (int cur = 0;;) {
        const int r = cur == 0 ? pass(L0, L1) : pass(L1, L0);
        if (r == 0) break;
        if (r == 2) cur ^= 1;
    }
    if (lane == 0) sh.c.jobs_total = F.jobs;
#ifdef HTM_STAMPS
    if (lane == 0) for (int k = 0; k < 8; ++k) sh.stamp_acc[k] = st_acc[k];
#endif
}

// ------------------------------------------------------------------------------------------------------------------
// E: the evaluators
// ------------------------------------------------------------------------------------------------------------------
// the chain's station corrections, 1 / vs and pi f / (qs vs) as a consistent snapshot (sequence lock against D's commits);
// the station coordinates in `st` are the wave's own copy, loaded once
template <int NCH>
__device__ __forceinline__ int pipe_params(PipeShared &sh, const Ring &rg, int S, int nc, int c, int lane, StaRegs<NCH> &st, double &rb, double &ka, double &beta, double &q)
{
    const double *tc = rg.mx + nc + c * S, *ac = rg.mx + 2 * nc + nc * S + c * S;
    for (;;) {
        const int p1 = lds_ld(&sh.pver[c]);
        if (!(p1 & 1)) {
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                const int j = lane + 64 * k;
                st.tc[k] = j < S ? tc[j] : 0.0; st.ac[k] = j < S ? ac[j] : 0.0;
            }
            rb = sh.rbeta[c]; ka = sh.katt[c]; beta = rg.mx[c]; q = rg.mx[nc + nc * S + c];
            asm volatile("" ::: "memory");
            if (lds_ld(&sh.pver[c]) == p1) return p1;
        }
        __builtin_amdgcn_s_sleep(1);
    }
}
// a vs / qs / correction proposal applied to the snapshot (type 1 vs, 2 t_corr, 3 qs, 4 a_corr; as worker_body forms them)
template <int NCH>
__device__ __forceinline__ void pipe_apply(int type, int idx, double xn, int lane, StaRegs<NCH> &st, double &rb, double &ka, double beta, double q)
{
    if (type == 2 || type == 4) {
#pragma unroll
        for (int k = 0; k < NCH; ++k)
            if (lane + 64 * k == idx) { if (type == 2) st.tc[k] = xn; else st.ac[k] = xn; }
    } else if (type == 1 || type == 3) {
        const double b_ = type == 1 ? xn : beta, q_ = type == 3 ? xn : q;
        rb = 1.0 / b_; ka = (kPi * kFreq) / (q_ * b_);
    }
}

// A single-event task as an evaluator holds it: taken from the queue (te, it, c), then -- once its iteration's records are
// published -- looked at (`seen`), and if it is a single-event update its event's observation rows requested (`live`)
template <int N, bool F32>
struct PipeTask {
    int te, it, c;
    bool taken, seen, live;
    int idx, evt;
    double x_new, hx, hy, hz;
    ObsRegs<N, F32> ob;
};

template <int NCH, bool F32, bool LOCK>
__device__ __forceinline__ void pipe_evaluator(FwRef f_, CsRef cs_, PipeShared &sh, const Ring &rg, const PipeRings &pr,
                                               const double *s_sx, const double *s_sy, const double *s_sz, int lane)
{
    CsRef cs = rebase(cs_);
    FwRef f = rebase(f_);
    const int nc = cs.n_chains, S_ = cs.S, i0 = sh.i0;
    constexpr int N = NCH > 0 ? NCH : 1;
#ifdef HTM_STAMPS
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long t_last = __builtin_amdgcn_s_memtime();
    struct Flush_ { unsigned long long *a; PipeShared *s; int lane; __device__ ~Flush_() { if (lane == 0) for (int k = 0; k < 8; ++k) atomicAdd(&s->stamp_acc[16 + k], a[k]); } } flush_{st_acc, &sh, lane};
#endif
    // take the next task from the queue
    auto take = [&](PipeTask<N, F32> &T) __attribute__((always_inline)) {
        unsigned long long qv = 0;
        if (lane == 0) qv = atomicAdd(&sh.q, 1ull);
        T.te = __builtin_amdgcn_readfirstlane((int)(unsigned)(qv >> 32));
        const int t = __builtin_amdgcn_readfirstlane((int)(unsigned)qv);
        const int dit = t / nc;
        T.it = i0 + 1 + dit; T.c = t - dit * nc;
        T.taken = true; T.seen = false; T.live = false;
    };
    // look at a taken task if its iteration's records are there: a single-event update has its rows requested
    auto look = [&](PipeTask<N, F32> &T) __attribute__((always_inline)) {
        const int s = T.it & (kPipeSlots - 1);
        // (the tag and, in the same batch, what it covers: used only if the tag is the iteration's)
        const int tg = lds_ld(&sh.hdr[s].tag);
        const int kdv = pr.I(s, PI_KIND, T.c), idv = pr.I(s, PI_IDX, T.c), evv = pr.I(s, PI_EVT, T.c);
        const double xn = pr.D(s, PD_XNEW, T.c), hx = pr.D(s, PD_HX, T.c), hy = pr.D(s, PD_HY, T.c), hz = pr.D(s, PD_HZ, T.c);
        if (tg != pipe_tag(T.te, T.it)) return;
        T.seen = true;
        if ((uni(kdv) & 15) != 1) return;        // (prior rejected: nothing to evaluate; full evaluation: the collector's)
        T.idx = uni(idv); T.evt = uni(evv);
        T.x_new = xn; T.hx = hx; T.hy = hy; T.hz = hz;
        load_obs_regs<N, F32>(T.ob, f, T.evt - 1, lane);
        T.live = true;
    };
    StaRegs<N> st;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const int j = lane + 64 * k;
        st.sx[k] = j < S_ ? s_sx[j] : 0.0; st.sy[k] = j < S_ ? s_sy[j] : 0.0; st.sz[k] = j < S_ ? s_sz[j] : 0.0;
        st.tc[k] = 0.0; st.ac[k] = 0.0;
    }
    PipeTask<N, F32> cur, nxt;
    cur.taken = false; cur.seen = false; cur.live = false; cur.te = 0; cur.it = 0; cur.c = 0; cur.idx = 0; cur.evt = 1;
    cur.x_new = 0.0; cur.hx = 0.0; cur.hy = 0.0; cur.hz = 0.0;
#pragma unroll
    for (int k = 0; k < N; ++k) { cur.ob.tob[k] = 0; cur.ob.tpr[k] = 0; cur.ob.aob[k] = 0; cur.ob.apr[k] = 0; }
    cur.ob.rpst = 1.0; cur.ob.rpsa = 1.0;
    nxt = cur;
    for (;;) {
        if (lds_ld(&sh.quit) != 0 || sh.c.err != 0) return;
        PSTAMP(3);
        // ---- the current task: taken, its records published, a single-event update
        if (!cur.taken) take(cur);
        if (!cur.seen) {
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            for (unsigned spin = 0;; ++spin) {
                look(cur);
                if (cur.seen) break;
                if (lds_ld(&sh.epoch) != cur.te || lds_ld(&sh.quit) != 0) break;
                if ((spin & 63u) == 63u && __builtin_amdgcn_s_memrealtime() - t0 > kPipeWaitTicks) { if (lane == 0) sh.c.err = -14; return; }
                __builtin_amdgcn_s_sleep(2);
            }
        }
        PSTAMP(0);
        if (!cur.live) {                        // nothing to do (or a task of a flushed epoch): the next one
            if (nxt.taken) { cur = nxt; nxt.taken = false; } else cur.taken = false;
            continue;
        }
        // ---- the task after it, so that its rows fly under this evaluation
        if (!nxt.taken) take(nxt);
        if (!nxt.seen) look(nxt);
        // ---- single-event update (cls_forward.f90:307-362): old and proposed position of the event
        const int te = cur.te, it = cur.it, c = cur.c;
        PTRACE(3, it, c);
        const int s = it & (kPipeSlots - 1), want = pipe_tag(te, it), bank = te & 1;
        const int ev = cur.evt - 1, cmp = cur.idx - 3 * ev;
        double rb = 0.0, ka = 0.0, beta = 0.0, q = 0.0;
        int dep = 0, pv = 0, d_type = 0, d_idx = 0;
        double d_xn = 0.0;
        bool live = true;
        const double *tcp = rg.mx + nc + c * S_, *acp = rg.mx + 2 * nc + nc * S_ + c * S_;
        for (;;) {
            // Everything the evaluation needs besides the record, requested in ONE batch (LDS operations of a wave execute in
            // order, so the sequence lock's two reads bracket the data between them): the decisions' progress, the kinds of the
            // chain's steps of the iterations before (as far back as F can be ahead of the decisions), the parameter version, the
            // chain's corrections and its two reciprocals, the version again.
            const int dd = lds_ld(&sh.d_done);
            int kb[kPipeAhead - 1];
#pragma unroll
            for (int j = 0; j < kPipeAhead - 1; ++j) kb[j] = pr.I((it - 1 - j) & (kPipeSlots - 1), PI_KIND, c);
            const int p1 = lds_ld(&sh.pver[c]);
#pragma unroll
            for (int k = 0; k < N; ++k) {
                const int j = lane + 64 * k;
                st.tc[k] = j < S_ ? tcp[j] : 0.0; st.ac[k] = j < S_ ? acp[j] : 0.0;
            }
            rb = sh.rbeta[c]; ka = sh.katt[c]; beta = rg.mx[c]; q = rg.mx[nc + nc * S_ + c];
            asm volatile("" ::: "memory");
            const int p2 = lds_ld(&sh.pver[c]);
            // undecided full-evaluation steps of this chain before this one: none -> one evaluation; one -> a second one
            // under its proposal; more -> wait
            unsigned fullm = 0;
#pragma unroll
            for (int j = 0; j < kPipeAhead - 1; ++j) fullm |= ((uni(kb[j]) & 15) == 2) ? (1u << j) : 0u;
            const int nund = it - 1 - dd;                                   // iterations it - 1 .. dd + 1 are undecided
            const unsigned um = nund <= 0 ? 0u : (fullm & ((nund >= 32 ? ~0u : ((1u << nund) - 1u))));
            const int ndep = __popc(um);
            dep = um ? it - 1 - (__ffs((int)um) - 1) : 0;
            pv = p1;
            if (ndep <= 1 && !(p1 & 1) && p1 == p2) {
                if (dep == 0) break;
                // (that step's proposal, read while the step is still undecided -- its slot cannot have been taken over --: checked by
                // reading the decisions' progress again)
                const int sd = dep & (kPipeSlots - 1);
                d_type = uni(pr.I(sd, PI_TYPE, c)); d_idx = uni(pr.I(sd, PI_IDX, c)); d_xn = pr.D(sd, PD_XNEW, c);
                if (lds_ld(&sh.d_done) == dd) break;
            }
            if (lds_ld(&sh.epoch) != te || lds_ld(&sh.quit) != 0) { live = false; break; }
            __builtin_amdgcn_s_sleep(2);
        }
        if (live) {
            const double px[2] = {cur.hx, cmp == 0 ? cur.x_new : cur.hx};
            const double py[2] = {cur.hy, cmp == 1 ? cur.x_new : cur.hy};
            const double pz[2] = {cur.hz, cmp == 2 ? cur.x_new : cur.hz};
            double out[2];
            event_misfit<N, 2, F32, true>(f, cur.ob, lane, st, px, py, pz, rb, ka, out);
            const double dA = wave_sum1(out[0] - out[1]);
            double dB = 0.0;
            if (dep) {
                pipe_apply<N>(d_type, d_idx, d_xn, lane, st, rb, ka, beta, q);
                event_misfit<N, 2, F32, true>(f, cur.ob, lane, st, px, py, pz, rb, ka, out);
                dB = wave_sum1(out[0] - out[1]);
            }
            // (flushed meanwhile: the record may be another's)
            if (lds_ld(&sh.hdr[s].tag) == want && lds_ld(&sh.epoch) == te && lane == 0) {
                pr.ED(bank, s, 0, c) = dA; pr.ED(bank, s, 1, c) = dB;
                pr.EI(bank, s, EI_PV, c) = pv; pr.EI(bank, s, EI_DEP, c) = dep;
                lds_st(&pr.EI(bank, s, EI_TAG, c), want);
            }
            PSTAMP(1); PCOUNT(4); if (dep) PCOUNT(6);
            PTRACE(4, it, c);
        }
        cur = nxt; nxt.taken = false;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// C: the collector of the full evaluations (cls_forward.f90:268-303), one wave.  F sends a step's order as soon as the chain's
// state is final `ahead` steps before it and posts it here (col_w).  The workers leave the events of those steps out of their
// partial sums and report them on their own at both candidate positions (worker_body, `leftout`); C watches the workers' tagged
// granules -- of up to four orders at a time -- and publishes the sums D decides from.
// ------------------------------------------------------------------------------------------------------------------
template <int NCH, bool F32, bool LOCK>
__device__ __forceinline__ void pipe_collector(FwRef f_, CsRef cs_, PipeShared &sh, const Ring &rg, const PipeRings &pr,
                                               const double *s_sx, const double *s_sy, const double *s_sz, int lane)
{
    CsRef cs = rebase(cs_);
    const int nc = cs.n_chains;
    const bool in = lane < nc;
    const int c = in ? lane : 0;
    int w_fill = sh.fill, w_ack = 0;     // the window holds the stream positions [w_fill - ring, w_fill)
    int seen_w = 0;                      // per lane: the order of the chain C has started to watch
    unsigned t_seen = 0;                 // per lane: since when (100 MHz clock, low word)
#ifdef HTM_STAMPS
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long t_last = __builtin_amdgcn_s_memtime();
#endif
    (void)f_; (void)s_sx; (void)s_sy; (void)s_sz;
    for (;;) {
        if (lds_ld(&sh.quit) != 0 || sh.c.err != 0) break;
        // ---- the LDS window of the stream rings, on F's request: 64 positions per round, never over positions F still reads
        {
            const int sq = lds_ld(&sh.win_seq);
            const int lo = lds_ld(&sh.win_lo), want = lds_ld(&sh.win_want);
            // (F went back -- a flush -- behind what the ring still holds: from there again; F sees that it is not covered)
            if (lo < w_fill - (rg.mask + 1)) { w_fill = lo; if (lane == 0) lds_st(&sh.win_fill, w_fill); }
            const int to = min(min(w_fill + 64, want), lo + rg.mask + 1 - 8);
            if (to > w_fill) {
                PfRegs pf;
                pf_load(pf, cs, sh, w_fill + lane, to);
                pf_store(pf, rg);
                w_fill = to;
                if (lane == 0) lds_st(&sh.win_fill, w_fill);
            } else if (lds_ld(&sh.win_fill) != w_fill && lane == 0) lds_st(&sh.win_fill, w_fill);
            // (the window as published now respects the win_lo read above: F may go on)
            if (sq != w_ack) { w_ack = sq; if (lane == 0) lds_st(&sh.win_ack, sq); }
        }
        const int w = in ? lds_ld(&sh.col_w[c]) : 0;
        const int ep = lds_ld(&sh.epoch);
        if (w != seen_w) { seen_w = w; t_seen = (unsigned)__builtin_amdgcn_s_memrealtime(); }
        const int stg_w = w;
        // ---- the workers' partial sums (and their sums of the left-out events) of up to four orders, requested together
        unsigned long long m2 = __ballot(w != 0 && ((unsigned)w >> 24) == ((unsigned)ep & 0xffu));
        PSTAMP(0);
        if (m2 == 0ull) { __builtin_amdgcn_s_sleep(4); PSTAMP(3); continue; }
        constexpr int kSweep = 4, kOrd = 2;       // <= 256 workers (host-checked)
        unsigned long long hi[kOrd][kSweep], lo[kOrd][kSweep], lg[kOrd];
        int oc[kOrd];
        unsigned otag[kOrd];
        const int n_wg = cs.n_wg, pgs = cs.pgran_stride;
#pragma unroll
        for (int k = 0; k < kOrd; ++k) {
            oc[k] = -1; otag[k] = 0; lg[k] = 0;
            if (m2) {
                oc[k] = __ffsll((long long)m2) - 1;
                m2 &= m2 - 1;
                otag[k] = (unsigned)uni((int)sh.col_tag[oc[k]]);
                const unsigned long long *pg = cs.pgran + (size_t)oc[k] * n_wg * pgs;
#pragma unroll
                for (int j = 0; j < kSweep; ++j) {
                    const int kk = j * 64 + lane;
                    hi[k][j] = 0; lo[k][j] = 0;
                    if (kk < n_wg) { hi[k][j] = ld_agent(pg + (size_t)pgs * kk); lo[k][j] = ld_agent(pg + (size_t)pgs * kk + 1); }
                }
                if (lane < 8) lg[k] = ld_agent(cs.lo_gran + (size_t)oc[k] * 16 + 8 + lane);
            }
        }
#pragma unroll
        for (int k = 0; k < kOrd; ++k) {
            if (oc[k] < 0) continue;
            const int wc = rl_i32(w, oc[k]);
            const int it = wc & 0xffffff, s = it & (kPipeSlots - 1), bank = (wc >> 24) & 1;
            const int kd = uni(pr.I(s, PI_KIND, oc[k]));
            bool got = true;
#pragma unroll
            for (int j = 0; j < kSweep; ++j)
                if (j * 64 + lane < n_wg) got = got && (unsigned)(hi[k][j] >> 32) == otag[k] && (unsigned)(lo[k][j] >> 32) == otag[k];
            if ((lane < 4 && (kd & 256)) || (lane >= 4 && lane < 8 && (kd & 512))) got = got && (unsigned)(lg[k] >> 32) == otag[k];
            if (!__all(got)) continue;
            double part = 0.0;
#pragma unroll
            for (int j = 0; j < kSweep; ++j)            // fixed order: worker lane, lane + 64, ...
                if (j * 64 < n_wg) part += (j * 64 + lane < n_wg) ? gran_f64(hi[k][j], lo[k][j]) : 0.0;
            const double Sp = wave_sum1(part);
            const unsigned llo = (unsigned)lg[k];
            unsigned gl[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) gl[j] = (unsigned)rl_i32((int)llo, j);
            if (lane == 0) {
                // (the slot still holds this iteration in this epoch: else the order was flushed and nobody reads the sums)
                if (lds_ld(&sh.hdr[s].tag) == wc && lds_ld(&sh.epoch) == ep) {
                    pr.ED(bank, s, 0, oc[k]) = Sp;
                    pr.ED(bank, s, 1, oc[k]) = (kd & 256) ? __longlong_as_double((long long)(((unsigned long long)gl[0] << 32) | gl[1])) : 0.0;
                    pr.ED(bank, s, 2, oc[k]) = (kd & 256) ? __longlong_as_double((long long)(((unsigned long long)gl[2] << 32) | gl[3])) : 0.0;
                    pr.ED(bank, s, 3, oc[k]) = (kd & 512) ? __longlong_as_double((long long)(((unsigned long long)gl[4] << 32) | gl[5])) : 0.0;
                    pr.ED(bank, s, 4, oc[k]) = (kd & 512) ? __longlong_as_double((long long)(((unsigned long long)gl[6] << 32) | gl[7])) : 0.0;
                    pr.EI(bank, s, EI_PV, oc[k]) = 0; pr.EI(bank, s, EI_DEP, oc[k]) = 0;
                    lds_st(&pr.EI(bank, s, EI_TAG, oc[k]), wc);
                }
                atomicCAS(&sh.col_w[oc[k]], wc, 0);
            }
            PCOUNT(5);
            PTRACE(5, it, oc[k]);
        }
        // a fail-stop: an order that stays unanswered for 5 s
        {
            const bool late = w != 0 && w == stg_w && (unsigned)__builtin_amdgcn_s_memrealtime() - t_seen > 500000000u && lds_ld(&sh.col_w[c]) == w;
            const unsigned long long ml = __ballot(late);
            if (ml) {
                const int cc = __ffsll((long long)ml) - 1;
                if (lane == 0) {
                    sh.c.err = -8;
                    const int wc = rl_i32(w, cc);
                    unsigned long long *dg = cs.diag;          // what was waited for: the host puts it into its message
                    dg[1] = cc; dg[2] = sh.col_tag[cc]; dg[3] = 1; dg[4] = 1 + ((pr.I(wc & (kPipeSlots - 1), PI_KIND, cc) >> 4) & 15); dg[5] = wc & 0xffffff;
                    dg[6] = pr.I(wc & (kPipeSlots - 1), PI_POS, cc); dg[7] = pr.I(wc & (kPipeSlots - 1), PI_TYPE, cc); dg[8] = pr.I(wc & (kPipeSlots - 1), PI_IDX, cc);
                    dg[9] = 0; dg[10] = 0; dg[11] = 1; dg[12] = 0; dg[0] = 1;
                }
                break;
            }
        }
        PSTAMP(2);
    }
#ifdef HTM_STAMPS
    if (lane == 0) for (int k = 0; k < 8; ++k) sh.stamp_acc[24 + k] = st_acc[k];
#endif
}

// ------------------------------------------------------------------------------------------------------------------
// D: the decider
// ------------------------------------------------------------------------------------------------------------------
template <int NCH, bool F32, bool LOCK>
__device__ __forceinline__ void pipe_decider(FwRef f_, CsRef cs_, PipeShared &sh, const Ring &rg, const PipeRings &pr, double *s_gath, int lane, int wmax)
{
    CsRef cs = rebase(cs_);
    FwRef f = rebase(f_);
    const int nc = cs.n_chains, S_ = cs.S, nh = 3 * cs.E, i0 = sh.i0;
    const int off_qs = nc + nc * S_, off_hy = 2 * nc + 2 * nc * S_;
    const int n_all = cs.n_procs * nc;
    const bool in = lane < nc;
    const int c = in ? lane : 0;
    double T = sh.temp[c], rT = sh.rtemp[c], L = sh.L[c];
    int cseq = 0, pver = 0, ah = 0, nextB = 0, epoch = 0;
    int last_iter = sh.c.iter_target, stop_code = 0;
    const int n_int = cs.n_interval, n_burn = cs.n_burn;
    int rec_phase = (i0 + 1) % n_int;
    unsigned long long n_full = 0, n_part = 0;
    int n_lik = uni(sh.c.n_lik), n_smp = uni(sh.c.n_smp);      // (D is their only writer during the launch)
    const int slog_cap = uni(sh.c.slog_cap), slog_n0 = uni(sh.c.slog_n);
    const int cap_lik = cs.cap_lik, cap_smp = cs.cap_smp;
    const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    __builtin_amdgcn_s_setprio(2);
    int it = i0 + 1;
    bool dead = false;
#ifdef HTM_STAMPS
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long t_last = __builtin_amdgcn_s_memtime();
#endif
    while (it <= last_iter) {
        PSTAMP(3);
        // ---- the commits of the iteration before have landed: F may read them, orders may rely on them
        drain_vmem();
        if (in) sh.cland[c] = cseq;
        if (lane == 0) lds_st(&sh.landed_it, it - 1);
        const int s = it & (kPipeSlots - 1), want = pipe_tag(epoch, it), bank = epoch & 1;
        PTRACE(6, it, 0);
        // ---- the iteration's proposal records and their evaluations: every chain's result tag (F sets it for a step that has
        // ---- nothing to evaluate; an evaluator or the collector writes it after having seen the slot's tag)
        {
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            bool stop = false;
            for (unsigned spin = 0;; ++spin) {
                const int et = in ? lds_ld(&pr.EI(bank, s, EI_TAG, c)) : want;
                if (__all(et == want)) break;
                if (lds_ld(&sh.f_stop_it) == it && lds_ld(&sh.hdr[s].tag) != want) { stop = true; break; }
                if ((spin & 63u) == 63u) {
                    if (sh.c.err != 0) { dead = true; break; }
                    if (__builtin_amdgcn_s_memrealtime() - t0 > kPipeWaitTicks) { if (lane == 0) sh.c.err = lds_ld(&sh.hdr[s].tag) != want ? -13 : -12; dead = true; break; }
                }
                __builtin_amdgcn_s_sleep(1);
            }
            if (dead) break;
            if (stop) {
                if constexpr (LOCK) { if (lane == 0) sh.c.err = -7; dead = true; break; }      // (lock-step ranks leave a launch only together: the stop request comes first)
                last_iter = it - 1; stop_code = 2; break;
            }
        }
        PSTAMP(0);
        PTRACE(7, it, 0);
        const int kd = in ? pr.I(s, PI_KIND, c) : 0, kind = kd & 15;
        const int type = in ? pr.I(s, PI_TYPE, c) : 5, idx = pr.I(s, PI_IDX, c), evt = pr.I(s, PI_EVT, c), o = pr.I(s, PI_O, c);
        const int seq = pr.I(s, PI_SEQ, c);
        const bool need_e = in && kind != 0;
        PSTAMP(1);
        // ---- what F and E assumed: no commit of the chain since F looked touched the element (or the event); the evaluation was
        // ---- made under the parameters the chain has now
        bool bad = false;
        int a_dep = 0, dep = 0;
        {
            const int n_new = in ? cseq - seq : 0;
            bad = n_new > kPipeLog;
            const int o_h = o - idx + 3 * (evt - 1);
            for (int j = 0; j < kPipeLog; ++j) {
                if (__ballot(j < n_new) == 0ull) break;
                if (j < n_new) {
                    const int oj = sh.clog[c][(seq + j) & (kPipeLog - 1)];
                    bad = bad || oj == o || (kind == 1 && (unsigned)(oj - o_h) < 3u);
                }
            }
            if (need_e && kind == 1) {                     // (a full evaluation's order went out after the last step that could change the parameters)
                const int pv = pr.EI(bank, s, EI_PV, c);
                dep = pr.EI(bank, s, EI_DEP, c);
                a_dep = dep ? ((ah >> (it - dep - 1)) & 1) : 0;
                bad = bad || (pv + 2 * a_dep != pver);
            }
        }
        if (__any(bad)) {
            // ---- flush: a new epoch from this iteration on (state final and landed: F reads it as it is)
            epoch += 1;
            if (lane == 0) {
                sh.n_flush += 1;
                lds_st(&sh.fl_it, it); lds_st(&sh.fl_base, nextB);
                lds_st(&sh.epoch, epoch);
                __hip_atomic_exchange(&sh.q, ((unsigned long long)(unsigned)epoch << 32) | (unsigned)((it - i0 - 1) * nc), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            PCOUNT(5);
            continue;
        }
        // ---- the temperature of this iteration: the swap of the iteration before (cls_parallel.f90:121-136, :285-302)
        if constexpr (LOCK) {
            // a lock-step rank: decided from ALL ranks' records (htm_step.hpp, exchange_finish; posted at the end of the iteration
            // before: their round trip ran under the wait and the checks above).  It may end the job (a rank asked everybody to
            // stop, or failed) or move this rank's stream (the judge draw was this rank's after all: cls_parallel.f90:163)
            if (it - 1 > i0) {
                exchange_finish(cs, sh, s_gath, it - 1, lane, false);
                if (sh.c.err != 0) { dead = true; break; }
                if (in) { T = sh.temp[c]; rT = sh.rtemp[c]; }
                if (sh.c.stop != 0) break;                                 // everybody leaves after iteration it - 1
                const int base = (int)(sh.c.spos - sh.origin);             // where this iteration really starts
                if (base != nextB) {
                    nextB = base;
                    epoch += 1;
                    if (lane == 0) {
                        sh.n_flush += 1;
                        lds_st(&sh.fl_it, it); lds_st(&sh.fl_base, nextB);
                        lds_st(&sh.epoch, epoch);
                        __hip_atomic_exchange(&sh.q, ((unsigned long long)(unsigned)epoch << 32) | (unsigned)((it - i0 - 1) * nc), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                    continue;
                }
            }
        } else
        if (it - 1 > i0 && n_all > 1) {
            const PipeHdr &hp = sh.hdr[(it - 1) & (kPipeSlots - 1)];
            const int i1 = uni(hp.i1), i2 = uni(hp.i2);
            const double sr = hp.sr, slr = hp.slr;
            if (i1 >= 0) {
                const double T1 = rl_f64(T, i1), T2 = rl_f64(T, i2), rT1 = rl_f64(rT, i1), rT2 = rl_f64(rT, i2);
                const double del_s = (rl_f64(L, i2) - rl_f64(L, i1)) * (rT1 - rT2);      // :292
                if (sr >= kEps && slr <= del_s) {                                         // :131-136
                    if (lane == i1) { T = T2; rT = rT2; }
                    if (lane == i2) { T = T1; rT = rT1; }
                }
            }
        }
        // ---- Metropolis (cls_mcmc.f90:193-203)
        const double x_new = pr.D(s, PD_XNEW, c), lpr = pr.D(s, PD_LPR, c), r = pr.D(s, PD_R, c), logr = pr.D(s, PD_LOGR, c);
        double L_new = 0.0;
        if (kind == 1) {
            const double d0 = pr.ED(bank, s, 0, c), d1 = pr.ED(bank, s, 1, c);
            L_new = L + ((dep && a_dep) ? d1 : d0);
        } else if (kind == 2) {
            double Sm = pr.ED(bank, s, 0, c);
            if (kd & 256) Sm = Sm + ((ah & 1) ? pr.ED(bank, s, 2, c) : pr.ED(bank, s, 1, c));
            if (kd & 512) Sm = Sm + ((ah & 2) ? pr.ED(bank, s, 4, c) : pr.ED(bank, s, 3, c));
            L_new = -Sm - f.const_sum;                                   // cls_forward.f90:277-300
        }
        const int ok = kind != 0 ? 1 : 0;
        const int acc = (in && ok && metropolis(L_new, L, rT, lpr, r, logr)) ? 1 : 0;
        const int cool = (T < 1.0 + kEps) ? 1 : 0;
        // ---- commit (cls_mcmc.f90:186-189, :207-219)
        if (in) {
            if (cool) sh.np[c * 7 + type - 1] += 1;
            if (acc) {
                st_agent(cs.xall + o, x_new);
                if (o < off_hy) {
                    // vs, qs or a station correction: what the evaluators read, under the sequence lock
                    lds_st(&sh.pver[c], pver + 1);
                    rg.mx[o] = x_new;
                    if (type == 1 || type == 3) {
                        const double b_ = rg.mx[c], q_ = rg.mx[off_qs + c];
                        sh.rbeta[c] = 1.0 / b_; sh.katt[c] = (kPi * kFreq) / (q_ * b_);
                    }
                    pver += 2;
                    lds_st(&sh.pver[c], pver);
                }
                sh.clog[c][cseq & (kPipeLog - 1)] = o;
                cseq += 1;
                L = L_new;
                if (cool) sh.na[c * 7 + type - 1] += 1;
            }
        }
        n_full += (unsigned long long)__popcll(__ballot(in && kind == 2));
        n_part += (unsigned long long)__popcll(__ballot(in && kind == 1));
        ah = (ah << 1) | acc;
        nextB = uni(sh.hdr[s].E) + uni(sh.hdr[s].nd);
        // ---- records of this iteration (hypo_tremor_mcmc.f90:270-280)
        if (__builtin_expect(slog_cap > 0, 0)) {
            const int row = slog_n0 + (it - i0 - 1) * nc + c;
            if (in && row < slog_cap) {
                int32_t *ir = cs.slog_i + 8 * (size_t)row;
                double *dr = cs.slog_d + 4 * (size_t)row;
                ir[0] = it; ir[1] = c; ir[2] = type; ir[3] = idx + 1; ir[4] = ok; ir[5] = acc; ir[6] = kind == 2 ? 1 : 0; ir[7] = 0;
                dr[0] = x_new; dr[1] = L_new; dr[2] = L; dr[3] = T;
            }
        }
        if (__builtin_expect(rec_phase == 1, 0)) {
            const bool rec_l = in && cool, rec_s = rec_l && it > n_burn;
            const unsigned long long ml = __ballot(rec_l), ms = __ballot(rec_s);
            const int sl = n_lik + __popcll(ml & below), ss0 = n_smp;
            if (rec_l && sl < cs.cap_lik) { cs.lik_iter[sl] = it; cs.lik_chain[sl] = c; cs.lik_val[sl] = L; }
            if (ms) {
                drain_vmem();                                            // (this iteration's commits are part of the samples)
                const int rec = nh + 2 * S_ + 2;
                const int vz = opaque_zero();
                unsigned long long m = ms;
                int k = 0;
                while (m) {
                    const int cc = __ffsll((long long)m) - 1;
                    m &= m - 1;
                    const int ss = ss0 + k; ++k;
                    if (ss < cs.cap_smp) {
                        double *dst = cs.smp_data + (size_t)ss * rec;
                        const double *hxp = cs.xall + off_hy + (size_t)cc * nh;
                        for (int j = lane; j < nh; j += 64) dst[j] = ld_agent(hxp + j + vz);
                        for (int j = lane; j < S_; j += 64) { dst[nh + j] = rg.mx[nc + cc * S_ + j]; dst[nh + S_ + j] = rg.mx[2 * nc + nc * S_ + cc * S_ + j]; }
                        if (lane == 0) { dst[nh + 2 * S_] = rg.mx[cc]; dst[nh + 2 * S_ + 1] = rg.mx[off_qs + cc]; cs.smp_iter[ss] = it; cs.smp_chain[ss] = cc; }
                    }
                }
            }
            n_lik += __popcll(ml); n_smp += __popcll(ms);
        }
        rec_phase = rec_phase + 1 == n_int ? 0 : rec_phase + 1;
        if constexpr (LOCK) {
            // this rank's swap record: pair (rank 0's select_pair), the pending judge_swap draw (peeked), (T, L) of all its chains,
            // and whether it asks everybody to stop after this iteration (record buffers or produced stream nearly used up)
            const PipeHdr &h = sh.hdr[s];
            const int pi1 = uni(h.i1), pi2 = uni(h.i2), pnd = uni(h.nd), pE = uni(h.E);
            const int own = (cs.rank == 0 && pi1 >= 0 && pi1 / nc == 0) ? 1 : 0;
            const int jp = pE + pnd - own;
            if (in) { sh.xrec[4 + 2 * c] = T; sh.xrec[5 + 2 * c] = L; }
            const bool my_stop = n_lik + 3 * nc > cap_lik || n_smp + 3 * nc > cap_smp || sh.avail < jp + 4 * wmax;
            if (lane == 0) {
                sh.xrec[0] = (double)pi1; sh.xrec[1] = (double)pi2; sh.xrec[2] = h.sr; sh.xrec[3] = (double)it;
                sh.c.swap_i1 = pi1; sh.c.swap_i2 = pi2; sh.c.swap_r = h.sr; sh.c.swap_logr = h.slr;
                sh.c.spos = sh.origin + jp;                       // RNG commit: draws consumed so far (apply_swap adds the judge draw if it is ours)
                sh.c.stage = ST_WAIT_SWAP;
            }
            exchange_post(cs, sh, it, lane, my_stop);
        }
        if (lane == 0) lds_st(&sh.d_done, it);
        PTRACE(8, it, 0);
        // the launch ends here if the record buffers are nearly used up
        if constexpr (!LOCK) {      // (a lock-step rank asks the others through its record)
            if (__builtin_expect(n_lik + 3 * nc > cap_lik || n_smp + 3 * nc > cap_smp, 0)) { if (it < last_iter) { last_iter = it; stop_code = 1; } }
        }
        it += 1;
        PSTAMP(2); PCOUNT(4);
    }
#ifdef HTM_STAMPS
    if (lane == 0) for (int k = 0; k < 8; ++k) sh.stamp_acc[8 + k] = st_acc[k];
#endif
    // ---- end of the launch: the swap of the last iteration, the chain set's state as the next launch (or the host) finds it
    const int last = it - 1;
    if (lane == 0) { sh.c.n_lik = n_lik; sh.c.n_smp = n_smp; }
    if constexpr (LOCK) {
        // (a lock-step rank leaves only when the swap of its last iteration has been applied: iteration counter, stream position,
        // temperatures and the stop word are settled with it)
        if (!dead && sh.c.err == 0 && last > i0 && sh.c.iter_done < last) exchange_finish(cs, sh, s_gath, last, lane, false);
        if (sh.c.err == 0) {
            if (in) cs.L[c] = L;
            if (lane == 0) {
                if (slog_cap > 0) sh.c.slog_n = min(slog_cap, slog_n0 + (sh.c.iter_done - i0) * nc);
                sh.c.n_full_evals += (long long)n_full;
                sh.c.n_partial_evals += (long long)n_part;
            }
        }
    } else
    if (!dead && sh.c.err == 0 && last > i0) {
        const PipeHdr &hp = sh.hdr[last & (kPipeSlots - 1)];
        const int i1 = uni(hp.i1), i2 = uni(hp.i2);
        const double sr = hp.sr, slr = hp.slr;
        if (n_all > 1 && i1 >= 0) {
            const double T1 = rl_f64(T, i1), T2 = rl_f64(T, i2), rT1 = rl_f64(rT, i1), rT2 = rl_f64(rT, i2);
            const double del_s = (rl_f64(L, i2) - rl_f64(L, i1)) * (rT1 - rT2);
            if (sr >= kEps && slr <= del_s) {
                if (lane == i1) { T = T2; rT = rT2; }
                if (lane == i2) { T = T1; rT = rT1; }
            }
        }
        if (in) { cs.temp[c] = T; cs.L[c] = L; }
        if (lane == 0) {
            if (n_all > 1) { sh.c.swap_i1 = i1; sh.c.swap_i2 = i2; sh.c.swap_r = sr; sh.c.swap_logr = slr; }
            sh.c.spos = sh.origin + nextB;
            sh.c.iter_done = last;
            sh.c.stage = ST_IDLE;
            if (sh.c.slog_cap > 0) sh.c.slog_n = min(sh.c.slog_cap, sh.c.slog_n + (last - i0) * nc);
            sh.c.n_full_evals += (long long)n_full;
            sh.c.n_partial_evals += (long long)n_part;
            if (last < sh.c.iter_target) sh.c.stop = stop_code ? stop_code : 2;
        }
    } else if (!dead && sh.c.err == 0 && lane == 0 && last < sh.c.iter_target) {
        sh.c.stop = stop_code ? stop_code : 2;          // (nothing could be produced: the host feeds the stream)
    }
    if (lane == 0) { cs.diag[25] += sh.n_flush; lds_st(&sh.quit, 1); }
}

// block 0 of a k_mcmc<NCH, F32, 5> launch
template <int NCH, bool F32, bool LOCK = false>
__device__ __forceinline__ void pipe_body(FwRef f_, CsRef cs_, int target_arg, int ring_size, int wmax, unsigned long long launch)
{
    CsRef cs = rebase(cs_);
    FwRef f = rebase(f_);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    PipeShared &sh = *reinterpret_cast<PipeShared *>(smem);
    char *carve = smem + ((sizeof(PipeShared) + 15) & ~size_t(15));
    Ring rg;
    rg.mask = ring_size - 1;
    rg.U = reinterpret_cast<double *>(carve);          carve += sizeof(double) * ring_size;
    rg.LOGU = reinterpret_cast<double *>(carve);       carve += sizeof(double) * ring_size;
    rg.pg = reinterpret_cast<double *>(carve);         carve += sizeof(double) * ring_size;
    rg.pr = reinterpret_cast<double *>(carve);         carve += sizeof(double) * ring_size;
    rg.plogr = reinterpret_cast<double *>(carve);      carve += sizeof(double) * ring_size;
    rg.dec = reinterpret_cast<int4 *>(carve);          carve += sizeof(int4) * ring_size;
    rg.sw = reinterpret_cast<int4 *>(carve);           carve += sizeof(int4) * ring_size;
    rg.hop = reinterpret_cast<int *>(carve);           carve += sizeof(int) * kHops * ring_size;
    double *s_sx = reinterpret_cast<double *>(carve);
    double *s_sy = s_sx + f.S;
    double *s_sz = s_sy + f.S;
    double *s_gath = s_sz + f.S;                       // the gathered swap records of a lock-step rank (exchange_finish)
    rg.mir_n = cs.mirror_n;
    rg.mx = s_gath + kGathStage;
    rg.mstep = rg.mx + rg.mir_n;
    rg.mir_steps = false;
    rg.lock = LOCK ? 1 : 0;
    const int nc = cs.n_chains;
    PipeRings pr;
    pr.nc = nc;
    pr.pd = rg.mx + rg.mir_n;
    pr.ed = pr.pd + (size_t)kPipeSlots * PD_N * nc;
    pr.pi = reinterpret_cast<int *>(pr.ed + (size_t)2 * kPipeSlots * ED_N * nc);
    pr.ei = pr.pi + (size_t)kPipeSlots * PI_N * nc;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    {
        const int vz0 = opaque_zero();
        constexpr int kCtrlWords = (int)(sizeof(Ctrl) / sizeof(int));
        if (tid < kCtrlWords) reinterpret_cast<int *>(&sh.c)[tid] = reinterpret_cast<const int *>(cs.ctrl)[tid + vz0];
        if (tid == kCtrlWords) sh.hop_end = cs.stream.hop_end[vz0];
    }
    for (int j = tid; j < f.S; j += blockDim.x) { s_sx[j] = f.sx[j]; s_sy[j] = f.sy[j]; s_sz[j] = f.sz[j]; }
    for (int k = tid; k < 7 * nc; k += blockDim.x) { sh.np[k] = 0; sh.na[k] = 0; }
    for (int k = tid; k < rg.mir_n; k += blockDim.x) rg.mx[k] = cs.xall[k];
    for (int k = tid; k < 2 * kPipeSlots * EI_N * nc; k += blockDim.x) pr.ei[k] = 0;
#ifdef HTM_STAMPS
    for (int k = tid; k < 96; k += blockDim.x) sh.stamp_acc[k] = 0ull;
#endif
    if (tid < kPipeSlots) sh.hdr[tid].tag = 0;
    __syncthreads();
    if (tid == 0) {
        if (target_arg >= 0) sh.c.iter_target = target_arg;
        sh.origin = sh.c.spos;
        const long long av = sh.hop_end - sh.c.spos;
        sh.avail = av > (1 << 30) ? (1 << 30) : (int)av;
        sh.fill = 0; sh.base = 0;
        sh.i0 = sh.c.iter_done;
        sh.q = 0ull; sh.epoch = 0; sh.fl_it = 0; sh.fl_base = 0;
        sh.d_done = sh.c.iter_done; sh.landed_it = sh.c.iter_done; sh.f_it = sh.c.iter_done; sh.f_stop_it = 0; sh.quit = 0;
        sh.n_flush = 0ull; sh.trace_n = 0u;
        sh.win_lo = 0; sh.win_want = 0; sh.win_seq = 0; sh.win_ack = 0;
        sh.xctl = 0u; sh.xctl_iter = -1;
    }
    __syncthreads();
    for (int c = tid; c < kMaxChains; c += blockDim.x) {
        sh.pver[c] = 0; sh.cland[c] = 0; sh.col_w[c] = 0; sh.col_tag[c] = 0;
        if (c < nc) {
            const double T = cs.temp[c], L = cs.L[c];
            sh.L[c] = L; sh.temp[c] = T; sh.rtemp[c] = 1.0 / T;
            const double b_ = cs.xall[c], q_ = cs.xall[nc + nc * cs.S + c];
            sh.rbeta[c] = 1.0 / b_; sh.katt[c] = (kPi * kFreq) / (q_ * b_);
        }
    }
    if (sh.c.iter_done >= sh.c.iter_target || sh.c.stop || sh.c.err) return;      // (uniform)
    if (sh.avail < 3 * wmax) {                                 // the produced stream does not cover a safe stretch: the host refills
        __syncthreads();
        if (tid == 0) { if (LOCK) sh.c.err = -7; else sh.c.stop = 2; *cs.ctrl = sh.c; }
        return;
    }
    prefetch_all(cs, sh, rg, min(2 * (6 * nc + 16) + 64, ring_size - 64));       // ends with a barrier
    if (tid == 0) sh.win_fill = sh.fill;
    __syncthreads();
#ifndef HTM_PIPE_SKIP
#define HTM_PIPE_SKIP 0
#endif
    if (wave == 0) { if constexpr (!(HTM_PIPE_SKIP & 1)) pipe_front<LOCK>(cs, sh, rg, pr, lane, launch); }
    else if (wave == 1) { if constexpr (!(HTM_PIPE_SKIP & 2)) pipe_decider<NCH, F32, LOCK>(f, cs, sh, rg, pr, s_gath, lane, wmax); }
    else if (wave == (int)(blockDim.x >> 6) - 1) { if constexpr (!(HTM_PIPE_SKIP & 4)) pipe_collector<NCH, F32, LOCK>(f, cs, sh, rg, pr, s_sx, s_sy, s_sz, lane); }
    else { if constexpr (!(HTM_PIPE_SKIP & 8)) pipe_evaluator<NCH, F32, LOCK>(f, cs, sh, rg, pr, s_sx, s_sy, s_sz, lane); }
    __syncthreads();
#ifdef HTM_STAMPS
    if (cs.stamps)
        for (int k = tid; k < 32; k += blockDim.x) if (sh.stamp_acc[k]) atomicAdd(&cs.stamps[32 + k], sh.stamp_acc[k]);
#endif
    for (int k = tid; k < 7 * nc; k += blockDim.x) {          // flush this launch's counters
        if (sh.np[k]) atomicAdd(&cs.n_propose[k], sh.np[k]);
        if (sh.na[k]) atomicAdd(&cs.n_accept[k], sh.na[k]);
    }
    if (tid == 0) *cs.ctrl = sh.c;
}

// One launch = the chain master (block 0) + W full-evaluation workers (blocks 1..W), all resident.
// `launch` = the host's count of k_mcmc launches of this chain set (1, 2, ...): orders and the quit word carry it, so
// nothing a previous launch left in memory can be mistaken for this launch's.
// Blocks of 8 waves; the pipelined master with one station per lane takes 12 (three per SIMD, 168 registers a wave): nine
// evaluators instead of five -- its throughput is the evaluators' (and the worker blocks take 12 events each).
#ifndef HTM_PIPE_THREADS
#define HTM_PIPE_THREADS 512
#endif
template <int NCH, int MK> constexpr int mcmc_threads() { return (MK >= 5 && NCH == 1) ? HTM_PIPE_THREADS : 512; }
template <int NCH, bool F32 = false, int MK = 0>
__global__ __launch_bounds__((mcmc_threads<NCH, MK>())) void k_mcmc(FwdDev f, ChainsDev cs, int mode, int target_arg,
                                               const double *gathered, int ring_size, int wmax,
                                               unsigned long long launch)
{
    const KArgLayout __attribute__((address_space(4))) &ka = *(const KArgLayout __attribute__((address_space(4))) *)__builtin_amdgcn_kernarg_segment_ptr();
    // MK 7: several master workgroups (blocks 0 .. n_mb - 1, eight chains each: flow_body<.., MB>); the workers follow them
    const int n_mb = MK == 7 ? (ka.cs.n_chains + 7) / 8 : 1;
    if ((int)blockIdx.x < n_mb) {
        // MK 3: the single-rank loop on the free-running master (flow_body); 0: the same loop with barriers (step_body)
        // 4: a lock-step rank (MODE_LOCKRUN, swap records exchanged inside the launch) on the free-running master; 2: with barriers
        if constexpr (MK == 3) flow_body<NCH, F32, false>(ka.f, ka.cs, target_arg, ring_size, wmax, launch);
        else if constexpr (MK == 4) flow_body<NCH, F32, true>(ka.f, ka.cs, target_arg, ring_size, wmax, launch);
        else if constexpr (MK == 5) pipe_body<NCH, F32, false>(ka.f, ka.cs, target_arg, ring_size, wmax, launch);      // the pipelined master: single rank
        else if constexpr (MK == 6) pipe_body<NCH, F32, true>(ka.f, ka.cs, target_arg, ring_size, wmax, launch);       // ... a lock-step rank (MODE_LOCKRUN)
        else if constexpr (MK == 7) { if (!flow_body<NCH, F32, false, true>(ka.f, ka.cs, target_arg, ring_size, wmax, launch)) return; }      // (only the workgroup that finishes last goes on)
        else step_body<NCH, true, F32, MK>(ka.f, ka.cs, mode, target_arg, gathered, ring_size, wmax, launch);
        // every exit of the master comes through here (its returns are uniform over the block): release the workers
        __syncthreads();
        if (threadIdx.x == 0) st_agent(&ka.cs.ps->quit, launch + 1ull);
    } else {
        worker_body<NCH, F32, mcmc_threads<NCH, MK>() / 64, (MK == 5 || MK == 6)>(ka.f, ka.cs, launch, (int)blockIdx.x - n_mb);
    }
}

}  // namespace htm
