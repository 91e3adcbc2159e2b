// htm_flow.hpp -- the FREE-RUNNING chain master of k_mcmc (single rank, MODE_RUN): the main loop of a rank
// (hypo_tremor_mcmc.f90:236-284) without workgroup barriers.
//
// step_body (htm_step.hpp) runs an iteration as  passes | barrier A | roles | barrier B | post : every iteration lasts as
// long as its slowest chain step plus ~3 k cycles of roles and post (round 2: 14.7 k cycles per iteration at 8 chains, of
// which a partial update is 7-8.6 k).  Here every chain wave runs its chain's steps back to back -- proposal, evaluation,
// decision, commit, records, the orders of its coming full evaluations, next step -- and the waves meet only where the
// algorithm couples the chains:
//   * the rank's random stream (mod_random, one serial stream shared by all chains): where a step starts depends on how
//     many draws every earlier step took.  The hop tables (htm_stream.hpp) predict that assuming every prior is ok; a
//     Rayleigh-prior rejection (cls_model.f90:178-181, no judge draw: cls_mcmc.f90:193) makes a step one draw shorter.
//     Rule: a step is COMMITTED only when every step before it in stream order -- (i, c') for c' < c, (i-1, c') for
//     c' > c -- has passed its prior check (`prog`, published ~3 k cycles into a step; the commit comes ~7 k cycles in:
//     that is the slack between the waves).  A step that finds its prior violated commits (as a rejection) when its turn
//     comes and then publishes an ANCHOR "step `key` starts at `pos`" under a new EPOCH: every later step re-predicts from
//     it (nothing after a pending rejection can have been committed, so nothing is ever undone).
//   * the temperature swap (cls_parallel.f90:121-136): the pair of iteration i-1 concerns two chains; each of their waves
//     evaluates the same decision from (T, L) of both chains after iteration i-1 (rings of four iterations in LDS), just
//     before its own decision of iteration i -- the first thing that needs the temperature.  Nobody else waits.
//   * records: slots by LDS atomics, sorted by (iteration, chain) on the host when drained.
// The protocol (prog / done / epoch / anchor, the turn rule, re-prediction from the anchor alone) is model-checked on the
// CPU against the serial loop under random interleavings: tools/flow_protocol_sim.py.
// Full evaluations: the worker blocks and the tagged-granule hand-off of htm_step.hpp, unchanged; the orders role P sent
// for all chains are sent by each chain's own wave right after its commit (one step ahead, or two around a hypocentre step).
#pragma once
#include "htm_step.hpp"

#define HTM_G __attribute__((address_space(1)))      // the global address space, stated where the compiler cannot infer it

namespace htm {

struct FlowShared : StepShared {
    unsigned long long prog[kMaxChains];   // {epoch << 1 | prior rejected : 32, key : 32} of the chain's latest checked step
    int done[kMaxChains];                  // key of the chain's latest committed step
    double L4[4][kMaxChains];              // log-likelihood after iteration i (index i & 3)
    double T4[4][kMaxChains];              // temperature DURING iteration i
    double rT4[4][kMaxChains];             // its reciprocal (what the Metropolis ratio and the swap multiply by; travels with T)
    unsigned long long anch[2];            // [epoch & 1] {key : 32, pos : 32}: step `key` starts at stream position `pos`
    int epoch;
    int Eof[4];                            // [i & 3] where the chain steps of iteration i end (= where its swap starts)
    // [i & 3] the swap of iteration i as the stream has it there: pair, draws it takes, judge_swap's draw and its log.  Written
    // with Eof by the last chain's wave when its step has passed its check (the positions are its own near future: inside the
    // LDS window); read when the swap is decided, up to two iterations later -- by then the window may have moved on
    int sw_i1[4], sw_i2[4], sw_nd[4];
    double sw_r[4], sw_logr[4];
    int last_iter;                         // the launch ends after this iteration
    int stop_code;
    int i0;                                // iterations completed before this launch
    // order book of every chain (owning wave only): the step starting at ob_pos has its order out under ob_tag
    int ob_pos[kMaxChains], ob_mode[kMaxChains], ob_mid[kMaxChains];     // ob_mid: type | event << 3 of the step in between (mode 2)
    unsigned ob_tag[kMaxChains];
    unsigned long long n_full_w, n_part_w;
    StreamDev sd;                          // the stream rings' addresses (chain 0's wave extends the LDS window from them every step)
};

// pf_load (htm_step.hpp) with the rings' addresses taken from LDS instead of the kernarg segment
__device__ __forceinline__ void flow_pf_load(PfRegs &r, const FlowShared &sh, int p, int limit)
{
    r.p = -1;
    if (p < limit) {
        const StreamDev &sd = sh.sd;
        const long long g = (sh.origin + p) & sd.mask;
        typedef const double HTM_G *gd; typedef const i32x4 HTM_G *gv;      // (pointers read from LDS: the address space is stated)
        r.U = ((gd)sd.U)[g]; r.LOGU = ((gd)sd.LOGU)[g]; r.pg = ((gd)sd.pg)[g]; r.pr = ((gd)sd.pr)[g]; r.plogr = ((gd)sd.plogr)[g];
        r.dec = ((gv)sd.dec)[g]; r.sw = ((gv)sd.sw)[g];
        const gv hs = (gv)(sd.hop + g * kHops);
        r.h0 = hs[0]; r.h1 = hs[1];
        r.p = p;
    }
}

// a wave-uniform value that reached a vector register (read from LDS, or the result of a vector compare) back in a scalar
// one: branches on it are scalar branches and the arithmetic behind it scalar arithmetic, not lane-masked code
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

// LDS words shared between the waves: relaxed accesses in program order.  LDS operations of a wave are executed in
// issue order and the LDS is one serialisation point for the workgroup, so "release" and "acquire" are compiler
// barriers here, not waits for outstanding memory operations.
__device__ __forceinline__ int lds_ld(const int *p) { const int v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); asm volatile("" ::: "memory"); return uni(v); }   // (every int read this way is at a wave-uniform address)
__device__ __forceinline__ unsigned long long lds_ld(const unsigned long long *p) { const unsigned long long v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); asm volatile("" ::: "memory"); return v; }
__device__ __forceinline__ void lds_st(int *p, int v) { asm volatile("" ::: "memory"); __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_st(unsigned long long *p, unsigned long long v) { asm volatile("" ::: "memory"); __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

struct FlowWave {                 // a wave's predictions (wave-uniform)
    int epoch, akey;              // the epoch they are made in, that epoch's anchor key
    int rc, rpos;                 // "chain rc's step of the current iteration starts at rpos" (rc = 0: the iteration's base)
    int rc1, rpos1;               // the same for the next iteration
    int B2;                       // base of the iteration after that (orders sent ahead only); -1 = not known
};

// The kernel arguments the loop uses at every step, read ONCE per launch and kept in scalar registers (or their spill lanes:
// a v_readlane) -- read at their uses (htm_step.hpp's rule for the 256-register kernels) each group of them is a scalar-cache
// round trip the wave waits for: a dozen per step, ~2.4 k of its 12 k cycles.  Member names as in FwdDev / ChainsDev, so that
// the forward model's templates (event_misfit, load_obs_regs) take it in place of the forward object.
// (launder() keeps the global address space visible: a pointer that went through a register move the compiler cannot see
// through would be accessed with FLAT instructions otherwise -- which also count as LDS operations in every lgkmcnt wait)
struct FlowHot {
    int S, E, use_time, use_amp;
    const double *t_obs, *t_prec, *a_obs, *a_prec, *rpsum_t, *rpsum_a;
    const float *t_obs32, *t_prec32, *a_obs32, *a_prec32;
    double const_sum;
    int n_chains, n_all;
    double *xall;
    const double *muall, *rs2all, *stall;
    const int *ptall;
    unsigned long long *slots, *pgran;
    int slot_rep, slot_stride, n_wg, pgran_stride;
    unsigned long long *stamps;
};
template <class T>
__device__ __forceinline__ void launder(T *&p) { T HTM_G *q = (T HTM_G *)p; asm volatile("" : "+s"(q)); p = (T *)q; }
__device__ __forceinline__ void launder(int &v) { asm volatile("" : "+s"(v)); }
__device__ __forceinline__ void flow_hot_load(FwRef f_, CsRef cs_, FlowHot &H)
{
    CsRef cs = rebase(cs_);
    FwRef f = rebase(f_);
    H.S = f.S; H.E = f.E; H.use_time = f.use_time; H.use_amp = f.use_amp;
    H.t_obs = f.t_obs; H.t_prec = f.t_prec; H.a_obs = f.a_obs; H.a_prec = f.a_prec; H.rpsum_t = f.rpsum_t; H.rpsum_a = f.rpsum_a;
    H.t_obs32 = f.t_obs32; H.t_prec32 = f.t_prec32; H.a_obs32 = f.a_obs32; H.a_prec32 = f.a_prec32;
    H.const_sum = f.const_sum;
    H.n_chains = cs.n_chains; H.n_all = cs.n_procs * cs.n_chains;
    H.xall = cs.xall; H.muall = cs.muall; H.rs2all = cs.rs2all; H.stall = cs.stall; H.ptall = cs.ptall;
    H.slots = cs.slots; H.pgran = cs.pgran;
    H.slot_rep = cs.slot_rep; H.slot_stride = cs.slot_stride; H.n_wg = cs.n_wg; H.pgran_stride = cs.pgran_stride;
    H.stamps = cs.stamps;
    // (opaque from here on: a value the compiler can re-load from the kernarg segment it would rather re-load than keep)
    launder(H.S); launder(H.E); launder(H.use_time); launder(H.use_amp);
    launder(H.t_obs); launder(H.t_prec); launder(H.a_obs); launder(H.a_prec); launder(H.rpsum_t); launder(H.rpsum_a);
    launder(H.t_obs32); launder(H.t_prec32); launder(H.a_obs32); launder(H.a_prec32);
    launder(H.n_chains); launder(H.n_all);
    launder(H.xall); launder(H.muall); launder(H.rs2all); launder(H.stall); launder(H.ptall);
    launder(H.slots); launder(H.pgran);
    launder(H.slot_rep); launder(H.slot_stride); launder(H.n_wg); launder(H.pgran_stride);
    launder(H.stamps);
}

#ifdef HTM_STAMPS
// diagnostic cycle accounting of the free-running master (tools/flow_stamps.py): per wave, [k] ticks of phase k summed over its
// partial-update steps (0 front: loads issued, 1 proposal + check published, 2 evaluation, 3 turn, 4 swap + decision + commit,
// 5 records + orders), 6 ticks of its full-evaluation steps, 7 / 8 the two counts, 9 ticks between steps (loop top), 10 wait part of 6
#define FSTAMP(k) do { if (lane == 0 && H.stamps) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); st_acc[k] += n_ - t_last; t_last = n_; } } while (0)
#else
#define FSTAMP(k) do { } while (0)
#endif

constexpr int kFlowRestart = -1;  // flow_step: the step's position was disproved before its turn came: run it again
constexpr int kFlowAbort = -2;    // flow_step: a wait gave up (sh.c.err is set)

// select_pair + the judge_swap draw starting at E (cls_parallel.f90:226-230, :163): pair, draws used in all (single rank:
// this rank draws both).  The stream service has the usual case precomputed (sw ring); more than 12 redraws follow the stream.
__device__ __forceinline__ bool flow_swap_at(int n_all, const Ring &rg, int E, int limit, int &i1, int &i2, int &nd)
{
    i1 = -1; i2 = -1; nd = 0;
    if (n_all <= 1) return true;
    if (E + 2 >= limit) return false;
    const i32x4 sw = reinterpret_cast<const i32x4 *>(rg.sw)[E & rg.mask];
    const int swz = uni(sw.z);
    if (swz > 0) { i1 = uni(sw.x); i2 = uni(sw.y); nd = swz + 1; return E + nd < limit; }
    int pos = E;
    i1 = uni((int)(rg.U[pos & rg.mask] * n_all));     // (= rand_u * n_proc * n_chain, cls_parallel.f90:226: one rank here)
    pos++;
    for (;;) {
        if (pos + 1 >= limit) return false;
        i2 = uni((int)(rg.U[pos & rg.mask] * n_all));
        pos++;
        if (i1 != i2) break;
    }
    nd = pos - E + 1;
    return true;
}

// base of the next iteration when n chain steps of this one remain from pos; -1 if the window does not cover it
__device__ __forceinline__ int flow_next_base(int n_all, const Ring &rg, int pos, int n, int limit)
{
    if (pos < 0) return -1;
    const int E = uni(hop_ahead(rg, pos, n));
    int i1, i2, nd;
    if (E + 16 >= limit || !flow_swap_at(n_all, rg, E, limit, i1, i2, nd)) return -1;
    return E + nd;
}

// the anchor as (iteration, chain, position of that chain's step).  A rejected LAST step of an iteration leaves pos = the
// end of that iteration's chain steps: chain 0 of the next iteration starts after the swap's draws.
__device__ __forceinline__ void flow_from_anchor(const FlowHot &H, const FlowShared &sh, const Ring &rg, unsigned long long a, int &ia, int &ca, int &ap)
{
    const int nc = H.n_chains;
    const int key = uni((int)(unsigned)(a >> 32)), pos = uni((int)(unsigned)a);
    ia = uni(sh.i0) + key / nc; ca = key - (key / nc) * nc; ap = pos;
    if (ca == 0) {
        int i1, i2, nd;
        flow_swap_at(H.n_all, rg, pos, 1 << 30, i1, i2, nd);     // (the rejected step's own wave read these positions: covered)
        ap = pos + nd;
    }
}

// every order this wave has out for its chains is void (their positions were predicted in another epoch): the workers are told
__device__ __forceinline__ void flow_void_books(const FlowHot &H, FlowShared &sh, int wave, int NW, int nc, int lane)
{
    for (int c = wave; c < nc; c += NW) {
        if (uni(sh.ob_pos[c]) != -1) {
            if (lane == 0) {
                for (int r = 0; r < H.slot_rep; ++r) st_gran(H.slots + (size_t)r * H.slot_stride + c * kGranPerSlot, 0u, 0u);   // (void_slot)
                sh.ob_pos[c] = -1;
            }
        }
    }
}

// this wave adopts epoch e (read from sh.epoch a moment ago).  `standing`: its current step (it, c) has passed its check and lies
// before the anchor -- it stands, and everything the wave runs after it starts at or after the anchor; else the current
// step itself starts at or after the anchor.  Returns false if the epoch moved on meanwhile (the caller looks again).
__device__ __forceinline__ bool flow_adopt(const FlowHot &H, FlowShared &sh, const Ring &rg, FlowWave &W, int e, int it, int c, bool in_turn, bool &stands)
{
    const int nc = H.n_chains;
    const unsigned long long a = lds_ld(&sh.anch[e & 1]);
    if (lds_ld(&sh.epoch) != e) return false;
    const int akey = uni((int)(unsigned)(a >> 32));
    const int key = (it - uni(sh.i0)) * nc + c;
    int ia, ca, ap;
    flow_from_anchor(H, sh, rg, a, ia, ca, ap);
    const int limit = uni(sh.fill);
    stands = in_turn && key < akey;
    W.epoch = e; W.akey = akey;
    if (!stands) {
        // the current step starts at or after the anchor: same iteration, or the anchor sits in the iteration before
        if (ia == it) { W.rc = ca; W.rpos = ap; }
        else { W.rc = 0; W.rpos = flow_next_base(H.n_all, rg, ap, nc - ca, 1 << 30); }
        W.rc1 = 0; W.rpos1 = flow_next_base(H.n_all, rg, W.rpos, nc - W.rc, limit);
        W.B2 = flow_next_base(H.n_all, rg, W.rpos1, nc, limit);
    } else if (ia == it) {
        W.rc = ca; W.rpos = ap;                         // (the wave's later chains of this iteration)
        W.rc1 = 0; W.rpos1 = flow_next_base(H.n_all, rg, ap, nc - ca, limit);
        W.B2 = flow_next_base(H.n_all, rg, W.rpos1, nc, limit);
    } else {                                            // the anchor is a step of the next iteration
        W.rc1 = ca; W.rpos1 = ap;
        W.B2 = flow_next_base(H.n_all, rg, ap, nc - ca, limit);
    }
    return true;
}

// The inputs of one chain step, REQUESTED ONE STEP AHEAD: where a wave's next step starts and what it proposes is in the
// stream window long before the step runs, so everything the step reads from memory -- the element it perturbs and the
// event's coordinates (gA: lane 0, lanes 1..3), prior mean, 1 / (2 sigma^2), step size and the event's reciprocal precision
// sums (gB: lanes 0..4), the prior type, the event's four observation rows and the chain's station corrections -- is
// requested with vector loads while the step before it is evaluated (the kernel has the registers: 150 of 256), and the
// step itself starts with its arithmetic.  What the step in between commits to the same chain is patched into the
// registers (flow_patch); vs, qs and the elements of the full-evaluation types come from the LDS mirror at use.
template <int N>
struct StepIn {
    int it, c, p, epoch;          // which step, its predicted start, the epoch of the prediction; p < 0: nothing requested
    int type, idx, evt, dec_w;    // the decoded proposal (htm_stream.hpp)
    double g, r, logr;            // its Gaussian, its Metropolis draw and log
    double gA, gB;
    int pt;
    double tob[N], tpr[N], aob[N], apr[N], tc[N], ac[N];
};

template <int NCH, bool F32>
__device__ __forceinline__ void flow_request(const FlowHot &H, const Ring &rg, const FlowWave &W, StepIn<(NCH > 0 ? NCH : 1)> &n,
                                             int it, int c, int p, int lane)
{
    const int M = rg.mask;
    asm volatile("" : "+v"(lane));
    n.it = it; n.c = c; n.p = p; n.epoch = W.epoch;
    const double *xall_ = H.xall, *muall_ = H.muall, *rs2all_ = H.rs2all, *stall_ = H.stall;
    const int *ptall_ = H.ptall;
    const int nc_ = H.n_chains, S_ = H.S, nh = 3 * H.E;
    const i32x4 dec = reinterpret_cast<const i32x4 *>(rg.dec)[p & M];
    const int type = __builtin_amdgcn_readfirstlane(dec.x), idx = __builtin_amdgcn_readfirstlane(dec.y);
    const int evt = __builtin_amdgcn_readfirstlane(dec.z);
    n.type = type; n.idx = idx; n.evt = evt; n.dec_w = __builtin_amdgcn_readfirstlane(dec.w);
    n.g = rg.pg[p & M]; n.r = rg.pr[p & M]; n.logr = rg.plogr[p & M];
    const bool partial = evt > 0 && it > 1;       // hypo_tremor_mcmc.f90:246
    const int off_tc = nc_, off_qs = nc_ + nc_ * S_, off_ac = 2 * nc_ + nc_ * S_, off_hy = 2 * nc_ + 2 * nc_ * S_;
    const int goff = type == 1 ? 0 : type == 2 ? off_tc : type == 3 ? off_qs : type == 4 ? off_ac : off_hy;
    const int gnx = (type == 1 || type == 3) ? 1 : (type == 2 || type == 4) ? S_ : nh;
    const int o = goff + c * gnx + idx;
    const int ev = evt > 0 ? evt - 1 : 0;
    const int o_h = off_hy + c * nh + 3 * ev;
    int ga = o;
    ga = lane == 1 ? o_h : ga; ga = lane == 2 ? o_h + 1 : ga; ga = lane == 3 ? o_h + 2 : ga;
    n.gA = xall_[ga];
    const double *pb = muall_ + o;
    pb = lane == 1 ? rs2all_ + o : pb; pb = lane == 2 ? stall_ + o : pb;
    pb = lane == 3 ? H.rpsum_t + ev : pb; pb = lane == 4 ? H.rpsum_a + ev : pb;
    n.gB = *pb;
    n.pt = ptall_[o + opaque_zero()];             // (a vector load: scalar loads in flight would hold up every LDS wait)
    if (partial) {
        if constexpr (NCH > 0) {
            const double *tc = xall_ + off_tc + c * S_, *ac = xall_ + off_ac + c * S_;
            const size_t base = (size_t)ev * (size_t)H.S;
            const bool ut = H.use_time != 0, ua = H.use_amp != 0;
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                const int j = lane + 64 * k;
                const bool valid = j < H.S;
                n.tob[k] = n.tpr[k] = n.aob[k] = n.apr[k] = 0.0; n.tc[k] = 0.0; n.ac[k] = 0.0;
                if (valid) {
                    n.tc[k] = tc[j]; n.ac[k] = ac[j];
                    if constexpr (F32) {
                        if (ut) { n.tob[k] = (double)H.t_obs32[base + j]; n.tpr[k] = (double)H.t_prec32[base + j]; }
                        if (ua) { n.aob[k] = (double)H.a_obs32[base + j]; n.apr[k] = (double)H.a_prec32[base + j]; }
                    } else {
                        if (ut) { n.tob[k] = H.t_obs[base + j]; n.tpr[k] = H.t_prec[base + j]; }
                        if (ua) { n.aob[k] = H.a_obs[base + j]; n.apr[k] = H.a_prec[base + j]; }
                    }
                }
            }
        }
    }
}

// what the step that has just been committed (chain c, accepted, element o := x_new) changes in the inputs requested for
// the wave's next step, if that is a step of the same chain
template <int N>
__device__ __forceinline__ void flow_patch(StepIn<N> &n, int c, int type, int idx, int evt, int o, double x_new, int lane,
                                           int nc_, int S_, int nh)
{
    if (n.p < 0 || n.c != c) return;
    if (type >= 5) {
        if (n.type >= 5) {
            const int off_hy = 2 * nc_ + 2 * nc_ * S_;
            const int on = off_hy + c * nh + n.idx;
            if (on == o && lane == 0) n.gA = x_new;
            if (n.evt == evt && lane == 1 + (idx - 3 * (evt - 1))) n.gA = x_new;
        }
    } else if (type == 2 || type == 4) {
#pragma unroll
        for (int k = 0; k < N; ++k)
            if (lane + 64 * k == idx) { if (type == 2) n.tc[k] = x_new; else n.ac[k] = x_new; }
    }
}

// The global stores of a step -- its commit and the order it sends ahead -- are ISSUED AT THE START OF THE WAVE'S NEXT STEP,
// behind that step's first use of its requested inputs.  Vector-memory operations of a wave complete in issue order and a
// write-through store is acknowledged by memory (~1 us): a step that waits for its inputs right after the commit store of
// the step before waits for that acknowledgement (measured: 2.5 k of a step's 12 k cycles).  Issued here, the stores are
// younger than everything the step waits for; they are a step old by the time anything waits behind them.
struct Deferred {
    int commit_o;                 // element to write, -1: none
    double commit_x;
    int ord_c;                    // chain whose order slot to write, -1: none
    unsigned ord_tag, ord_w1, ord_co, ord_rep;
    double ord_x, ord_cx;
};
__device__ __forceinline__ void flow_issue(const FlowHot &H, Deferred &df, int lane, unsigned long long launch)
{
    if (df.commit_o >= 0) {
        if (lane == 0) st_agent(H.xall + df.commit_o, df.commit_x);
        df.commit_o = -1;
    }
    if (df.ord_c >= 0) {
        if (lane < H.slot_rep * kGranPerSlot) {
            const int gi = lane & 7;
            const unsigned long long xb = (unsigned long long)__double_as_longlong(df.ord_x);
            const unsigned long long cb = (unsigned long long)__double_as_longlong(df.ord_cx);
            const unsigned pay = gi == 0 ? (unsigned)launch : gi == 1 ? df.ord_w1
                               : gi == 2 ? (unsigned)(xb >> 32) : gi == 3 ? (unsigned)xb
                               : gi == 4 ? df.ord_co                                    // the commit the workers must see, or ~0
                               : gi == 5 ? (unsigned)(cb >> 32) : gi == 6 ? (unsigned)cb
                               : df.ord_rep;                                            // element of the step in between (+1; 0 = none)
            st_gran(H.slots + (size_t)(lane >> 3) * H.slot_stride + df.ord_c * kGranPerSlot + gi, df.ord_tag, pay);
        }
        df.ord_c = -1;
    }
}

// the orders a chain may send ahead, looked up from positions alone while the step is evaluated; what depends on the
// step's outcome is filled in after its commit (flow_step)
struct PlanIn {
    int mode, pj, jt, ji, jo, mid, o_mid, epoch;
    double jx_old, jstep, jg;
};

// One chain step: its inputs were requested a step ago (cur); proposal, the request of the wave's NEXT step's inputs (nx),
// evaluation, the step's turn, decision, commit, records and the orders of the chain's coming full evaluations (the
// free-running counterpart of chain_pass).  All 64 lanes execute with identical (uniform) values; lane <-> station only inside
// event_misfit.  `ext`: this wave keeps the LDS window of the stream ahead (chain 0's wave, one round of <= 64 positions
// per step, in flight under the step's arithmetic).  Returns the stream position after the step, kFlowRestart or kFlowAbort.
template <int NCH, bool F32>
__device__ __forceinline__ int flow_step(const FlowHot &H, CsRef cs_, FlowShared &sh, const Ring &rg, FlowWave &W,
                                         const StepIn<(NCH > 0 ? NCH : 1)> &cur, StepIn<(NCH > 0 ? NCH : 1)> &nx, Deferred &df,
                                         const double (&rsx)[(NCH > 0 ? NCH : 1)], const double (&rsy)[(NCH > 0 ? NCH : 1)],
                                         const double (&rsz)[(NCH > 0 ? NCH : 1)],
                                         const double *s_sx, const double *s_sy, const double *s_sz,
                                         int lane, int wave, int NW, unsigned long long launch, bool ext, int look, int back,
                                         bool rec_now)
{
    constexpr int N = NCH > 0 ? NCH : 1;
    CsRef cs = rebase(cs_);      // (cold paths only: records, diagnostics; the loop's arguments are in H)
    // (lane predicates -- lane == 0, lane < n, ... -- are one compare where they are used; as loop invariants the compiler
    // keeps each as a 64-bit mask in a spilled scalar pair: two v_readlane per use)
    asm volatile("" : "+v"(lane));
    const int M = rg.mask;
    const int c = cur.c, iter = cur.it;
    const int p = cur.p;
#ifdef HTM_STAMPS
    unsigned long long st_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long t_last = __builtin_amdgcn_s_memtime(), t_wait = 0;
    const unsigned long long t_step0 = t_last;
#endif
    const int nc_ = H.n_chains, S_ = H.S, nh = 3 * H.E;
    const int type = cur.type, idx = cur.idx, evt = cur.evt, dec_w = cur.dec_w;
    const double g = cur.g, r_ring = cur.r, logr_ring = cur.logr;
    const bool partial = evt > 0 && iter > 1;       // hypo_tremor_mcmc.f90:246
    const int off_tc = nc_, off_qs = nc_ + nc_ * S_, off_ac = 2 * nc_ + nc_ * S_, off_hy = 2 * nc_ + 2 * nc_ * S_;
    const int goff = type == 1 ? 0 : type == 2 ? off_tc : type == 3 ? off_qs : type == 4 ? off_ac : off_hy;
    const int gnx = (type == 1 || type == 3) ? 1 : (type == 2 || type == 4) ? S_ : nh;
    const int o = goff + c * gnx + idx;             // element of the rank's parameter vector this step perturbs
    const int ev = partial ? evt - 1 : 0;
    const double *tc = H.xall + off_tc + c * S_, *ac = H.xall + off_ac + c * S_;
    // the book of this chain: is this step's order out already, and how
    const int book_pos = uni(sh.ob_pos[c]), book_mode = uni(sh.ob_mode[c]), book_mid = uni(sh.ob_mid[c]);
    const unsigned book_tag = (unsigned)uni((int)sh.ob_tag[c]);
    const bool pre = !partial && book_pos == p;
    const int pre_mode = pre ? book_mode : 0;
    // a full-evaluation step whose order went out two steps ahead adds the event of the step in between itself (below):
    // its inputs are requested now
    StaRegs<N> st;
    ObsRegs<N> ob;
    int d_e = 0;
    double d_ex = 0.0, d_ey = 0.0, d_ez = 0.0;
    const bool own_evt = NCH > 0 && pre_mode == 2 && (book_mid & 7) >= 5;
    if constexpr (NCH > 0) {
        if (__builtin_expect(own_evt, 0)) {
            d_e = __builtin_amdgcn_readfirstlane(book_mid >> 3) - 1;
            const int vzd = opaque_zero();
            const double *hypd = H.xall + off_hy + c * nh + 3 * d_e;
            d_ex = ld_state(hypd, vzd); d_ey = ld_state(hypd + 1, vzd); d_ez = ld_state(hypd + 2, vzd);
            // (the step in between is the chain's latest: if its commit is still waiting to be issued, memory has the old value)
            const int od = off_hy + c * nh + 3 * d_e;
            if (df.commit_o == od) d_ex = df.commit_x;
            if (df.commit_o == od + 1) d_ey = df.commit_x;
            if (df.commit_o == od + 2) d_ez = df.commit_x;
            load_sta_regs<NCH>(st, H.S, lane, s_sx, s_sy, s_sz, tc, ac, 0, -1, 0.0);
            load_obs_regs<NCH, F32>(ob, H, d_e, lane);
        } else if (partial) {
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                st.sx[k] = rsx[k]; st.sy[k] = rsy[k]; st.sz[k] = rsz[k]; st.tc[k] = cur.tc[k]; st.ac[k] = cur.ac[k];
                ob.tob[k] = cur.tob[k]; ob.tpr[k] = cur.tpr[k]; ob.aob[k] = cur.aob[k]; ob.apr[k] = cur.apr[k];
            }
            ob.rpst = H.use_time ? rl_f64(cur.gB, 3) : 1.0;
            ob.rpsa = H.use_amp ? rl_f64(cur.gB, 4) : 1.0;
        }
    }
    FSTAMP(0);
    const double x_old = type >= 5 ? rl_f64(cur.gA, 0) : rg.mx[o];       // (types 1..4: the LDS mirror, kept current by the commits)
    const double hx = rl_f64(cur.gA, 1), hy = rl_f64(cur.gA, 2), hz = rl_f64(cur.gA, 3);
    const double beta = rg.mx[c], q = rg.mx[off_qs + c];
    const double mu = rl_f64(cur.gB, 0), rs2 = rl_f64(cur.gB, 1), step = rl_f64(cur.gB, 2);
    const int ptype = __builtin_amdgcn_readfirstlane(cur.pt);
    FSTAMP(6);
    // ---- everything this step had requested is in: the stores of the step before go out now (see Deferred)
    drain_vmem();
    flow_issue(H, df, lane, launch);
    FSTAMP(7);
    const double L_cur = sh.L[c];
    const double x_new = x_old + g * step;                      // cls_model.f90:172
    const double da = x_new - mu, db = x_old - mu;
    double lpr = -(da * da - db * db) * rs2;                    // :175-177 (rs2 = 1 / (2 sigma^2), formed once on the host)
    int ok = 1;
    if (ptype == 1) {                                           // :178-187
        if (x_new <= mu) { lpr = (double)-1.0e+30f; ok = 0; }
        else lpr = lpr + log(x_new - mu) - log(x_old - mu);
    }
    ok = uni(ok);
    FSTAMP(8);
    const double r = ok ? r_ring : 0.0, logr = ok ? logr_ring : 0.0;
    const int cnt = dec_w - 1 + ok;                             // the judge draw happens only if prior_ok
    const int i0_ = uni(sh.i0), fill_ = uni(sh.fill), it_target = uni(sh.c.iter_target);
    const int key = (iter - i0_) * nc_ + c;
    // ---- the step has passed (or failed) its prior check: the later steps may go ahead on it
    if (c == nc_ - 1) {                                         // where this iteration's swap starts, and what it draws there
        const int E = p + cnt, k4 = iter & 3;
        int i1, i2, nd;
        flow_swap_at(H.n_all, rg, E, 1 << 30, i1, i2, nd);
        if (lane == 0) {
            sh.sw_i1[k4] = i1; sh.sw_i2[k4] = i2; sh.sw_nd[k4] = nd;
            if (nd > 0) { sh.sw_r[k4] = rg.U[(E + nd - 1) & M]; sh.sw_logr[k4] = rg.LOGU[(E + nd - 1) & M]; }
            lds_st(&sh.Eof[k4], E);
        }
    }
    if (lane == 0)
        lds_st(&sh.prog[c], ((unsigned long long)(((unsigned)W.epoch << 1) | (ok ? 0u : 1u)) << 32) | (unsigned)key);
    FSTAMP(9);
    // ---- the wave's NEXT step: where it starts (a prediction in this epoch), and the request of its inputs -- in flight
    // ---- under this step's evaluation, turn and commit
    nx.p = -1;
    {
        const int cn = c + NW < nc_ ? c + NW : wave;
        const bool same_it = cn > c;
        const int itn = same_it ? iter : iter + 1;
        const int rcn = same_it ? W.rc : W.rc1, rpn = same_it ? W.rpos : W.rpos1;
        if (ok != 0 && rpn >= 0 && cn >= rcn && itn <= it_target) {
            const int pn = uni(hop_ahead(rg, rpn, cn - rcn));
            if (pn + 16 < fill_) flow_request<NCH, F32>(H, rg, W, nx, itn, cn, pn, lane);
        }
    }
    // ---- the orders this chain may send ahead after this step (what role P does for all chains in step_body): its next
    // ---- step's if that needs the full evaluation (one step ahead), else -- that step being a hypocentre step -- the one
    // ---- after it (two steps ahead: the workers leave the event of the step in between out, this wave adds it).
    // ---- Looked up here from positions alone; sent after the commit.  A step uses an order only if it starts exactly where
    // ---- the order was written for, and an epoch change voids the book.
    FSTAMP(10);
    PlanIn pl;
    pl.mode = 0; pl.epoch = W.epoch; pl.pj = 0; pl.jt = 0; pl.ji = 0; pl.jo = 0; pl.mid = 0; pl.o_mid = -1; pl.jx_old = 0.0; pl.jstep = 0.0; pl.jg = 0.0;
    if (rg.mir_n > 0 && (book_pos == -1 || book_pos == p) && ok != 0 && iter + 1 <= it_target) {
        const int lim = fill_ - 8;
        int p1 = -1, d1x = 0, d1y = 0, d1z = 0;
        if (nx.p >= 0 && nx.c == c) { p1 = nx.p; d1x = nx.type; d1y = nx.idx; d1z = nx.evt; }     // (a wave with one chain)
        else if (W.rpos1 >= 0 && c >= W.rc1) {
            p1 = uni(hop_ahead(rg, W.rpos1, c - W.rc1));
            if (p1 < lim) {
                const i32x4 d1 = reinterpret_cast<const i32x4 *>(rg.dec)[p1 & M];
                d1x = uni(d1.x); d1y = uni(d1.y); d1z = uni(d1.z);
            } else p1 = -1;
        }
        const bool w1 = p1 >= 0;
        const bool job1 = w1 && d1x >= 1 && d1x <= 4;
        int mode = job1 ? 1 : 0, pj = p1, jt = d1x, ji = d1y;
        if (HTM_ALLOW2 && NCH > 0 && w1 && !job1 && iter + 2 <= it_target && W.B2 >= 0) {
            const int p2 = uni(hop_ahead(rg, W.B2, c));
            if (p2 < lim) {
                const i32x4 d2 = reinterpret_cast<const i32x4 *>(rg.dec)[p2 & M];
                const int d2x = uni(d2.x);
                if (d2x >= 1 && d2x <= 4) { mode = 2; pj = p2; jt = d2x; ji = uni(d2.y); }
            }
        }
        if (mode) {
            const int jgoff = jt == 1 ? 0 : jt == 2 ? nc_ : jt == 3 ? nc_ + nc_ * S_ : 2 * nc_ + nc_ * S_;
            pl.jo = jgoff + c * ((jt == 1 || jt == 3) ? 1 : S_) + ji;
            pl.jx_old = rg.mx[pl.jo];                                         // LDS mirror (this step's own commit: below)
            pl.jstep = rg.mir_steps ? rg.mstep[pl.jo] : H.stall[pl.jo + opaque_zero()];
            pl.jg = rg.pg[pj & M];
            pl.mode = mode; pl.pj = pj; pl.jt = jt; pl.ji = ji; pl.mid = d1x | (d1z << 3); pl.o_mid = off_hy + c * nh + d1y;
        }
    }
    FSTAMP(1);
    double L_new = 0.0;
    int need_full = 0;
    if (__builtin_expect(ok != 0, 1)) {
        if (__builtin_expect(partial, 1)) {
            const int cmp = idx - 3 * ev;        // 0 x, 1 y, 2 z of event ev (selects: see htm_step.hpp)
            const double px[2] = {hx, cmp == 0 ? x_new : hx};
            const double py[2] = {hy, cmp == 1 ? x_new : hy};
            const double pz[2] = {hz, cmp == 2 ? x_new : hz};
            double out[2];
            if constexpr (NCH > 0) event_misfit<NCH, 2, F32>(H, ob, lane, st, px, py, pz, beta, q, out);
            else event_misfit_generic<2>(H, ev, lane, s_sx, s_sy, s_sz, tc, ac, 0, -1, 0.0, px, py, pz, beta, q, out);
            L_new = L_cur + wave_sum1(out[0] - out[1]);
        } else {
            need_full = 1;
            // ---- work order: recognised (sent by this wave one or two steps ago) or sent now -------------------------
            unsigned long long tk = 0;
            if (lane == 0) {
                tk = pre ? (unsigned long long)book_tag : ((atomicAdd(&sh.c.jobs_total, 1ull) + 1ull) & 0x7fffffffull);
                if (tk == 0) tk = 0x7fffffffull;      // 0 = empty slot: never a tag (the counter wraps after 2^31 orders)
                // an order of its own overwrites the slot: whatever else is on the book for this chain is void with it
                sh.ob_pos[c] = -1;
            }
            const unsigned tag = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)tk);
            // every chain-state store of this wave has landed before a worker can see the order (write-through stores,
            // drained here; an order sent ahead names the commit the workers have to see instead)
            if (!pre) drain_vmem();
            if (!pre && lane < H.slot_rep * kGranPerSlot) {
                const int gi = lane & 7;
                const unsigned long long xb = (unsigned long long)__double_as_longlong(x_new);
                const unsigned pay = gi == 0 ? (unsigned)launch : gi == 1 ? ((unsigned)type | ((unsigned)idx << 3))
                                   : gi == 2 ? (unsigned)(xb >> 32) : gi == 3 ? (unsigned)xb
                                   : gi == 4 ? 0xffffffffu : 0u;          // no commit to wait for (drained above), nothing left out
                st_gran(H.slots + (size_t)(lane >> 3) * H.slot_stride + c * kGranPerSlot + gi, tag, pay);
            }
            // ---- the workers' partial sums: tagged granules, fixed summation order; two rounds of loads in flight --
            const unsigned long long *pg = H.pgran + (size_t)c * H.n_wg * H.pgran_stride;
            const int pgs = H.pgran_stride;
            double part = 0.0;
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();      // 100 MHz
            constexpr int kSweep = 4;                 // <= 256 workers (host-checked)
            unsigned long long hi[2][kSweep], lo[2][kSweep];
            int which = 0;
            auto issue = [&](int b) __attribute__((always_inline)) {
#pragma unroll
                for (int j = 0; j < kSweep; ++j) {
                    const int k = j * 64 + lane;
                    hi[b][j] = 0; lo[b][j] = 0;
                    if (k < H.n_wg) { hi[b][j] = ld_agent(pg + (size_t)pgs * k); lo[b][j] = ld_agent(pg + (size_t)pgs * k + 1); }
                }
            };
            auto complete = [&](int b) __attribute__((always_inline)) {
                bool got = true;
#pragma unroll
                for (int j = 0; j < kSweep; ++j)
                    if (j * 64 + lane < H.n_wg) got = got && (unsigned)(hi[b][j] >> 32) == tag && (unsigned)(lo[b][j] >> 32) == tag;
                return __all(got);
            };
            // an order sent two steps ahead was answered a step ago: its granules are requested now, under the evaluation
            // of the wave's own event
            const bool early = pre_mode == 2;
            if (early) issue(0);
            // An order sent TWO steps ahead was evaluated while the step in between (a hypocentre step of this chain) may
            // or may not have committed: the workers LEFT THAT EVENT OUT, this wave adds its misfit -- at the position the
            // event has now, under this step's proposed parameters.  The result does not depend on when the workers looked.
            double own_lane = 0.0;
            if constexpr (NCH > 0) {
                if (own_evt) {
                    if (type == 2 || type == 4) {          // this step's proposed correction, on the lane of its station
#pragma unroll
                        for (int k = 0; k < NCH; ++k) {
                            if (lane + 64 * k == idx) { if (type == 2) st.tc[k] = x_new; else st.ac[k] = x_new; }
                        }
                    }
                    const double pxd[1] = {d_ex}, pyd[1] = {d_ey}, pzd[1] = {d_ez};
                    double outd[1];
                    event_misfit<NCH, 1, F32>(H, ob, lane, st, pxd, pyd, pzd, type == 1 ? x_new : beta, type == 3 ? x_new : q, outd);
                    own_lane = outd[0];
                }
            }
            bool have = false;
            if (early) have = complete(0);
#ifdef HTM_STAMPS
            const unsigned long long tw0_ = __builtin_amdgcn_s_memtime();
#endif
            if (!have) {
                issue(0);
                for (;;) {
                    issue(1);
                    if (complete(0)) { which = 0; break; }
                    issue(0);
                    if (complete(1)) { which = 1; break; }
                    if (sh.c.err != 0) return kFlowAbort;
                    if (__builtin_amdgcn_s_memrealtime() - t0 > 500000000ull) {
                        if (lane == 0) {
                            sh.c.err = -8;
                            unsigned long long *dg = cs.diag;          // what was waited for: the host puts it into its message
                            dg[1] = c; dg[2] = tag; dg[3] = pre; dg[4] = pre_mode; dg[5] = iter; dg[6] = p; dg[7] = type; dg[8] = idx;
                            dg[9] = hi[0][0]; dg[10] = lo[0][0]; dg[11] = 1; dg[12] = p; dg[0] = 1;
                        }
                        return kFlowAbort;
                    }
                }
            }
#ifdef HTM_STAMPS
            t_wait = __builtin_amdgcn_s_memtime() - tw0_;
#endif
#pragma unroll
            for (int j = 0; j < kSweep; ++j)            // fixed order: worker lane, lane + 64, ...
                if (j * 64 < H.n_wg)
                    part += (j * 64 + lane < H.n_wg) ? (which == 0 ? gran_f64(hi[0][j], lo[0][j]) : gran_f64(hi[1][j], lo[1][j])) : 0.0;
            L_new = -wave_sum1(part + own_lane) - H.const_sum;       // cls_forward.f90:277-300
        }
    }
    // ---- the round of the LDS window of the stream rings that chain 0's wave owes per step: requested here, behind the
    // ---- evaluation (whose registers are free again), stored at the end of the step
    PfRegs pf;
    pf.p = -1;
    int fill_to = 0;
    if (ext) {
        const int fl = sh.fill;
        // (this step is chain c's: the iteration's base lies 4 c .. 6 c positions back; the other waves may still read `back`
        // positions behind it, and want `look` positions ahead of it)
        fill_to = min(min(p - 4 * c + look, sh.avail), p - 6 * c - back + M + 1);
        if (fill_to > fl + 64) fill_to = fl + 64;
        if (fill_to > fl) flow_pf_load(pf, sh, fl + lane, fill_to);
    }
    FSTAMP(2);

    // ---- the step's turn: every step before it in stream order has passed its check in this epoch (or lies before the
    // ---- epoch's anchor: checked earlier, final).  Lanes <-> chains.
    {
        const int key_i = (iter - i0_) * nc_, key_m = key_i - nc_;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        for (unsigned spin = 0;; ++spin) {
            const unsigned long long pv = lane < nc_ ? lds_ld(&sh.prog[lane]) : 0ull;
            const int e = lds_ld(&sh.epoch);
            if (__builtin_expect(e != W.epoch, 0)) {
                bool stands = false;
                if (!flow_adopt(H, sh, rg, W, e, iter, c, true, stands)) continue;
                flow_void_books(H, sh, wave, NW, nc_, lane);
                df.ord_c = -1;                   // (an order not yet issued is void with the book)
                nx.p = -1;                       // (requested for a position of the old epoch)
                if (!stands) return kFlowRestart;
                continue;
            }
            const int need = (lane < c ? key_i : key_m) + lane;
            const int pk = (int)(unsigned)pv;
            const unsigned ph = (unsigned)(pv >> 32);
            const bool okl = lane >= nc_ || lane == c || pk > need ||
                             (pk == need && (pk < W.akey || (ph == ((unsigned)W.epoch << 1))));
            if (__all(okl)) break;
            if ((spin & 15u) == 15u) {
                if (sh.c.err != 0) return kFlowAbort;
                if (__builtin_amdgcn_s_memrealtime() - t0 > 500000000ull) { if (lane == 0) sh.c.err = -12; return kFlowAbort; }
            }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    FSTAMP(3);
    // ---- the temperature of this iteration: the swap of the iteration before (cls_parallel.f90:121-136, :285-302), decided
    // ---- here by the waves of the two chains it concerns -- each evaluates the same expression on the same values
    const int par = iter & 3, ppar = (iter - 1) & 3;
    double T = sh.T4[par][c], rT = sh.rT4[par][c];        // (first iteration of a launch: written by the prologue)
    if (iter - 1 > i0_) {
        T = sh.T4[ppar][c]; rT = sh.rT4[ppar][c];
        if (H.n_all > 1) {
            // (written by the last chain's wave before it published its check; this step's turn has seen that check)
            const int i1 = lds_ld(&sh.sw_i1[ppar]), i2 = uni(sh.sw_i2[ppar]);
            if (c == i1 || c == i2) {
                const int o2 = c == i1 ? i2 : i1;
                const int want = (iter - 1 - i0_) * nc_ + o2;
                const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                for (unsigned spin = 0; lds_ld(&sh.done[o2]) < want; ++spin) {
                    if ((spin & 15u) == 15u) {
                        if (sh.c.err != 0) return kFlowAbort;
                        if (__builtin_amdgcn_s_memrealtime() - t0 > 500000000ull) { if (lane == 0) sh.c.err = -12; return kFlowAbort; }
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                const double sr = sh.sw_r[ppar], slr = sh.sw_logr[ppar];
                const double T1 = sh.T4[ppar][i1], T2 = sh.T4[ppar][i2], rT1 = sh.rT4[ppar][i1], rT2 = sh.rT4[ppar][i2];
                const double del_s = (sh.L4[ppar][i2] - sh.L4[ppar][i1]) * (rT1 - rT2);      // :292 (1/T formed once per temperature)
                if (sr >= kEps && slr <= del_s) { T = c == i1 ? T2 : T1; rT = c == i1 ? rT2 : rT1; }      // :131-136
            }
        }
    }
    const int acc = uni((ok != 0 && metropolis(L_new, L_cur, rT, lpr, r, logr)) ? 1 : 0);       // cls_mcmc.f90:193-203
    // this wave's chain-state stores of EARLIER steps have landed before this step's commit goes out: an order sent after
    // the commit names only the commit itself for the workers to wait for
    drain_vmem();
    const int cool = uni((T < 1.0 + kEps) ? 1 : 0);
    const double L_post = acc ? L_new : L_cur;
    if (lane == 0) {
        sh.T4[par][c] = T; sh.rT4[par][c] = rT;
        if (ok != 0) atomicAdd(need_full ? &sh.n_full_w : &sh.n_part_w, 1ull);      // (a step that is run again left before this point)
        if (cool) sh.np[c * 7 + type - 1] += 1;                 // cls_mcmc.f90:186-189
        if (acc) {                                              // :207-219
            if (o < rg.mir_n) rg.mx[o] = x_new;
            sh.L[c] = L_new;
            if (cool) sh.na[c * 7 + type - 1] += 1;
        }
        sh.L4[par][c] = L_post;
        lds_st(&sh.done[c], key);
    }
    if (acc) {
        // the store to the chain's state in memory: at the start of the wave's next step (Deferred) -- unless this step's
        // sample record is about to read the state back
        if (rec_now && cool) { if (lane == 0) st_agent(H.xall + o, x_new); }
        else { df.commit_o = o; df.commit_x = x_new; }
        flow_patch<N>(nx, c, type, idx, evt, o, x_new, lane, nc_, S_, nh);
    }
    FSTAMP(4);
    // ---- a rejected prior: this step was one draw shorter than the hop tables assume.  Everything after it starts
    // ---- elsewhere: new epoch, anchored at the step after this one
    if (__builtin_expect(ok == 0, 0)) {
        const int e1 = W.epoch + 1;
        if (lane == 0) {
            lds_st(&sh.anch[e1 & 1], ((unsigned long long)(unsigned)(key + 1) << 32) | (unsigned)(p + cnt));
            lds_st(&sh.epoch, e1);
        }
        bool stands = false;
        // (this wave's own view: as any wave whose step stands before the anchor; a later rejection may already have moved on)
        while (!flow_adopt(H, sh, rg, W, lds_ld(&sh.epoch), iter, c, true, stands)) { }
        flow_void_books(H, sh, wave, NW, nc_, lane);
        df.ord_c = -1;
        nx.p = -1;
    }
    // ---- records of this step (hypo_tremor_mcmc.f90:270-280): slots by LDS atomics, put in order on the host
    if (__builtin_expect(uni(sh.c.slog_cap) > 0, 0)) {
        const int row = uni(sh.c.slog_n) + (iter - i0_ - 1) * nc_ + c;
        if (lane == 0 && row < sh.c.slog_cap) {
            int32_t *ir = cs.slog_i + 8 * (size_t)row;
            double *dr = cs.slog_d + 4 * (size_t)row;
            ir[0] = iter; ir[1] = c; ir[2] = type; ir[3] = idx + 1; ir[4] = ok; ir[5] = acc; ir[6] = need_full; ir[7] = 0;
            dr[0] = x_new; dr[1] = L_new; dr[2] = L_post; dr[3] = T;
        }
    }
    if (__builtin_expect(rec_now && cool, 0)) {
        int sl = 0, ss = -1;
        if (lane == 0) {
            sl = atomicAdd(&sh.c.n_lik, 1);
            if (iter > cs.n_burn) ss = atomicAdd(&sh.c.n_smp, 1);
        }
        sl = __builtin_amdgcn_readfirstlane(sl); ss = __builtin_amdgcn_readfirstlane(ss);
        if (lane == 0 && sl < cs.cap_lik) { cs.lik_iter[sl] = iter; cs.lik_chain[sl] = c; cs.lik_val[sl] = L_post; }
        if (ss >= 0 && ss < cs.cap_smp) {
            const int S = cs.S, rec = nh + 2 * S + 2;
            double *dst = cs.smp_data + (size_t)ss * rec;
            const double *hxp = cs.xall + off_hy + (size_t)c * nh;
            const int vz = opaque_zero();
            for (int k = lane; k < nh; k += 64) dst[k] = ld_state(hxp + k, vz);
            for (int k = lane; k < S; k += 64) {
                dst[nh + k] = ld_state(cs.xall + off_tc + (size_t)c * S + k, vz);
                dst[nh + S + k] = ld_state(cs.xall + off_ac + (size_t)c * S + k, vz);
            }
            if (lane == 0) {
                dst[nh + 2 * S] = ld_state(cs.xall + c, vz);
                dst[nh + 2 * S + 1] = ld_state(cs.xall + off_qs + c, vz);
                cs.smp_iter[ss] = iter; cs.smp_chain[ss] = c;
            }
        }
    }
    // ---- the order looked up above goes out, if this epoch still stands and the book is free
    if (pl.mode != 0 && pl.epoch == W.epoch && uni(sh.ob_pos[c]) == -1) {
        int mode = pl.mode;
        const double jx_old = (acc && o == pl.jo) ? x_new : pl.jx_old;        // (this step's own commit of that very element)
        const double jx_new = jx_old + pl.jg * pl.jstep;                       // cls_model.f90:172, as the step will compute it
        if (cs.rayleigh14) {                                                  // a Rayleigh prior among vs/qs/corrections (:178-187)
            if (ld_const(cs.ptall + pl.jo) == 1 && jx_new <= ld_const(cs.muall + pl.jo)) mode = 0;      // prior rejects: no evaluation
        }
        // two ahead: the workers wait for this step's commit by reading its value back; the step in between must not
        // be able to overwrite that very element before they look
        if (mode == 2 && acc && o == pl.o_mid) mode = 0;
        if (mode) {
            unsigned long long tk = 0;
            if (lane == 0) {
                tk = (atomicAdd(&sh.c.jobs_total, 1ull) + 1ull) & 0x7fffffffull;
                if (tk == 0) tk = 0x7fffffffull;
                sh.ob_pos[c] = pl.pj; sh.ob_tag[c] = (unsigned)tk; sh.ob_mode[c] = mode; sh.ob_mid[c] = pl.mid;
            }
            const unsigned tag = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)tk);
            df.ord_c = c; df.ord_tag = tag; df.ord_w1 = (unsigned)pl.jt | ((unsigned)pl.ji << 3);
            df.ord_x = jx_new; df.ord_cx = x_new;
            df.ord_co = acc ? (unsigned)o : 0xffffffffu;
            df.ord_rep = mode == 2 ? (unsigned)pl.o_mid + 1u : 0u;
        }
    }
    if (ext && fill_to > sh.fill) {
        pf_store(pf, rg);
        if (lane == 0) lds_st(&sh.fill, fill_to);
    }
#ifdef HTM_STAMPS
    FSTAMP(5);
    if (lane == 0 && H.stamps) {
        unsigned long long *a = sh.stamp_acc + 12 * (wave & 7);
        if (need_full) { a[6] += t_last - t_step0; a[8] += 1; a[10] += t_wait; }
        else { for (int k = 0; k < 6; ++k) a[k] += st_acc[k]; a[7] += 1; if (wave == 3) for (int k = 6; k < 11; ++k) atomicAdd((unsigned long long *)&H.stamps[96 + k], st_acc[k]); }
    }
#endif
    return p + cnt;
}

// block 0 of a k_mcmc<NCH, F32, 0> launch when the host selects the free-running master (htm_hip.hip: flow_ok)
template <int NCH, bool F32>
__device__ __forceinline__ void flow_body(FwRef f_, CsRef cs_, int target_arg, int ring_size, int wmax, unsigned long long launch)
{
    CsRef cs = rebase(cs_);
    FwRef f = rebase(f_);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    FlowShared &sh = *reinterpret_cast<FlowShared *>(smem);
    char *carve = smem + ((sizeof(FlowShared) + 15) & ~size_t(15));
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
    double *s_gath = s_sz + f.S;
    rg.mir_n = cs.mirror_n;
    rg.mx = s_gath + kGathStage;
    rg.mstep = rg.mx + rg.mir_n;
    rg.mir_steps = cs.mirror_steps != 0;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NW = blockDim.x >> 6;
    const int nc = cs.n_chains;
    {
        const int vz0 = opaque_zero();
        constexpr int kCtrlWords = (int)(sizeof(Ctrl) / sizeof(int));
        if (tid < kCtrlWords) reinterpret_cast<int *>(&sh.c)[tid] = reinterpret_cast<const int *>(cs.ctrl)[tid + vz0];
        if (tid == kCtrlWords) sh.hop_end = cs.stream.hop_end[vz0];
    }
    for (int j = tid; j < f.S; j += blockDim.x) { s_sx[j] = f.sx[j]; s_sy[j] = f.sy[j]; s_sz[j] = f.sz[j]; }
    for (int k = tid; k < 7 * nc; k += blockDim.x) { sh.np[k] = 0; sh.na[k] = 0; }
#ifdef HTM_STAMPS
    for (int k = tid; k < 96; k += blockDim.x) sh.stamp_acc[k] = 0ull;
#endif
    for (int k = tid; k < rg.mir_n; k += blockDim.x) { rg.mx[k] = cs.xall[k]; if (rg.mir_steps) rg.mstep[k] = cs.stall[k]; }
    __syncthreads();
    if (tid == 0) {
        if (target_arg >= 0) sh.c.iter_target = target_arg;
        sh.origin = sh.c.spos;
        const long long av = sh.hop_end - sh.c.spos;
        sh.avail = av > (1 << 30) ? (1 << 30) : (int)av;
        sh.fill = 0; sh.base = 0;
        sh.epoch = 0; sh.anch[0] = 0ull; sh.anch[1] = 0ull;
        sh.i0 = sh.c.iter_done; sh.last_iter = sh.c.iter_target; sh.stop_code = 0;
        sh.n_full_w = 0ull; sh.n_part_w = 0ull;
        sh.sd.raw = cs.stream.raw; sh.sd.U = cs.stream.U; sh.sd.LOGU = cs.stream.LOGU; sh.sd.G = cs.stream.G; sh.sd.dec = cs.stream.dec;
        sh.sd.pg = cs.stream.pg; sh.sd.pr = cs.stream.pr; sh.sd.plogr = cs.stream.plogr; sh.sd.hop = cs.stream.hop; sh.sd.sw = cs.stream.sw;
        sh.sd.mask = cs.stream.mask; sh.sd.gen = cs.stream.gen; sh.sd.hop_end = cs.stream.hop_end;
    }
    __syncthreads();
    const int i0 = sh.i0;
    for (int c = tid; c < kMaxChains; c += blockDim.x) {
        sh.ob_pos[c] = -1;
        sh.prog[c] = (unsigned long long)(unsigned)c;          // key(i0, c), epoch 0, prior ok
        sh.done[c] = c;
        if (c < nc) {
            const double T = cs.temp[c], L = cs.L[c];
            sh.L[c] = L;
            sh.T4[(i0 + 1) & 3][c] = T; sh.T4[i0 & 3][c] = T; sh.L4[i0 & 3][c] = L;
            sh.rT4[(i0 + 1) & 3][c] = 1.0 / T; sh.rT4[i0 & 3][c] = 1.0 / T;
        }
    }
    if (sh.c.iter_done >= sh.c.iter_target || sh.c.stop || sh.c.err) return;      // (uniform)
    if (sh.avail < 3 * wmax) {                                 // the produced stream does not cover a safe stretch: the host refills
        __syncthreads();
        if (tid == 0) { sh.c.stop = 2; *cs.ctrl = sh.c; }
        return;
    }
    // the draws an iteration can take: 6 per chain step + select_pair's and judge_swap's
    const int wd = 6 * nc + 16;
    const int look = 3 * wd + 24, back = wd + 8;
    prefetch_all(cs, sh, rg, min(look, ring_size - back));       // ends with a barrier
    const int n_int = cs.n_interval;
    int rec_phase = (i0 + 1) % n_int;                            // iteration % n_interval, kept by increments

    FlowHot H;
    flow_hot_load(f, cs, H);
    FlowWave W;
    W.epoch = 0; W.akey = 0; W.rc = 0; W.rpos = 0;
    W.rc1 = 0; W.rpos1 = flow_next_base(H.n_all, rg, 0, nc, sh.fill);
    W.B2 = flow_next_base(H.n_all, rg, W.rpos1, nc, sh.fill);
    constexpr int N = NCH > 0 ? NCH : 1;
    // station coordinates of this wave's lanes: resident in registers for the whole launch
    double rsx[N], rsy[N], rsz[N];
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const int j = lane + 64 * k;
        const bool valid = NCH > 0 && j < f.S;
        rsx[k] = valid ? s_sx[j] : 0.0; rsy[k] = valid ? s_sy[j] : 0.0; rsz[k] = valid ? s_sz[j] : 0.0;
    }
    Deferred df;
    df.commit_o = -1; df.commit_x = 0.0; df.ord_c = -1; df.ord_tag = 0; df.ord_w1 = 0; df.ord_co = 0; df.ord_rep = 0; df.ord_x = 0.0; df.ord_cx = 0.0;
    StepIn<N> nx;                     // the inputs of the wave's next step, requested while the step before it runs
    nx.p = -1; nx.it = 0; nx.c = 0; nx.epoch = 0; nx.type = 5; nx.idx = 0; nx.evt = 1; nx.dec_w = 6;
    nx.g = 0.0; nx.r = 0.0; nx.logr = 0.0; nx.gA = 0.0; nx.gB = 0.0; nx.pt = 0;
#pragma unroll
    for (int k = 0; k < N; ++k) { nx.tob[k] = nx.tpr[k] = nx.aob[k] = nx.apr[k] = 0.0; nx.tc[k] = nx.ac[k] = 0.0; }
    int iter = i0 + 1;
    int c = wave;
    bool alive = wave < nc;
#ifdef HTM_STAMPS
    const unsigned long long t_loop0 = __builtin_amdgcn_s_memtime();
#endif
    while (alive) {
        // ---- top of a step: the epoch its position is predicted in
        {
            const int e = lds_ld(&sh.epoch);
            if (__builtin_expect(e != W.epoch, 0)) {
                bool stands = false;
                if (!flow_adopt(H, sh, rg, W, e, iter, c, false, stands)) continue;
                flow_void_books(H, sh, wave, NW, nc, lane);
                df.ord_c = -1;
                nx.p = -1;
            }
        }
        if (iter > lds_ld(&sh.last_iter) || lds_ld(&sh.c.err) != 0) break;
        if (__builtin_expect(W.rpos1 < 0 || W.B2 < 0, 0)) {      // predictions the window did not cover when they were made
            if (W.rpos1 < 0) { W.rpos1 = flow_next_base(H.n_all, rg, W.rpos, nc - W.rc, sh.fill); W.rc1 = 0; }
            if (W.B2 < 0 && W.rpos1 >= 0) W.B2 = flow_next_base(H.n_all, rg, W.rpos1, nc - W.rc1, sh.fill);
        }
        if (__builtin_expect(nx.p < 0 || nx.it != iter || nx.c != c || nx.epoch != W.epoch, 0)) {
            // nothing (valid) was requested ahead for this step -- first step of a launch, after an epoch change, or the
            // window did not reach: look its position up and request its inputs now
            const int p = uni(hop_ahead(rg, W.rpos, c - W.rc));
            if (__builtin_expect(p + 16 >= lds_ld(&sh.fill), 0)) {
                // the window covers every step that can run (chain 0's wave keeps it 3 iterations ahead); a fail-stop
                if (wave == 0) { if (lane == 0) sh.c.err = -13; break; }
                const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                bool dead = false;
                while (p + 16 >= lds_ld(&sh.fill)) {
                    if (sh.c.err != 0 || __builtin_amdgcn_s_memrealtime() - t0 > 500000000ull) { dead = true; break; }
                    __builtin_amdgcn_s_sleep(4);
                }
                if (dead) { if (lane == 0 && sh.c.err == 0) sh.c.err = -13; break; }
            }
            // (the request reads the chain's state: a commit still waiting to be issued goes out first)
            flow_issue(H, df, lane, launch);
            flow_request<NCH, F32>(H, rg, W, nx, iter, c, p, lane);
        }
        const StepIn<N> cur = nx;
        if (wave == 0 && c == 0 && lane == 0) {
            // chain 0's wave decides where the launch ends: record buffers or produced stream nearly used up.  Everybody
            // learns it before committing a step of this iteration (its turn waits for chain 0's check)
            int code = 0;
            if (sh.c.n_lik + 3 * nc > cs.cap_lik || sh.c.n_smp + 3 * nc > cs.cap_smp) code = 1;
            else if (sh.avail < cur.p + 3 * wd + 32) code = 2;
            if (code && lds_ld(&sh.last_iter) > iter) { sh.stop_code = code; lds_st(&sh.last_iter, iter); }
        }
        const int r = flow_step<NCH, F32>(H, cs, sh, rg, W, cur, nx, df, rsx, rsy, rsz, s_sx, s_sy, s_sz, lane, wave, NW, launch,
                                          wave == 0, look, back, rec_phase == 1);
        if (r == kFlowRestart) continue;
        if (r == kFlowAbort) break;
        // ---- this wave's next step
        c += NW;
        if (c >= nc) {
            c = wave;
            iter += 1;
            rec_phase = rec_phase + 1 == n_int ? 0 : rec_phase + 1;
            if (W.rpos1 < 0) {    // (the window did not cover the prediction when it was made: it does now)
                W.rpos1 = flow_next_base(H.n_all, rg, W.rpos, nc - W.rc, 1 << 30); W.rc1 = 0;
                W.B2 = -1;
            }
            W.rc = W.rc1; W.rpos = W.rpos1;
            W.rc1 = 0; W.rpos1 = W.B2 >= 0 ? W.B2 : flow_next_base(H.n_all, rg, W.rpos, nc - W.rc, sh.fill);
            W.B2 = flow_next_base(H.n_all, rg, W.rpos1, nc, sh.fill);
        }
    }
    flow_issue(H, df, lane, launch);      // the last step's stores
#ifdef HTM_STAMPS
    if (lane == 0 && H.stamps && wave < 8) sh.stamp_acc[12 * wave + 11] += __builtin_amdgcn_s_memtime() - t_loop0;
#endif
    __syncthreads();
#ifdef HTM_STAMPS
    if (cs.stamps)
        for (int k = tid; k < 96; k += blockDim.x) if (sh.stamp_acc[k]) atomicAdd(&cs.stamps[k], sh.stamp_acc[k]);
#endif
    // ---- every step up to last_iter is committed: the swap of the last iteration, counters, the launch's end state
    if (tid == 0 && sh.c.err == 0) {
        const int last = min(sh.last_iter, sh.c.iter_target);
        if (last > i0) {
            const int par = last & 3;
            const int E = sh.Eof[par], i1 = sh.sw_i1[par], i2 = sh.sw_i2[par], nd = sh.sw_nd[par];
            if (cs.n_procs * nc > 1) {
                const double sr = sh.sw_r[par], slr = sh.sw_logr[par];
                const double T1 = sh.T4[par][i1], T2 = sh.T4[par][i2];
                const double del_s = (sh.L4[par][i2] - sh.L4[par][i1]) * (sh.rT4[par][i1] - sh.rT4[par][i2]);
                if (sr >= kEps && slr <= del_s) { sh.T4[par][i1] = T2; sh.T4[par][i2] = T1; }
                sh.c.swap_i1 = i1; sh.c.swap_i2 = i2; sh.c.swap_r = sr; sh.c.swap_logr = slr;
            }
            for (int k = 0; k < nc; ++k) { cs.temp[k] = sh.T4[last & 3][k]; cs.L[k] = sh.L[k]; }
            sh.c.spos = sh.origin + E + nd;
            sh.c.iter_done = last;
            sh.c.stage = ST_IDLE;
            if (sh.c.slog_cap > 0) sh.c.slog_n = min(sh.c.slog_cap, sh.c.slog_n + (last - i0) * nc);
            sh.c.n_full_evals += (long long)sh.n_full_w;
            sh.c.n_partial_evals += (long long)sh.n_part_w;
            if (last < sh.c.iter_target) sh.c.stop = sh.stop_code;
        }
    }
    __syncthreads();
    for (int k = tid; k < 7 * nc; k += blockDim.x) {          // flush this launch's counters
        if (sh.np[k]) atomicAdd(&cs.n_propose[k], sh.np[k]);
        if (sh.na[k]) atomicAdd(&cs.n_accept[k], sh.na[k]);
    }
    if (tid == 0) *cs.ctrl = sh.c;
}

// One launch = the chain master (block 0) + W full-evaluation workers (blocks 1..W), all resident.
// `launch` = the host's count of k_mcmc launches of this chain set (1, 2, ...): orders and the quit word carry it, so
// nothing a previous launch left in memory can be mistaken for this launch's.
template <int NCH, bool F32 = false, int MK = 0>
__global__ __launch_bounds__(512) void k_mcmc(FwdDev f, ChainsDev cs, int mode, int target_arg,
                                               const double *gathered, int ring_size, int wmax,
                                               unsigned long long launch)
{
    const KArgLayout __attribute__((address_space(4))) &ka = *(const KArgLayout __attribute__((address_space(4))) *)__builtin_amdgcn_kernarg_segment_ptr();
    if (blockIdx.x == 0) {
        // MK 3: the single-rank loop on the free-running master (flow_body); 0: the same loop with barriers (step_body)
        if constexpr (MK == 3) flow_body<NCH, F32>(ka.f, ka.cs, target_arg, ring_size, wmax, launch);
        else step_body<NCH, true, F32, MK>(ka.f, ka.cs, mode, target_arg, gathered, ring_size, wmax, launch);
        // every exit of the master comes through here (its returns are uniform over the block): release the workers
        __syncthreads();
        if (threadIdx.x == 0) st_agent(&ka.cs.ps->quit, launch + 1ull);
    } else {
        worker_body<NCH, F32>(ka.f, ka.cs, launch);
    }
}

}  // namespace htm
