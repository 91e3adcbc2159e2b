// htm_step.hpp -- k_mcmc / k_step: the main loop of a rank (hypo_tremor_mcmc.f90:236-284) on the device.
//
// k_mcmc (the product path): block 0 = chain MASTER, blocks 1..W = full-evaluation WORKERS, one persistent launch.
// k_step  (fallback, HTM_PERSIST=0): the master's code alone; exits at every full evaluation, k_full is its own launch.
//
// Master workgroup, per iteration
//   passes   chain waves, wave <-> chain (chains c = wave, wave+NW, ...), lane <-> station: look up where the step
//            starts in the rank's random stream, read the pre-decoded proposal found there (cls_mcmc.f90:134-165),
//            perturb the model (cls_model.f90:162-190), evaluate the single-event partial update
//            (cls_forward.f90:307-362) -- or collect the workers' partial sums of the full evaluation
//            (cls_forward.f90:268-303) --, Metropolis decision (cls_mcmc.f90:193-203), speculative commit.
//            No workgroup barrier in between.
//   roles    between the two barriers, lanes <-> chains, one wave each: V validates the optimistic stream
//            positions (a Rayleigh-prior rejection makes a step one draw shorter, cls_mcmc.f90:193) and commits the
//            RNG position; R record slots and step log; W temperature swap (cls_parallel.f90:121-136,:226-230);
//            P (k_mcmc) publishes the work orders of the NEXT iteration's full evaluations; one more wave extends
//            the LDS window of the stream rings (htm_stream.hpp).  Lane-parallel (ballot / DPP scan), no serial
//            loops over chains.
//   post     swap applied by the waves owning the two chains, samples recorded.
// The loop ends when the target is reached, a record buffer is full or the produced stream runs out.
#pragma once
#include "htm_device.hpp"
#include "htm_stream.hpp"

#ifndef HTM_ALLOW2
#define HTM_ALLOW2 1     // diagnostics: 0 = role P sends orders one iteration ahead only
#endif

namespace htm {

// Kernel arguments are read where they are used, through references into the kernarg segment (constant address space:
// scalar loads), re-based at the entry of every phase by an opaque move -- so the ~190 dwords of FwdDev + ChainsDev are
// never all alive at once.  (Taken by value they are loaded at kernel entry, hoisted out of the iteration loop and
// carried through it in spilled lanes: ~800 v_readlane reloads in the master's loop.)
typedef const ChainsDev __attribute__((address_space(4))) &CsRef;
typedef const FwdDev __attribute__((address_space(4))) &FwRef;
typedef const StreamDev __attribute__((address_space(4))) &SdRef;
template <class T>
__device__ __forceinline__ const T __attribute__((address_space(4))) &rebase(const T __attribute__((address_space(4))) &r)
{
    const T __attribute__((address_space(4))) *p = &r;
    asm volatile("" : "+s"(p));
    return *p;
}
struct KArgLayout { FwdDev f; ChainsDev cs; };

constexpr int kGathStage = 512;   // doubles of LDS for the gathered swap records (else they are read in place)

struct StepShared {
    Proposal prop[kMaxChains];
    double temp[kMaxChains], L[kMaxChains];
    double rtemp[kMaxChains];     // 1 / temperature (the Metropolis ratio multiplies by it), renewed wherever temp is written
    int start[kMaxChains];        // stream position (relative) at which each chain step started
    int start_fix[kMaxChains];    // corrected starts for a repeat pass (written by wave 0)
    int cnt[kMaxChains];          // draws the step consumed (judge draw included iff prior_ok)
    int slot_l[kMaxChains], slot_s[kMaxChains];
    int redone[kMaxChains];       // the chain repeated its step in this iteration (undo + new commit: role P stays out)
    // role P (k_mcmc): [iteration & 1][chain]: the chain's step of that iteration, starting at position pre_p, already
    // has its order out under pre_tag; pre_mode 1: sent one iteration ahead, 2: two iterations ahead (evaluated on
    // the state before the step in between; see chain_pass)
    int pre_p[2][kMaxChains];
    unsigned pre_tag[2][kMaxChains];
    int pre_mode[2][kMaxChains];
    int pre_pa[2][kMaxChains];   // two-ahead orders: where the step in between was expected to start
    int np[kMaxChains * 7], na[kMaxChains * 7];   // proposal / acceptance counters of this launch
    int sw_do, sw_c1, sw_c2;      // swap decided between the barriers; applied by the waves owning the chains
    double sw_T1, sw_T2;          // new temperatures of chains sw_c1 / sw_c2
    long long origin;             // absolute stream position of relative position 0 (= spos at launch)
    long long hop_end;            // StreamDev::hop_end when this launch started
    int avail;                    // the produced stream covers relative positions < avail
    int fill;                     // the LDS ring holds relative positions < fill
    int base;                     // relative position at which the current iteration starts
    int redo;                     // >= 0: chains >= redo repeat their pass; -1: validated; -2: aborted
    int catchup;                  // written between the barriers: the LDS window must be extended first
    int rolep_iter;               // role P has finished for this iteration (see step_body)
    double xrec[4 + 2 * kMaxChains];   // MODE_LOCKRUN: this rank's swap record of the iteration (what goes to every rank's inbox)
    int xdone;                    // MODE_LOCKRUN: the swap of every iteration <= xdone has been applied (see exchange_finish)
    int xstop, xspec;             // decided by role V: this rank asks for a stop; the next iteration starts before the swap is known
    unsigned xctl;                // the control word this rank posted with its latest record (exchange_post; read back by exchange_finish)
    int xctl_iter;                // ... and the iteration it belongs to
    Ctrl c;
#ifdef HTM_STAMPS
    unsigned long long stamp_acc[96];   // diagnostic cycle accounting of this launch, flushed to ChainsDev::stamps at its end
#endif
};

struct Ring {                     // LDS window of the stream rings, index = relative position & mask
    double *U, *LOGU, *pg, *pr, *plogr;
    int4 *dec, *sw;
    int *hop;
    int mask;
    // k_mcmc only: LDS mirror of the non-hypocentre part of the rank's parameter vector (vs, t_corr, qs, a_corr
    // of every chain; same offsets as ChainsDev::xall) and of its step sizes, kept current by the commits, so
    // that role P needs no memory round trip.  mir_n = 0: no mirror (too large, or not the persistent kernel)
    double *mx, *mstep;
    int mir_n;
    bool mir_steps;
    int lock;                     // htm_flow.hpp: 1 = a lock-step rank (select_pair is rank 0's, judge_swap's draw the pair's first rank's)
};

constexpr int kPassRestart = -1;  // chain_pass: the swap of the iteration before moved this rank's stream position: start again
constexpr int kPassAbort = -2;    // chain_pass: the job stops (or failed) after the iteration before: this step is not taken


// one stream position in flight from the global rings to the LDS window
typedef int i32x4 __attribute__((ext_vector_type(4)));      // (a plain vector: HIP's int4 class keeps temporaries in private memory)
struct PfRegs {
    double U, LOGU, pg, pr, plogr;
    i32x4 dec, sw, h0, h1;
    int p;                        // relative position, -1: nothing to store
};

__device__ __forceinline__ void pf_load(PfRegs &r, CsRef cs_, const StepShared &sh, int p, int limit)
{
    CsRef cs = rebase(cs_);
    r.p = -1;
    if (p < limit) {
        SdRef sd = cs.stream;
        const long long g = (sh.origin + p) & sd.mask;
        r.U = sd.U[g]; r.LOGU = sd.LOGU[g]; r.pg = sd.pg[g]; r.pr = sd.pr[g]; r.plogr = sd.plogr[g];
        r.dec = reinterpret_cast<const i32x4 *>(sd.dec)[g]; r.sw = reinterpret_cast<const i32x4 *>(sd.sw)[g];
        const i32x4 *hs = reinterpret_cast<const i32x4 *>(sd.hop + g * kHops);
        r.h0 = hs[0]; r.h1 = hs[1];
        r.p = p;
    }
}

__device__ __forceinline__ void pf_store(const PfRegs &r, const Ring &rg)
{
    if (r.p >= 0) {
        const int l = r.p & rg.mask;
        rg.U[l] = r.U; rg.LOGU[l] = r.LOGU; rg.pg[l] = r.pg; rg.pr[l] = r.pr; rg.plogr[l] = r.plogr;
        reinterpret_cast<i32x4 *>(rg.dec)[l] = r.dec; reinterpret_cast<i32x4 *>(rg.sw)[l] = r.sw;
        i32x4 *hd = reinterpret_cast<i32x4 *>(rg.hop + l * kHops);
        hd[0] = r.h0; hd[1] = r.h1;
    }
}

// every thread of the workgroup: extend the LDS window to cover relative positions < target
// (kernel start; later only if a long select_pair redraw run ate the look-ahead)
__device__ __forceinline__ void prefetch_all(CsRef cs_, StepShared &sh, const Ring &rg, int target)
{
    CsRef cs = rebase(cs_);
    if (target > sh.avail) target = sh.avail;
    if (target > sh.base + rg.mask + 1 - 8) target = sh.base + rg.mask + 1 - 8;
    const int fl = sh.fill;
    for (int p = fl + (int)threadIdx.x; p < target; p += (int)blockDim.x) {
        PfRegs r;
        pf_load(r, cs, sh, p, target);
        pf_store(r, rg);
    }
    __syncthreads();
    if (threadIdx.x == 0 && target > sh.fill) sh.fill = target;
    __syncthreads();
}

// The rank's parameter vector is ONE allocation per field, groups in proposal-type order [vs | t_corr | qs | a_corr | hypo],
// each [n_chains][nx] (ChainsDev::xall and friends): every element is base + integer offset -- scalar arithmetic on one
// pointer instead of a choice between the five ModelDev windows (fewer kernel arguments alive in the loop).
struct GroupOff { int tc, qs, ac, hy, nh; };
__device__ __forceinline__ GroupOff group_offsets(CsRef cs_)
{
    CsRef cs = rebase(cs_);
    const int nc = cs.n_chains, S = cs.S;
    GroupOff g;
    g.tc = nc; g.qs = nc + nc * S; g.ac = 2 * nc + nc * S; g.hy = 2 * nc + 2 * nc * S; g.nh = 3 * cs.E;
    return g;
}
// element idx of chain c's model of proposal type `type` (1 vs, 2 t_corr, 3 qs, 4 a_corr, 5..7 hypo)
__device__ __forceinline__ int elem_offset(CsRef cs_, int type, int c, int idx)
{
    CsRef cs = rebase(cs_);
    const GroupOff g = group_offsets(cs);
    const int goff = type == 1 ? 0 : type == 2 ? g.tc : type == 3 ? g.qs : type == 4 ? g.ac : g.hy;
    const int gnx = (type == 1 || type == 3) ? 1 : (type == 2 || type == 4) ? cs.S : g.nh;
    return goff + c * gnx + idx;
}

// inclusive prefix sum over the 64 lanes (DPP row_shr scan + row_bcast, ints)
__device__ __forceinline__ int wave_incl_scan(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false);   // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false);   // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false);   // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false);   // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);   // row_bcast:15
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);   // row_bcast:31
    return v;
}

// Chain state (x vectors, temperatures, log-likelihoods) is rewritten by this kernel while it loops over
// iterations.  hipcc turns uniform-address loads into scalar (K$) loads, and the scalar cache is not
// coherent with vector stores, so mutable state must be read with VECTOR loads.  `volatile` would do that but
// also makes the backend wait for every single load (s_waitcnt vmcnt(0) after each), serialising ~700-cycle
// latencies; instead the address gets a per-lane zero the compiler cannot see through, which keeps the loads
// ordinary (L1-cached, freely overlapped) vector loads.
__device__ __forceinline__ int opaque_zero()
{
    int z;
    asm volatile("v_mov_b32 %0, 0" : "=v"(z));
    return z;
}
__device__ __forceinline__ double ld_state(const double *p, int vz) { return p[vz]; }

// lane l of a double, as a wave-uniform value
__device__ __forceinline__ double rl_f64(double v, int l)
{
    const long long b = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, l);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)b >> 32), l);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// agent-scope relaxed accesses: sc1 write-through stores / L1-bypassing loads (MI355X_MICROARCH.md,
// inter-workgroup visibility).  Used for everything another workgroup of the same launch reads or writes.
__device__ __forceinline__ void st_agent(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(unsigned long long *p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ld_agent(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int ld_agent(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long ld_agent(const unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_gran(unsigned long long *g, unsigned tag, unsigned payload)
{
    st_agent(g, ((unsigned long long)tag << 32) | payload);
}
__device__ __forceinline__ void st_gran_f64(unsigned long long *g, unsigned tag, double v)   // two granules
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    st_gran(g, tag, (unsigned)(b >> 32));
    st_gran(g + 1, tag, (unsigned)b);
}
__device__ __forceinline__ double gran_f64(unsigned long long hi, unsigned long long lo)
{
    return __longlong_as_double((long long)(((hi & 0xffffffffull) << 32) | (lo & 0xffffffffull)));
}
__device__ __forceinline__ void drain_vmem() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// system-scope relaxed accesses: what another GPU (or another process on this one) writes into / reads from this
// rank's memory over xGMI while the kernel runs (swap-record inboxes, fine-grained allocations)
__device__ __forceinline__ void st_sys(unsigned long long *p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ unsigned long long ld_sys(const unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

// The master takes back whatever order chain c's slot holds (k_mcmc): granule 0 loses its tag, so the slot is no longer
// a complete order -- a worker that has not looked yet never takes it, a worker waiting for the order's named commit
// sees the slot change and drops it.  This is how a void order is SIGNALLED (the chain repeated a pass, or the order
// was written for a stream position the step does not start at); nothing on the hand-off path is decided by a clock.
__device__ __forceinline__ void void_slot(CsRef cs_, int c)
{
    CsRef cs = rebase(cs_);
    for (int r = 0; r < cs.slot_rep; ++r) st_gran(cs.slots + (size_t)r * cs.slot_stride + c * kGranPerSlot, 0u, 0u);
}

__device__ __forceinline__ bool metropolis(double L_new, double L_cur, double T, double lpr, double r,
                                           double logr)
{
    double ratio = (L_new - L_cur) * T;     // cls_mcmc.f90:194-195; T = 1 / temperature here (formed once per swap, not per step)
    ratio = ratio + lpr;
    return r >= kEps && logr <= ratio;      // :198-199
}

#ifdef HTM_STAMPS
#define CSTAMP(k) do { if (stamp_me) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); sh.stamp_acc[32 + (k)] += n_ - t_last; t_last = n_; } } while (0)
#else
#define CSTAMP(k) do { } while (0)
#endif

// Orders of role P that a step's ACTUAL start position proves wrong are taken back before the step runs (k_mcmc, chain
// wave, after role P of this iteration is through).  A wave's first chain starts exactly where role P predicted (same
// base, same hop table); with more chains than waves the later ones start where the wave's previous chain really ended,
// and in the lock-step kernel every step restarts elsewhere when a swap's judge draw turns out to be this rank's.  Then
// (a) an order addressed to this iteration but to another start position will never be used, and (b) a two-ahead order
// for the next iteration was written around ANOTHER step in between: it names the chain's latest commit for the workers
// to wait for, and this step may overwrite that very element (role P ruled that out for the step it predicted only).
// Workers waiting for a value that never shows serve no other order -- every chain would end up waiting for them -- and
// role P's own stale checks run one iteration too late for that.  The slot holds one order: whatever is on the books goes.
__device__ __forceinline__ void drop_disproved_orders(CsRef cs, StepShared &sh, int c, int p, int iter, int lane)
{
    if (__builtin_expect(cs.dbg & 1, 0)) return;     // HTM_DEBUG_NO_DROP=1 (tests: the workers' deferral alone must keep the job alive)
    const int book_a = sh.pre_p[iter & 1][c], book_b = sh.pre_p[(iter + 1) & 1][c];
    if (__builtin_expect((book_a != -1 && book_a != p) || book_b != -1, 0)) {
        const bool dead_a = book_a != -1 && book_a != p;
        const bool dead_b = book_b != -1 && sh.pre_mode[(iter + 1) & 1][c] == 2 && sh.pre_pa[(iter + 1) & 1][c] != p;
        if ((dead_a || dead_b) && lane == 0) {
            void_slot(cs, c);
            sh.pre_p[0][c] = -1; sh.pre_p[1][c] = -1;
        }
    }
}

// One chain step: proposal, single-event partial update, Metropolis decision and -- speculatively -- its
// commit (cls_mcmc.f90:186-189,:207-219).  The commit is final unless the validation finds that an earlier
// chain's Rayleigh prior rejected (then the step is undone and repeated one draw earlier, see undo_chain).
// All 64 lanes execute with identical (uniform) values; lane <-> station only inside event_misfit.
// Returns the stream position after this step.
//
// PERSIST (k_mcmc): a proposal that needs the full evaluation (cls_forward.f90:268-303) is handed to the
// worker blocks from HERE, by the chain's own wave, as soon as the proposed value is known: the order goes out
// while the other chain waves are still in their partial updates, so the workers' round trip is hidden behind
// them.  The wave then collects the workers' partial sums, judges and commits like any other step.
template <int NCH, bool PERSIST, bool F32, int MK>
__device__ __forceinline__ int chain_pass(FwRef f_, CsRef cs_, StepShared &sh, const Ring &rg,
                                          const double *s_sx, const double *s_sy, const double *s_sz, int c,
                                          int p, int iter, int lane, unsigned long long launch, bool wait_rolep, bool first_pass,
                                          int xwait, int base_used)
{
    constexpr bool LOCK = MK == 2;         // persistent lock-step (the swap of the iteration before may still be open)
    CsRef cs = rebase(cs_);
    FwRef f = rebase(f_);
    const int M = rg.mask;
#ifdef HTM_STAMPS
    const bool stamp_me = cs.stamps && lane == 0 && c == 0;
    unsigned long long t_last = __builtin_amdgcn_s_memtime();
#endif
    p = __builtin_amdgcn_readfirstlane(p);
    // the kernel arguments of the step's front, requested in ONE batch before the stream window is read (left to the
    // compiler they are loaded one dependent round trip after the other, next to their uses, behind the window's data)
    const double *xall_ = cs.xall;
    const PriorRec *prior_ = cs.prior;
    const int nc_ = cs.n_chains, S_ = cs.S, nh = 3 * cs.E, psame_ = cs.prior_same;
    const int4 dec = rg.dec[p & M];                 // decoded ahead of time (htm_stream.hpp)
    asm volatile("" : "+s"(xall_), "+s"(prior_));
    // wave-uniform by construction: keep them in scalar registers (addresses and selects become SALU work)
    const int type = __builtin_amdgcn_readfirstlane(dec.x), idx = __builtin_amdgcn_readfirstlane(dec.y);
    const int evt = __builtin_amdgcn_readfirstlane(dec.z), dec_w = __builtin_amdgcn_readfirstlane(dec.w);
    const double g = rg.pg[p & M], r_ring = rg.pr[p & M], logr_ring = rg.plogr[p & M];
    const bool partial = evt > 0 && iter > 1;       // hypo_tremor_mcmc.f90:246
    const int off_tc = nc_, off_qs = nc_ + nc_ * S_, off_ac = 2 * nc_ + nc_ * S_, off_hy = 2 * nc_ + 2 * nc_ * S_;
    const int goff = type == 1 ? 0 : type == 2 ? off_tc : type == 3 ? off_qs : type == 4 ? off_ac : off_hy;
    const int gnx = (type == 1 || type == 3) ? 1 : (type == 2 || type == 4) ? S_ : nh;
    const int o = goff + c * gnx + idx;             // element of the rank's parameter vector this step perturbs
    const int ev = partial ? evt - 1 : 0;
    // every global load of the step is issued here, before any dependent arithmetic.  The mutable scalars
    // (current value, event coordinates, vs, qs) come with ONE gathered vector load, lane k <- item k;
    // the immutable ones (prior, step size) with scalar loads.
    const int o_h = off_hy + c * nh + 3 * ev;
    int goffs = o;
    goffs = lane == 1 ? o_h : goffs; goffs = lane == 2 ? o_h + 1 : goffs; goffs = lane == 3 ? o_h + 2 : goffs;
    goffs = lane == 4 ? c : goffs; goffs = lane == 5 ? off_qs + c : goffs;
    const double gathered_v = xall_[goffs];
    const PriorRec prr = ld_prior(prior_ + (psame_ ? o - c * gnx : o));      // (one 32-byte scalar load: PriorRec, htm_device.hpp)
    const double mu = prr.mu, rs2 = prr.rs2, step = prr.step;
    const int ptype = prr.ptype;
    const double *tc = xall_ + off_tc + c * S_, *ac = xall_ + off_ac + c * S_;
    StaRegs<(NCH > 0 ? NCH : 1)> st;
    ObsRegs<(NCH > 0 ? NCH : 1), F32> ob;
    if (partial) {
        if constexpr (NCH > 0) {
            load_sta_regs<NCH>(st, f.S, lane, s_sx, s_sy, s_sz, tc, ac, 0, -1, 0.0);
            load_obs_regs<NCH, F32>(ob, f, ev, lane);      // in flight while the proposal is worked out
        }
    }
    // A full-evaluation step whose order went out two iterations ahead may need the one-event correction (below): its
    // inputs -- that event's observation rows and coordinates, the corrections -- are requested now, into the registers
    // a partial update would use.  Only a hint (role P of this iteration may still be writing pre_*): the decision to
    // use them is taken after the wait on sh.rolep_iter.
    bool dhint = false;
    int d_e = 0;
    double d_ex = 0.0, d_ey = 0.0, d_ez = 0.0;
    if constexpr (PERSIST && NCH > 0) {
        if (__builtin_expect(!partial && wait_rolep, 0)) {
            const Proposal &pv = sh.prop[c];
            if (first_pass && sh.pre_p[iter & 1][c] == p && sh.pre_mode[iter & 1][c] == 2 && pv.type >= 5) {
                dhint = true;
                d_e = __builtin_amdgcn_readfirstlane(pv.evt) - 1;
                const int vzd = opaque_zero();
                const double *hypd = xall_ + off_hy + c * nh + 3 * d_e;
                d_ex = ld_state(hypd, vzd); d_ey = ld_state(hypd + 1, vzd); d_ez = ld_state(hypd + 2, vzd);
                load_sta_regs<NCH>(st, f.S, lane, s_sx, s_sy, s_sz, tc, ac, 0, -1, 0.0);
                load_obs_regs<NCH, F32>(ob, f, d_e, lane);
            }
        }
    }
    const double x_old = rl_f64(gathered_v, 0);
    const double hx = rl_f64(gathered_v, 1), hy = rl_f64(gathered_v, 2), hz = rl_f64(gathered_v, 3);
    const double beta = rl_f64(gathered_v, 4), q = rl_f64(gathered_v, 5);
    const double L_cur = sh.L[c];
    double T = sh.temp[c], rT = sh.rtemp[c];   // (LOCK: read again after the swap is known, below)
    // MODE_LOCKRUN runs the front of this step -- loads, proposal, misfit -- while the swap records of the iteration
    // before (xwait) are still travelling between the ranks.  Everything that needs the swap's outcome waits HERE, just
    // before the decision: the temperature, whether the job goes on, and where this rank's stream really stands (the
    // judge_swap draw is taken from the stream of the rank owning chain 1 of the pair, cls_parallel.f90:163).
    // Wave 0 collects the records and decides the swap before its own step (step_body); the others wait for its word here.
    CSTAMP(0);   // decode + load issue
    const double x_new = x_old + g * step;                      // cls_model.f90:172
    const double da = x_new - mu, db = x_old - mu;
    double lpr = -(da * da - db * db) * rs2;                    // :175-177 (rs2 = 1 / (2 sigma^2), formed once on the host)
    int ok = 1;
    if (__builtin_expect(ptype == 1, 0)) {                      // :178-187
        if (x_new <= mu) { lpr = (double)-1.0e+30f; ok = 0; }
        else lpr = lpr + htm_log(x_new - mu) - htm_log(x_old - mu);      // (the same logarithm as flow_step and the pipelined front: whichever loop runs a job, the decisions agree)
    }
    const double r = ok ? r_ring : 0.0, logr = ok ? logr_ring : 0.0;
    const int cnt = dec_w - 1 + ok;                             // the judge draw happens only if prior_ok
    CSTAMP(1);   // proposal arithmetic (waits for the model loads)

    if constexpr (PERSIST) {
        // from here on this step reads what role P writes (orders already out) and overwrites what it reads (the
        // previous step's record, the LDS mirror): role P of this iteration must be through -- it has been for long
        if (wait_rolep)
            while (__hip_atomic_load(&sh.rolep_iter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != iter) __builtin_amdgcn_s_sleep(1);
    }
    double L_new = 0.0;
    int need_full = 0, acc = 0;
    if (__builtin_expect(ok != 0, 1)) {
        if (__builtin_expect(partial, 1)) {
            // plain selects: an if/else-if/else chain of stores into these arrays was miscompiled by
            // hipcc 7.2 at -O3 (the final else-store was dropped), see DESIGN.md §7
            const int cmp = idx - 3 * ev;        // 0 x, 1 y, 2 z of event ev
            const double px[2] = {hx, cmp == 0 ? x_new : hx};
            const double py[2] = {hy, cmp == 1 ? x_new : hy};
            const double pz[2] = {hz, cmp == 2 ? x_new : hz};
            double out[2];
            if constexpr (NCH > 0) event_misfit<NCH, 2, F32>(f, ob, lane, st, px, py, pz, beta, q, out);
            else event_misfit_generic<2>(f, ev, lane, s_sx, s_sy, s_sz, tc, ac, 0, -1, 0.0, px, py, pz, beta, q, out);
            CSTAMP(2);   // event_misfit
            L_new = L_cur + wave_sum1(out[0] - out[1]);
            if constexpr (!LOCK) acc = metropolis(L_new, L_cur, rT, lpr, r, logr) ? 1 : 0;      // (LOCK: decided below, once the swap is known)
            CSTAMP(3);   // final sum + decision
        } else {
            need_full = 1;
            if constexpr (PERSIST) {
                // ---- work order: tag = ticket (unique over the life of the chain set) ------------------------------
                const int par = iter & 1;
                // role P sent it already, one or two iterations ago.  Only in the first pass of an iteration: in a repeated
                // pass this chain has run -- and taken back -- another step meanwhile, and workers that got to the order late
                // may have read that step's value
                const bool pre = first_pass && sh.pre_p[par][c] == p;
                const int pre_mode = pre ? sh.pre_mode[par][c] : 0;
#ifdef HTM_STAMPS
                if (lane == 0 && cs.stamps) atomicAdd(&sh.stamp_acc[80 + (c & 7) + (pre_mode == 2 ? 8 : 0)], 1ull);
#endif
                const int prev_type = sh.prop[c].type, prev_evt = sh.prop[c].evt;     // the step in between (two-ahead orders)
                unsigned long long tk = 0;
                if (lane == 0) {
                    tk = pre ? (unsigned long long)sh.pre_tag[par][c] : ((atomicAdd(&sh.c.jobs_total, 1ull) + 1ull) & 0x7fffffffull);
                    if (tk == 0) tk = 0x7fffffffull;      // 0 = empty slot: never a tag (the counter wraps after 2^31 orders)
                    sh.pre_p[par][c] = -1;
                    // an order of its own overwrites the workers' sums of any order role P has out for this chain (a repeated
                    // pass can bring the step back to the position such an order was written for): those are void now
                    if (!pre) { sh.pre_p[par][c] = -1; sh.pre_p[par ^ 1][c] = -1; }
                }
                const unsigned tag = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)tk);
                // every chain-state store of this wave (earlier commits, undo) must have landed before a worker
                // can see the order: write-through stores, drained here; the order itself is one store
                // instruction (lane -> replica, granule), and its granules carry the tag, so no flag follows.  (An order role P
                // sent needs none of this -- and the wait would be for the loads of the wave's own event, issued a moment ago.)
                if (!pre) drain_vmem();
                if (!pre && lane < cs.slot_rep * kGranPerSlot) {
                    const int gi = lane & 7;
                    const unsigned long long xb = (unsigned long long)__double_as_longlong(x_new);
                    const unsigned pay = gi == 0 ? (unsigned)launch : gi == 1 ? ((unsigned)type | ((unsigned)idx << 3))
                                       : gi == 2 ? (unsigned)(xb >> 32) : gi == 3 ? (unsigned)xb
                                       : gi == 4 ? 0xffffffffu : 0u;          // no commit to wait for (drained above), nothing to report
                    st_gran(cs.slots + (size_t)(lane >> 3) * cs.slot_stride + c * kGranPerSlot + gi, tag, pay);
                }
#ifdef HTM_STAMPS
                if (lane == 0 && cs.stamps) { atomicAdd(&cs.stamps[20], __builtin_amdgcn_s_memrealtime()); atomicAdd(&cs.stamps[26], 1ull); if (pre) atomicAdd(&cs.stamps[28], 1ull); }
#endif
                // ---- the workers' partial sums: tagged granules, fixed summation order; two rounds of loads are
                // ---- kept in flight so that a granule is seen at most half a round trip after it lands ----------
                const unsigned long long *pg = cs.pgran + (size_t)c * cs.n_wg * cs.pgran_stride;
                const int pgs = cs.pgran_stride;
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
                        if (k < cs.n_wg) { hi[b][j] = ld_agent(pg + (size_t)pgs * k); lo[b][j] = ld_agent(pg + (size_t)pgs * k + 1); }
                    }
                };
                auto complete = [&](int b) __attribute__((always_inline)) {
                    bool got = true;
#pragma unroll
                    for (int j = 0; j < kSweep; ++j)
                        if (j * 64 + lane < cs.n_wg) got = got && (unsigned)(hi[b][j] >> 32) == tag && (unsigned)(lo[b][j] >> 32) == tag;
                    return __all(got);
                };
#ifdef HTM_STAMPS
                unsigned long long rounds_ = 0;
#endif
                // A two-ahead order was answered an iteration ago: its granules are requested NOW, so that their round trip (the
                // workers sit on other XCDs: ~0.9 us through the fabric) runs under the evaluation of the wave's own event, and
                // looked at before anything else is requested.  Only if they are not all there -- or for the other kinds of
                // order, whose answers are still being worked out -- does the polling loop with two rounds in flight start.
                const bool early = pre_mode == 2;
                if (early) issue(0);
                // An order sent TWO iterations ahead is evaluated while the step in between (a hypocentre step of this
                // chain) may or may not have committed: the workers LEAVE THAT EVENT OUT, and this wave adds its misfit
                // -- at the position the event has now, under this step's proposed parameters -- itself, while it waits
                // for the workers' sums.  The result does not depend on when the workers looked.
                CSTAMP(5);   // with-job step: order recognised / sent
                double own_lane = 0.0;
                if constexpr (NCH > 0) {
                    if (pre_mode == 2 && prev_type >= 5) {
                        const int e = prev_evt - 1;
                        if (!(dhint && d_e == e)) {            // the hint missed: request the inputs now
                            const int vzd = opaque_zero();
                            const double *hyp = cs.xall + off_hy + c * nh + 3 * e;
                            d_ex = ld_state(hyp, vzd); d_ey = ld_state(hyp + 1, vzd); d_ez = ld_state(hyp + 2, vzd);
                            load_sta_regs<NCH>(st, f.S, lane, s_sx, s_sy, s_sz, tc, ac, 0, -1, 0.0);
                            load_obs_regs<NCH, F32>(ob, f, e, lane);
                        }
                        if (type == 2 || type == 4) {          // this step's proposed correction, on the lane of its station
#pragma unroll
                            for (int k = 0; k < NCH; ++k) {
                                if (lane + 64 * k == idx) { if (type == 2) st.tc[k] = x_new; else st.ac[k] = x_new; }
                            }
                        }
                        const double pxd[1] = {d_ex}, pyd[1] = {d_ey}, pzd[1] = {d_ez};
                        double outd[1];
                        event_misfit<NCH, 1, F32>(f, ob, lane, st, pxd, pyd, pzd, type == 1 ? x_new : beta, type == 3 ? x_new : q, outd);
                        own_lane = outd[0];
                    }
                }
                CSTAMP(6);   // with-job step: own event (two-ahead orders)
                bool have = false;
                if (early) have = complete(0);
                if (!have) {
                    issue(0);
                    for (;;) {
#ifdef HTM_STAMPS
                        rounds_ += 1;
#endif
                        issue(1);
                        if (complete(0)) { which = 0; break; }
                        issue(0);
                        if (complete(1)) { which = 1; break; }
                        if (__builtin_amdgcn_s_memrealtime() - t0 > 500000000ull) {
                            if (lane == 0) {
                                sh.c.err = -8;
                                unsigned long long *dg = cs.diag;          // what was waited for: the host puts it into its message
                                dg[1] = c; dg[2] = tag; dg[3] = pre; dg[4] = pre_mode; dg[5] = iter; dg[6] = p; dg[7] = type; dg[8] = idx;
                                dg[9] = hi[0][0]; dg[10] = lo[0][0]; dg[11] = first_pass; dg[12] = sh.start[c]; dg[0] = 1;
                            }
#ifdef HTM_STAMPS
                            if (lane == 0 && cs.stamps) {          // what was waited for (tools/diag_wait.py)
                                cs.stamps[100] = 1; cs.stamps[101] = c; cs.stamps[102] = tag; cs.stamps[103] = pre; cs.stamps[104] = pre_mode;
                                cs.stamps[105] = iter; cs.stamps[106] = p; cs.stamps[107] = type; cs.stamps[108] = idx;
                                cs.stamps[109] = hi[0][0]; cs.stamps[110] = lo[0][0]; cs.stamps[111] = sh.start[c];
                            }
#endif
                            break;
                        }
                    }
                }
#ifdef HTM_STAMPS
                if (stamp_me) {      // the wait by kind of order: [10..12] two-ahead {ticks, jobs, poll rounds}, [13..15] others
                    const unsigned long long n_ = __builtin_amdgcn_s_memtime();
                    const int b_ = pre_mode == 2 ? 10 : 13;
                    sh.stamp_acc[32 + b_] += n_ - t_last; sh.stamp_acc[32 + b_ + 1] += 1ull; sh.stamp_acc[32 + b_ + 2] += rounds_;
                }
#endif
                CSTAMP(7);   // with-job step: all granules there
#pragma unroll
                for (int j = 0; j < kSweep; ++j)            // fixed order: worker lane, lane + 64, ...
                    if (j * 64 < cs.n_wg)
                        part += (j * 64 + lane < cs.n_wg) ? (which == 0 ? gran_f64(hi[0][j], lo[0][j]) : gran_f64(hi[1][j], lo[1][j])) : 0.0;
                L_new = -wave_sum1(part + own_lane) - f.const_sum;       // cls_forward.f90:277-300
                if constexpr (!LOCK) acc = metropolis(L_new, L_cur, rT, lpr, r, logr) ? 1 : 0;
                CSTAMP(8);   // with-job step: sum + decision
#ifdef HTM_STAMPS
                if (stamp_me) sh.stamp_acc[32 + 9] += 1ull;
                if (lane == 0 && cs.stamps) atomicAdd(&cs.stamps[25], __builtin_amdgcn_s_memrealtime());
#endif
            }
        }
    }
    if (LOCK && xwait >= 0) {
        while (__hip_atomic_load(&sh.xdone, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < xwait) __builtin_amdgcn_s_sleep(1);
        if (sh.c.stop || sh.c.err) return kPassAbort;
        if (sh.base != base_used) return kPassRestart;
        T = sh.temp[c]; rT = sh.rtemp[c];
    }
    if constexpr (LOCK) { if (ok != 0) acc = metropolis(L_new, L_cur, rT, lpr, r, logr) ? 1 : 0; }   // cls_mcmc.f90:193-203
    // This wave's chain-state stores of EARLIER iterations have landed before it reaches barrier A of this one: role P
    // relies on it (role_prepublish_plan).  Waited for here, at the end of the pass -- every load of the step has been
    // consumed by now and those stores are a whole pass old, so the wait is free; at the top of the pass it waited for
    // the acknowledgements of the post phase's stores.
    if constexpr (PERSIST) drain_vmem();
    if (lane == 0) {
        const int cool = (T < 1.0 + kEps) ? 1 : 0;
        Proposal &pr = sh.prop[c];
        pr.type = type; pr.idx = idx; pr.evt = evt; pr.prior_ok = ok; pr.need_full = need_full; pr.accepted = acc;
        pr.cool = cool; pr.pad_ = 0;
        pr.x_new = x_new; pr.lpr = lpr; pr.r_judge = r; pr.logr_judge = logr; pr.L_new = L_new;
        pr.x_old = x_old; pr.L_old = L_cur;
        sh.start[c] = p; sh.cnt[c] = cnt;
        if (cool) sh.np[c * 7 + type - 1] += 1;                 // cls_mcmc.f90:186-189
        if (acc) {                                              // :207-219
            st_agent(cs.xall + o, x_new);
            if (o < rg.mir_n) rg.mx[o] = x_new;
            sh.L[c] = L_new;
            cs.L[c] = L_new;
            if (cool) sh.na[c * 7 + type - 1] += 1;
        }
    }
    CSTAMP(4);   // commit + LDS write-back
    return p + cnt;
}

// lane 0 of the owning wave: take back the speculative effects of chain c's step
__device__ __forceinline__ void undo_chain(CsRef cs_, StepShared &sh, const Ring &rg, int c)
{
    CsRef cs = rebase(cs_);
    Proposal &pr = sh.prop[c];
    if (pr.cool) sh.np[c * 7 + pr.type - 1] -= 1;
    if (pr.accepted) {
        const int om = elem_offset(cs, pr.type, c, pr.idx);
        st_agent(cs.xall + om, pr.x_old);
        if (om < rg.mir_n) rg.mx[om] = pr.x_old;
        sh.L[c] = pr.L_old;
        cs.L[c] = pr.L_old;
        if (pr.cool) sh.na[c * 7 + pr.type - 1] -= 1;
        pr.accepted = 0;
    }
}

// ---- between the two barriers of an iteration: three roles on three waves (lanes <-> chains) ----------
// Each role recomputes the cheap validation and reads only data no other role writes in this interval.
struct Valid {
    unsigned long long bad, mf;
    int total, base;
};
template <bool PERSIST>
__device__ __forceinline__ Valid validate(const StepShared &sh, int nc, int lane)
{
    Valid v;
    const bool in = lane < nc;
    const int my_cnt = in ? sh.cnt[lane] : 0, my_start = in ? sh.start[lane] : 0;
    const int incl = wave_incl_scan(my_cnt);
    v.base = sh.start[0];                       // chain 0 always starts at the iteration's base
    v.bad = __ballot(in && v.base + incl - my_cnt != my_start);
    v.mf = PERSIST ? 0ull : __ballot(in && sh.prop[lane].need_full != 0);   // PERSIST: already evaluated
    v.total = __builtin_amdgcn_readlane(incl, 63);
    return v;
}

// role R: record slots (hypo_tremor_mcmc.f90:270-280), step log, per-chain part of the swap record
__device__ __forceinline__ void role_records(CsRef cs_, StepShared &sh, int iter, bool lockstep, int lane, double *rec)
{
    CsRef cs = rebase(cs_);
    const int nc = cs.n_chains;
    const bool in = lane < nc;
    const int c = in ? lane : 0;
    const double T = sh.temp[c], L_post = sh.L[c];
    const bool rec_l = in && T < 1.0 + kEps && (iter % cs.n_interval) == 1;
    const bool rec_s = rec_l && iter > cs.n_burn;
    const unsigned long long ml = __ballot(rec_l), ms = __ballot(rec_s);
    const unsigned long long below = (1ull << lane) - 1ull;
    if (in) {
        sh.slot_l[c] = rec_l ? sh.c.n_lik + __popcll(ml & below) : -1;
        sh.slot_s[c] = rec_s ? sh.c.n_smp + __popcll(ms & below) : -1;
        const int row = sh.c.slog_n + c;
        if (row < sh.c.slog_cap) {
            const Proposal pr = sh.prop[c];
            int32_t *ir = cs.slog_i + 8 * (size_t)row;
            double *dr = cs.slog_d + 4 * (size_t)row;
            ir[0] = iter; ir[1] = c; ir[2] = pr.type; ir[3] = pr.idx + 1; ir[4] = pr.prior_ok;
            ir[5] = pr.accepted; ir[6] = pr.need_full; ir[7] = 0;
            dr[0] = pr.x_new; dr[1] = pr.L_new; dr[2] = L_post; dr[3] = T;
        }
        if (lockstep) { rec[4 + 2 * c] = T; rec[5 + 2 * c] = L_post; }
    }
    if (lane == 0) {
        sh.c.n_lik += __popcll(ml); sh.c.n_smp += __popcll(ms);
        if (sh.c.slog_n < sh.c.slog_cap) sh.c.slog_n += nc;
    }
}

// role W: the temperature swap of cls_parallel.f90:121-136 + :285-302 (single rank) or the header of this
// rank's record (lock-step).  Uniform over the wave; lane 0 writes.
__device__ __forceinline__ void role_swap(CsRef cs_, StepShared &sh, int iter, bool lockstep, int lane,
                                          int i1, int i2, double sr, double slr, double *rec)
{
    CsRef cs = rebase(cs_);
    if (lane != 0) return;
    sh.sw_do = 0;
    if (lockstep) {
        rec[0] = (double)i1; rec[1] = (double)i2; rec[2] = sr; rec[3] = (double)iter;
    } else if (cs.n_procs * cs.n_chains > 1) {
        const double T1 = sh.temp[i1], T2 = sh.temp[i2];
        const double del_s = (sh.L[i2] - sh.L[i1]) * (1.0 / T1 - 1.0 / T2);
        if (sr >= kEps && slr <= del_s) {       // applied after the barrier by the waves owning the chains
            sh.sw_do = 1; sh.sw_c1 = i1; sh.sw_c2 = i2; sh.sw_T1 = T2; sh.sw_T2 = T1;
        }
    }
}

// select_pair + the judge_swap draw starting at relative position `end` (uniform over the wave).
// Returns false if the look-ahead window does not cover it (p < 1e-40).
__device__ __forceinline__ bool swap_plan(CsRef cs_, const StepShared &sh, const Ring &rg, bool lockstep,
                                          int end, int &pos_out, int &i1, int &i2, double &sr, double &slr)
{
    CsRef cs = rebase(cs_);
    const int nc = cs.n_chains, n_all = cs.n_procs * nc;
    int pos = end;
    i1 = -1; i2 = -1; sr = 0.0; slr = 0.0;
    const int limit = sh.fill - 2;
    if (n_all > 1) {
        if (cs.rank == 0) {
            const int4 sw = rg.sw[pos & rg.mask];           // precomputed in the stream (htm_stream.hpp)
            if (sw.z > 0) { i1 = sw.x; i2 = sw.y; pos += sw.z; }
            else {                                          // > 12 redraws: follow the stream here
                i1 = (int)(rg.U[pos & rg.mask] * cs.n_procs * nc);
                pos++;
                for (;;) {
                    if (pos >= limit) return false;
                    i2 = (int)(rg.U[pos & rg.mask] * cs.n_procs * nc);
                    pos++;
                    if (i1 != i2) break;
                }
            }
        }
        if (pos >= limit) return false;
        sr = rg.U[pos & rg.mask]; slr = rg.LOGU[pos & rg.mask];
        if (!lockstep) pos++;                               // single rank: this rank is always rank1
    }
    pos_out = pos;
    return true;
}

// role P (k_mcmc), lanes <-> chains: the next iteration's steps are already decoded in the stream window, so a
// chain whose NEXT step needs the full evaluation (types 1..4) can have its order out now -- one roles phase,
// one post phase and one step front (~2 us) before its own wave would send it.  Only for chains that did not
// commit in this iteration: their state stores have provably landed (every wave drains before the commit of its pass).
// The chain wave recognises the order by its start position and goes straight to collecting the partial sums.
// start of the chain step n steps after the one that starts at pos (hop tables cover kHops steps at a time)
__device__ __forceinline__ int hop_ahead(const Ring &rg, int pos, int n)
{
    int p = pos;
    while (n > kHops) { p += rg.hop[(p & rg.mask) * kHops + kHops - 1]; n -= kHops; }
    if (n > 0) p += rg.hop[(p & rg.mask) * kHops + n - 1];
    return p;
}

struct PreOrder {            // per lane (<-> chain)
    bool job;
    int c;
    unsigned tag, w1, x_hi, x_lo, co, c_hi, c_lo, rep;
};
// For every chain (lane): the order of its step of iteration iter + 1 if that step needs the full evaluation (mode 1);
// else -- that step being a hypocentre step -- the order of its step of iteration iter + 2 if THAT one does (mode 2):
// sent two iterations ahead, the workers' round trip disappears behind a whole iteration.  A mode-2 order is
// evaluated on the state that memory holds when the workers get to it, before or after the step in between commits;
// the order names that step's element: the workers leave that event out of their sums and the chain's own wave adds it
// (chain_pass), so the result does not depend on when the workers looked.
__device__ __forceinline__ PreOrder role_prepublish_plan(CsRef cs_, StepShared &sh, const Ring &rg, int iter,
                                                         int pos, int lane, bool allow2, bool lockstep)
{
    CsRef cs = rebase(cs_);
    PreOrder po;
    po.job = false; po.c = 0; po.tag = 0; po.w1 = 0; po.x_hi = 0; po.x_lo = 0; po.co = 0xffffffffu; po.c_hi = 0; po.c_lo = 0;
    po.rep = 0;
    const int nc = cs.n_chains, M = rg.mask, S_ = cs.S;
    if (rg.mir_n == 0 || iter + 1 > sh.c.iter_target) return po;        // no mirror / no next iteration in this launch
    const bool in = lane < nc;
    const int c = in ? lane : 0;
    const bool clean = in && sh.redone[c] == 0;
    if (lane < nc) sh.redone[lane] = 0;
    // ---- step A: this chain's step of iteration iter + 1 (positions validated up to `pos`)
    const int pA = hop_ahead(rg, pos, c);
    const bool winA = pA + 8 < sh.fill;
    const int4 dA = rg.dec[pA & M];
    const int par1 = (iter + 1) & 1, par2 = iter & 1;
    // A two-ahead order for step A is out already -- unless the step in between (the one that just ended) was not the
    // one it was written around: a repeated pass can start that step elsewhere and still leave step A where it was
    // (the draw counts cancel).  Such an order is dropped: its report names the wrong element, and if the step in
    // between was itself a full evaluation the workers' sums have been overwritten.
    const bool stale2 = in && sh.pre_p[par1][c] == pA && sh.pre_mode[par1][c] == 2 && sh.pre_pa[par1][c] != sh.start[c];
    const bool haveA = sh.pre_p[par1][c] == pA && !stale2;               // sent two iterations ahead already
    // ---- step B: its step of iteration iter + 2, if every step of iteration iter + 1 draws its usual randoms
    int pb0 = hop_ahead(rg, pos, nc);
    bool goodB = winA;
    if (cs.n_procs * nc > 1) {
        const int4 sw = rg.sw[pb0 & M];
        if (!lockstep) {                                  // single rank: select_pair + judge_swap both draw here
            goodB = goodB && sw.z > 0;
            pb0 += sw.z + 1;
        } else if (cs.rank == 0) {                        // rank 0 draws the pair; the judge draw is rank1's (cls_parallel.f90:163)
            goodB = goodB && sw.z > 0;
            pb0 += sw.z + ((sw.x / nc) == 0 ? 1 : 0);
        }
        // other ranks draw nothing for the pair, and the judge draw only if they own chain 1 of it (1 time in n_procs):
        // the bet is "not"; a lost bet shows as an order written for the wrong position, which is taken back (void_slot)
    }
    const int pB = hop_ahead(rg, pb0, c);
    goodB = goodB && pb0 >= pos && pB >= pb0 && pB + 8 < sh.fill && iter + 2 <= sh.c.iter_target;
    const int4 dB = rg.dec[pB & M];
    const bool jobA = dA.x >= 1 && dA.x <= 4, jobB = dB.x >= 1 && dB.x <= 4;
    int mode = 0;
    if (clean && winA && jobA && !haveA) mode = 1;
    else if (allow2 && clean && goodB && !jobA && jobB) mode = 2;
    const int type = mode == 2 ? dB.x : dA.x, idx = mode == 2 ? dB.y : dA.y, pJ = mode == 2 ? pB : pA;
    const int goff = type == 1 ? 0 : type == 2 ? nc : type == 3 ? nc + nc * S_ : 2 * nc + nc * S_;
    const int gnx = (type == 1 || type == 3) ? 1 : S_;
    const int o = mode ? goff + c * gnx + idx : 0;
    const double x_old = rg.mx[o];                                      // LDS mirror: no memory round trip here
    double step;
    if (rg.mir_steps) step = rg.mstep[o];
    else step = mode ? cs.stall[o] : 0.0;                               // immutable; only where the mirror of x took the room
    const double x_new = x_old + rg.pg[pJ & M] * step;                  // cls_model.f90:172, as chain_pass computes it
    if (mode && cs.rayleigh14) {                                        // a Rayleigh prior among vs/qs/corrections (:178-187)
        if (cs.ptall[o] == 1 && x_new <= cs.muall[o]) mode = 0;         // prior rejects: no evaluation
    }
    // the commit this chain made in the iteration that just ended may still be on its way to memory: the workers
    // wait until they read that value back
    const Proposal pr = sh.prop[c];
    const int ct = pr.type;
    const int o_hy = 2 * nc + 2 * nc * S_ + c * 3 * cs.E;
    const int cgoff = ct == 1 ? 0 : ct == 2 ? nc : ct == 3 ? nc + nc * S_ : ct == 4 ? 2 * nc + nc * S_ : 2 * nc + 2 * nc * S_;
    const int cgnx = (ct == 1 || ct == 3) ? 1 : (ct == 2 || ct == 4) ? S_ : 3 * cs.E;
    const unsigned long long xb = (unsigned long long)__double_as_longlong(x_new);
    const unsigned long long cb = (unsigned long long)__double_as_longlong(pr.x_new);
    // two-ahead: the workers wait for the commit of the iteration that just ended by reading its value back; the step in
    // between must not be able to overwrite that very element before they look
    if (mode == 2 && pr.accepted && cgoff + c * cgnx + pr.idx == o_hy + dA.y) mode = 0;
    po.job = mode != 0; po.c = c;
    // own tag space (chain_pass tags stay below 2^31); a step can get a two-ahead order AND, if that one turned out to
    // be addressed to the wrong position, a one-ahead order: the two must not share a tag
    po.tag = 0x80000000u | (mode == 2 ? 0x40000000u : 0u) | (((unsigned)(iter + mode) & 0x01ffffffu) << 5) | (unsigned)c;
    po.w1 = (unsigned)type | ((unsigned)idx << 3);
    po.x_hi = (unsigned)(xb >> 32); po.x_lo = (unsigned)xb;
    po.co = pr.accepted ? (unsigned)(cgoff + c * cgnx + pr.idx) : 0xffffffffu;
    po.c_hi = (unsigned)(cb >> 32); po.c_lo = (unsigned)cb;
    po.rep = mode == 2 ? (unsigned)(o_hy + dA.y) + 1u : 0u;            // element of the step in between (+1; 0 = none)
    // Orders still on the books that no step will use: (a) addressed to iteration iter + 1 but written for another start
    // position (a Rayleigh rejection shifted the stream) or around another step in between (stale2); (b) addressed to the
    // iteration that just ended and never taken (its step started elsewhere and was a partial update).  The workers are
    // told (void_slot) unless a new order of this chain overwrites the slot right now.
    // (Orders are rare -- one step in ten -- so the books are usually empty: one ballot skips all of this.)
    const int bookA = in ? sh.pre_p[par1][c] : -1, bookB = in ? sh.pre_p[par2][c] : -1;
    if (__builtin_expect(__ballot(bookA != -1 || bookB != -1) != 0ull, 0)) {
        const bool staleA = bookA != -1 && (bookA != pA || stale2);
        const bool deadB = bookB != -1;
        if (staleA && mode != 1) sh.pre_p[par1][lane] = -1;           // (a one-ahead order replaces the entry)
        if (deadB && mode != 2) sh.pre_p[par2][lane] = -1;
        if ((staleA || deadB) && mode == 0 && !(haveA && in)) void_slot(cs, c);   // haveA: the slot holds a live order (written after the dead one)
    }
    if (lane < nc && mode) {
        const int par = mode == 2 ? par2 : par1;
        sh.pre_p[par][lane] = pJ; sh.pre_tag[par][lane] = po.tag; sh.pre_mode[par][lane] = mode; sh.pre_pa[par][lane] = pA;
    }
    return po;
}
__device__ __forceinline__ void role_prepublish_send(CsRef cs_, const PreOrder &po, unsigned long long launch)
{
    CsRef cs = rebase(cs_);
    if (!po.job) return;
    for (int r = 0; r < cs.slot_rep; ++r) {
        unsigned long long *sl = cs.slots + (size_t)r * cs.slot_stride + po.c * kGranPerSlot;
        st_gran(sl + 0, po.tag, (unsigned)launch);
        st_gran(sl + 1, po.tag, po.w1);
        st_gran(sl + 2, po.tag, po.x_hi);
        st_gran(sl + 3, po.tag, po.x_lo);
        st_gran(sl + 4, po.tag, po.co);
        st_gran(sl + 5, po.tag, po.c_hi);
        st_gran(sl + 6, po.tag, po.c_lo);
        st_gran(sl + 7, po.tag, po.rep);
    }
}

// chain wave, after the second barrier: this iteration's swap (if it touches chain c) and its records
__device__ __forceinline__ void post_chain(CsRef cs_, StepShared &sh, int c, int iter, int lane)
{
    CsRef cs = rebase(cs_);
    const int sl = sh.slot_l[c], ss = sh.slot_s[c];
    if (lane == 0) {
        if (sh.sw_do) {     // cls_parallel.f90:131-136
            if (c == sh.sw_c1) { sh.temp[c] = sh.sw_T1; sh.rtemp[c] = 1.0 / sh.sw_T1; cs.temp[c] = sh.sw_T1; }
            if (c == sh.sw_c2) { sh.temp[c] = sh.sw_T2; sh.rtemp[c] = 1.0 / sh.sw_T2; cs.temp[c] = sh.sw_T2; }
        }
        if (sl >= 0) { cs.lik_iter[sl] = iter; cs.lik_chain[sl] = c; cs.lik_val[sl] = sh.L[c]; }
    }
    if (ss >= 0) {
        const GroupOff go = group_offsets(cs);
        const int nh = go.nh, S = cs.S, rec = nh + 2 * S + 2;
        double *dst = cs.smp_data + (size_t)ss * rec;
        const double *hx = cs.xall + go.hy + (size_t)c * nh;
        const int vz = opaque_zero();
        for (int k = lane; k < nh; k += 64) dst[k] = ld_state(hx + k, vz);
        for (int k = lane; k < S; k += 64) {
            dst[nh + k] = ld_state(cs.xall + go.tc + (size_t)c * S + k, vz);
            dst[nh + S + k] = ld_state(cs.xall + go.ac + (size_t)c * S + k, vz);
        }
        if (lane == 0) {
            dst[nh + 2 * S] = ld_state(cs.xall + c, vz);
            dst[nh + 2 * S + 1] = ld_state(cs.xall + go.qs + c, vz);
            cs.smp_iter[ss] = iter; cs.smp_chain[ss] = c;
        }
    }
}

// thread 0: temperature swap between chains of any two ranks from the all-gathered records
// (cls_parallel.f90:118-213, :285-302).  Every rank evaluates the same decision from the same records.
__device__ __forceinline__ void apply_swap(CsRef cs_, StepShared &sh, const double *gathered)
{
    CsRef cs = rebase(cs_);
    const int nc = cs.n_chains, RW = 4 + 2 * nc;
    const int iter = sh.c.iter_done + 1;
    if (cs.n_procs * nc > 1) {
        for (int r = 0; r < cs.n_procs; ++r)
            if ((int)gathered[(size_t)r * RW + 3] != iter) sh.c.err = -6;
        const int i1 = (int)gathered[0], i2 = (int)gathered[1];
        const int rank1 = i1 / nc, chain1 = i1 % nc, rank2 = i2 / nc, chain2 = i2 % nc;
        const double T1 = gathered[(size_t)rank1 * RW + 4 + 2 * chain1];
        const double L1 = gathered[(size_t)rank1 * RW + 5 + 2 * chain1];
        const double T2 = gathered[(size_t)rank2 * RW + 4 + 2 * chain2];
        const double L2 = gathered[(size_t)rank2 * RW + 5 + 2 * chain2];
        const double r = gathered[(size_t)rank1 * RW + 2];
        const double del_s = (L2 - L1) * (1.0 / T1 - 1.0 / T2);
        bool acc = false;
        if (r >= kEps) { if (log(r) <= del_s) acc = true; }
        if (acc) {
            if (cs.rank == rank1) { cs.temp[chain1] = T2; sh.temp[chain1] = T2; sh.rtemp[chain1] = 1.0 / T2; }
            if (cs.rank == rank2) { cs.temp[chain2] = T1; sh.temp[chain2] = T1; sh.rtemp[chain2] = 1.0 / T1; }
        }
        if (cs.rank == rank1) sh.c.spos += 1;       // judge_swap's rand_u() came from rank1's stream
    }
    sh.c.iter_done = iter;
    sh.c.stage = ST_IDLE;
}

// MODE_LOCKRUN, wave 0, after the roles of iteration `iter`: this rank's swap record goes into EVERY rank's inbox
// (its own included) as tagged 8-byte granules {iteration : 32, payload : 32} written with system-scope stores -- over
// xGMI into the peers' memory, no collective, no kernel boundary --, then the wave polls its own inbox until all
// n_procs records of this iteration are complete, stages them in LDS as the doubles apply_swap reads, and lane 0
// decides the swap exactly as every other rank does (cls_parallel.f90:118-213).  Record = the (4 + 2 n_chains) words
// of the all-gather protocol + one control word: bit 0 = this rank asks everybody to stop after this iteration
// (record buffers or produced random stream nearly used up: the host drains / refills and launches again), bit 1 = it
// has failed.  Inbox parity = iter & 1: a rank can be at most one iteration ahead of the slowest one.
constexpr int kXLoads = 8;        // granule loads per lane in flight while polling

// every thread, after the roles of an iteration: does this rank ask everybody to stop after it?
__device__ __forceinline__ bool want_stop_now(CsRef cs_, const StepShared &sh, int wmax)
{
    CsRef cs = rebase(cs_);
    const int nc = cs.n_chains;
    return sh.c.n_lik + 2 * nc > cs.cap_lik || sh.c.n_smp + 2 * nc > cs.cap_smp || sh.avail < sh.base + 3 * wmax;
}

// one wave: this rank's record of iteration `iter` into every rank's inbox
__device__ __forceinline__ void exchange_post(CsRef cs_, StepShared &sh, int iter, int lane, bool want_stop)
{
    CsRef cs = rebase(cs_);
    const int nc = cs.n_chains, RW = 4 + 2 * nc, G = cs.xg, np = cs.n_procs, par = iter & 1;
    const unsigned tag = (unsigned)iter;
    const unsigned ctl = (want_stop ? 1u : 0u) | (sh.c.err ? 2u : 0u);
    if (lane == 0) { sh.xctl = ctl; sh.xctl_iter = iter; }      // (this rank's own record is taken from LDS when the records are collected)
    if (__builtin_expect(cs.dbg_xfail_iter > 0 && iter >= cs.dbg_xfail_iter, 0)) return;      // (fault injection: the peers never see this record)
    for (int g0 = 0; g0 < G; g0 += 64) {
        const int g = g0 + lane;
        if (g < G) {
            const int w = g >> 1;
            unsigned pay;
            if (w < RW) {
                const unsigned long long b = (unsigned long long)__double_as_longlong(sh.xrec[w]);
                pay = (g & 1) ? (unsigned)b : (unsigned)(b >> 32);
            } else pay = (g & 1) ? 0u : ctl;
            const unsigned long long v = ((unsigned long long)tag << 32) | pay;
            for (int q = 0; q < np; ++q) st_sys(ld_const(cs.outbox + q) + (size_t)(par * np + cs.rank) * G + g, v);
        }
    }
}

// the same wave, later: wait for the n_procs records of iteration `iter`, decide the swap, publish sh.xdone = iter.
// Rows of 64 granules, row = (rank r, chunk k of its record); kXLoads rows in flight.
__device__ __forceinline__ void exchange_finish(CsRef cs_, StepShared &sh, double *s_gath, int iter, int lane, bool publish = true)
{
    CsRef cs = rebase(cs_);
    const int nc = cs.n_chains, RW = 4 + 2 * nc, G = cs.xg, np = cs.n_procs, par = iter & 1;
    const unsigned tag = (unsigned)iter;
    const unsigned long long *in = cs.inbox + (size_t)par * np * G;
    unsigned *gu = reinterpret_cast<unsigned *>(s_gath);
    const int chunks = (G + 63) >> 6, rows = np * chunks;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();       // 100 MHz
    unsigned anyctl = 0;
    bool dead = false;
    // This rank's own record does not travel: it is read from LDS where exchange_post took it from (sh.xrec, sh.xctl) -- the round
    // trip of the rank's own stores through fine-grained memory was on the path of every lock-step iteration (HTM_XOWN=0 restores it).
    const bool own_lds = cs.xown != 0 && sh.xctl_iter == iter;
    for (int row0 = 0; row0 < rows && !dead; row0 += kXLoads) {
        unsigned long long v[kXLoads];
        for (;;) {
            bool ok = true;
#pragma unroll
            for (int j = 0; j < kXLoads; ++j) {
                const int row = row0 + j, r = row / chunks, k = (row - r * chunks) * 64 + lane;
                v[j] = (unsigned long long)tag << 32;
                if (row < rows && k < G) {
                    if (own_lds && r == cs.rank) {
                        const int w = k >> 1;
                        unsigned pay;
                        if (w < RW) {
                            const unsigned long long b = (unsigned long long)__double_as_longlong(sh.xrec[w]);
                            pay = (k & 1) ? (unsigned)b : (unsigned)(b >> 32);
                        } else pay = (k & 1) ? 0u : sh.xctl;
                        v[j] = ((unsigned long long)tag << 32) | pay;
                    } else v[j] = ld_sys(in + (size_t)r * G + k);
                }
            }
#pragma unroll
            for (int j = 0; j < kXLoads; ++j) ok = ok && (unsigned)(v[j] >> 32) == tag;
            if (__all(ok)) break;
            if (__builtin_amdgcn_s_memrealtime() - t0 > cs.xwait_ticks) {       // 20 s: a peer is gone
                if (lane == 0) sh.c.err = -10;
                dead = true;
                break;
            }
        }
        if (dead) break;
#pragma unroll
        for (int j = 0; j < kXLoads; ++j) {
            const int row = row0 + j, r = row / chunks, k = (row - r * chunks) * 64 + lane;
            if (row < rows && k < G) {
                const int w = k >> 1;
                if (w < RW) gu[2 * (r * RW + w) + ((k & 1) ? 0 : 1)] = (unsigned)v[j];      // little-endian halves of the double
                else if (!(k & 1)) anyctl |= (unsigned)v[j];
            }
        }
    }
    const bool stop_any = __ballot((anyctl & 1u) != 0) != 0ull, err_any = __ballot((anyctl & 2u) != 0) != 0ull;
    if (lane == 0) {
        if (!dead && sh.c.err == 0 && sh.c.stage == ST_WAIT_SWAP) {
            apply_swap(cs, sh, s_gath);                         // iteration done; rank1 consumed its judge_swap draw
            sh.base = (int)(sh.c.spos - sh.origin);
        }
        if (stop_any && sh.c.stop == 0) sh.c.stop = 3;          // everybody leaves after this iteration
        if (err_any && sh.c.err == 0) sh.c.err = -11;           // a peer reported a failure
        if (publish) __hip_atomic_store(&sh.xdone, iter, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

#ifdef HTM_STAMPS
#define STAMP(k)                                                                                   \
    do {                                                                                           \
        if (tid == 0 && cs.stamps) {                                                               \
            const unsigned long long now_ = __builtin_amdgcn_s_memtime();                          \
            sh.stamp_acc[k] += now_ - stamp_last_;   /* LDS: a global RMW here would bill ~1 k cycles to the next phase */ \
            stamp_last_ = now_;                                                                    \
        }                                                                                          \
    } while (0)
#else
#define STAMP(k) do { } while (0)
#endif

// PERSIST = true: this is block 0 of a k_mcmc launch; full evaluations are handed to the worker blocks of the
// same launch through PSync (no kernel exit).  PERSIST = false: the kernel exits at a hand-over and k_full runs
// as its own launch (fallback path, also used for profiling the two stages separately).
template <int NCH, bool PERSIST, bool F32, int MK>
__device__ __forceinline__ void step_body(FwRef f_, CsRef cs_, int mode, int target_arg,
                                          const double *gathered, int ring_size, int wmax, unsigned long long launch)
{
    CsRef cs = rebase(cs_);
    FwRef f = rebase(f_);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    StepShared &sh = *reinterpret_cast<StepShared *>(smem);
    char *carve = smem + ((sizeof(StepShared) + 15) & ~size_t(15));
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
    double *s_gath = s_sz + f.S;                   // [kGathStage] the all-gathered swap records, staged
    rg.mir_n = (PERSIST && (MK == 0 || MK == 2 || (MK < 0 && mode == MODE_RUN))) ? cs.mirror_n : 0;
    rg.mx = s_gath + kGathStage;
    rg.mstep = rg.mx + rg.mir_n;
    rg.mir_steps = cs.mirror_steps != 0;
    rg.lock = 0;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NW = blockDim.x >> 6;                // every wave is a chain wave
    const int nc = cs.n_chains;
    // MK: which main loop this instantiation is -- 0 the single-rank loop (MODE_RUN), 1 one lock-step iteration per launch
    // (MODE_ADVANCE / _FINISH / _APPLY), 2 persistent lock-step (MODE_LOCKRUN), -1 decided at run time (k_step)
    constexpr bool LOCK = MK == 2;
    const bool lockstep = MK < 0 ? (mode != MODE_RUN) : (MK != 0);
    // MODE_LOCKRUN: lock-step with the swap records exchanged inside the launch (exchange_post / exchange_finish) -- the kernel stays
    // resident over the iterations, so the orders of role P and the LDS state carry over as in the single-rank loop
    constexpr bool lockrun = PERSIST && LOCK;          // (the host launches the LOCK instantiation with MODE_LOCKRUN only)
    const bool rolep_on = PERSIST && (!lockstep || lockrun);
    double *rec = lockrun ? sh.xrec : cs.swap_rec;
    // helper roles between the barriers: validation+bookkeeping on wave 0, records and swap on other waves
    const int wave_R = NW > 1 ? 1 : 0, wave_W = NW > 2 ? 2 : 0;
    const int wave_P = NW > 4 ? 4 : NW - 1;        // extends the LDS window during the roles phase (no role of its own if NW > 4)
    // role P (orders of the coming iterations' full evaluations): a wave without a chain if there is one, else an older
    // wave (it has the slack before its own step)
    const int wave_Q = nc < NW ? NW - 1 : NW > 3 ? 3 : 0;
#ifdef HTM_STAMPS
    unsigned long long stamp_last_ = __builtin_amdgcn_s_memtime();
#endif

    // Everything the launch needs from memory is requested in ONE round of vector loads (a scalar load of the
    // control block alone costs a cold K$ + L2 miss of its own): control block, stream coverage, station table,
    // temperatures, gathered records.
    {
        const int vz0 = opaque_zero();
        constexpr int kCtrlWords = (int)(sizeof(Ctrl) / sizeof(int));
        if (tid < kCtrlWords) reinterpret_cast<int *>(&sh.c)[tid] = reinterpret_cast<const int *>(cs.ctrl)[tid + vz0];
        if (tid == kCtrlWords) sh.hop_end = cs.stream.hop_end[vz0];
    }
    for (int j = tid; j < f.S; j += blockDim.x) { s_sx[j] = f.sx[j]; s_sy[j] = f.sy[j]; s_sz[j] = f.sz[j]; }
    for (int c = tid; c < nc; c += blockDim.x) { sh.temp[c] = cs.temp[c]; sh.rtemp[c] = 1.0 / cs.temp[c]; sh.L[c] = cs.L[c]; }
    for (int k = tid; k < 7 * nc; k += blockDim.x) { sh.np[k] = 0; sh.na[k] = 0; }
#ifdef HTM_STAMPS
    for (int k = tid; k < 96; k += blockDim.x) sh.stamp_acc[k] = 0ull;
#endif
    for (int k = tid; k < rg.mir_n; k += blockDim.x) { rg.mx[k] = cs.xall[k]; if (rg.mir_steps) rg.mstep[k] = cs.stall[k]; }
    // the gathered records of the previous lock-step iteration come in with the same round of loads
    const bool do_apply = (MK == 1 || MK < 0) && (mode == MODE_APPLY || mode == MODE_ADVANCE) && gathered != nullptr;
    const int n_gath = cs.n_procs * (4 + 2 * nc);
    const bool staged = do_apply && n_gath <= kGathStage;
    if (staged)
        for (int k = tid; k < n_gath; k += blockDim.x) s_gath[k] = gathered[k];
    __syncthreads();

    // ---------------- swap of the previous lock-step iteration (cls_parallel.f90:118-213) --------------
    // MODE_APPLY does only this; MODE_ADVANCE does it first when the host passes the gathered records along
    if (tid == 0) {
        if (target_arg >= 0) sh.c.iter_target = target_arg;
        sh.origin = sh.c.spos;
        const long long av = sh.hop_end - sh.c.spos;
        sh.avail = av > (1 << 30) ? (1 << 30) : (int)av;
        sh.fill = 0; sh.base = 0; sh.redo = -1; sh.sw_do = 0; sh.catchup = 0; sh.rolep_iter = -1;
        sh.xdone = sh.c.iter_done; sh.xstop = 0; sh.xspec = 0; sh.xctl = 0u; sh.xctl_iter = -1;
    }
    for (int c = tid; c < kMaxChains; c += blockDim.x) { sh.pre_p[0][c] = -1; sh.pre_p[1][c] = -1; sh.redone[c] = 0; }
    {
        if (do_apply && tid == 0 && sh.c.stage == ST_WAIT_SWAP && sh.c.err == 0) {
            apply_swap(cs, sh, staged ? s_gath : gathered);
            const int delta = (int)(sh.c.spos - sh.origin);     // rank1 consumed its judge_swap draw
            sh.origin = sh.c.spos;
            sh.avail -= delta;
            if (mode == MODE_APPLY) *cs.ctrl = sh.c;
        }
        __syncthreads();
    }
    if (mode == MODE_APPLY) return;

    bool resume = !PERSIST && (sh.c.stage == ST_WAIT_FULL);     // (k_mcmc evaluates in place: it never waits for k_full)
    if (mode == MODE_FINISH && !resume) return;
    if (sh.c.stage == ST_WAIT_SWAP) return;                   // nothing to do until the swap is applied
    if (!resume && (sh.c.iter_done >= sh.c.iter_target || sh.c.stop || sh.c.err)) return;
    if (!resume && sh.avail < wmax) {                         // the produced stream does not cover an iteration
        __syncthreads();
        if (tid == 0) { if (lockstep) sh.c.err = -7; else sh.c.stop = 2; *cs.ctrl = sh.c; }
        return;
    }
    STAMP(0);   // prologue

    // ---------------- P0: judge + commit of the chains that came back from k_full; first stream window ----
    if (!PERSIST && __builtin_expect(resume, 0)) {
        if (tid == 0) cs.desc->n = 0;                          // work order consumed
        for (int c = wave; c < nc; c += NW) {
            Proposal pr = cs.prop[c];
            if (pr.need_full) {
                double acc = 0.0;
                for (int k = lane; k < cs.n_wg; k += 64) acc += cs.partial[(size_t)c * cs.n_wg + k];
                pr.L_new = -wave_sum1(acc) - f.const_sum;      // cls_forward.f90:277-300
                pr.accepted = metropolis(pr.L_new, sh.L[c], sh.rtemp[c], pr.lpr, pr.r_judge, pr.logr_judge) ? 1 : 0;
                if (lane == 0 && pr.accepted) {                // cls_mcmc.f90:207-219
                    st_agent(cs.xall + elem_offset(cs, pr.type, c, pr.idx), pr.x_new);
                    sh.L[c] = pr.L_new;
                    cs.L[c] = pr.L_new;
                    if (pr.cool) sh.na[c * 7 + pr.type - 1] += 1;
                }
            }
            if (lane == 0) sh.prop[c] = pr;
        }
    }
    prefetch_all(cs, sh, rg, 2 * wmax);           // ends with a barrier
    STAMP(1);   // P0

    bool have_prev = false;          // an iteration has completed in this launch (role P has something to look at)
    int fill_next = 0;               // wave_P: window extent staged in the last roles phase, published before the next barrier A
    int it_local = sh.c.iter_done;   // iterations through their roles in this rank (MODE_LOCKRUN: the last one's swap may be pending)
    bool xpend = false;              // MODE_LOCKRUN: the records of iteration it_local are posted, its swap not yet decided
    bool leave = false;
    for (;;) {
        const int iter = it_local + 1;
        if (__builtin_expect(!resume, 1)) {
            // ---------------- anything left to do? ---------------------------------------------------
            // (xpend: iteration `iter` starts on the bet that the job goes on -- role V made sure that neither the target
            // nor this rank's buffers end it; sh.c is being updated by the wave that decides the swap, nobody reads it here)
            if (!(lockrun && xpend) && (sh.c.iter_done >= sh.c.iter_target || sh.c.stop || sh.c.err)) break;
            if (sh.c.n_lik + nc > cs.cap_lik || sh.c.n_smp + nc > cs.cap_smp) {
                __syncthreads();
                if (tid == 0) { if (lockstep) sh.c.err = -5; else sh.c.stop = 1; }
                __syncthreads();
                break;
            }
            if (sh.avail < sh.base + wmax) {                  // produced stream exhausted: the host refills
                __syncthreads();
                if (tid == 0) { if (lockstep) sh.c.err = -7; else sh.c.stop = 2; }
                __syncthreads();
                break;
            }
            if (__builtin_expect(sh.catchup != 0, 0)) prefetch_all(cs, sh, rg, sh.base + 2 * wmax);   // rare; flag is uniform (set between barriers)
            if constexpr (PERSIST) {
                // ---------------- role P: orders of the next two iterations' full evaluations -----------------------
                // On wave_Q, before its own chain step (an older wave: it has the slack; or a wave without a chain), from the validated base of THIS
                // iteration (= the iteration that just ended is iter - 1).  The other waves look at what it leaves only
                // after their step's arithmetic (chain_pass waits on sh.rolep_iter), by when it is long done.
                if (have_prev && rolep_on && wave == wave_Q) {
                    role_prepublish_send(cs, role_prepublish_plan(cs, sh, rg, iter - 1, sh.base, lane, HTM_ALLOW2 && NCH > 0, lockstep), launch);
                    if (lane == 0) __hip_atomic_store(&sh.rolep_iter, iter, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
            // ---------------- passes: propose -> partial update -> decide -> commit, per chain wave ----
            int redo = 0;
            bool first = true;
#ifdef HTM_STAMPS
            const unsigned long long t_top = __builtin_amdgcn_s_memtime();
#endif
            for (;;) {
                // window bounds for this iteration's extension of the LDS window (done in the roles phase below)
                const int pf_limit = min(min(sh.base + 4 * wmax, sh.avail), sh.base + rg.mask + 1 - 8);   // two iterations + their swaps ahead (role P)
                int p = 0;
                bool have_p = false;
                // the swap of the iteration before: wave 0 waits for the ranks' records and decides it, then takes its own
                // step; the other waves are in their steps' fronts meanwhile and ask for the outcome at their decisions
                if constexpr (lockrun) { if (xpend && first && wave == 0) exchange_finish(cs, sh, s_gath, iter - 1, lane); }
                for (int c = wave; c < nc; c += NW) {
                    if (c < redo) continue;
                    if (__builtin_expect(!first, 0)) {
                        if (lane == 0) {
                            undo_chain(cs, sh, rg, c);
                            sh.redone[c] = 1;
                            // workers that get to an order of this chain late may have read the step just taken back: whatever
                            // role P has out for this chain is void -- and the workers are told so
                            if constexpr (PERSIST) { if (sh.pre_p[0][c] != -1 || sh.pre_p[1][c] != -1) void_slot(cs, c); }
                            sh.pre_p[0][c] = -1; sh.pre_p[1][c] = -1;
                        }
                        p = sh.start_fix[c];                                // corrected by the validation
                    }                                                       // else: optimistic start, c steps after base (below)
                    int base_used = sh.base;
                    if (first && !have_p) p = c == 0 ? base_used : base_used + rg.hop[(base_used & rg.mask) * kHops + c - 1];
                    // a later round of this wave: the step starts where the previous chain ended (role P is through: the
                    // wave's first step waited for it)
                    if constexpr (PERSIST) { if (first && have_p && have_prev && rolep_on) drop_disproved_orders(cs, sh, c, p, iter, lane); }
                    [[maybe_unused]] int xw = (lockrun && xpend && first && !have_p) ? iter - 1 : -1;       // the wave's first step of the iteration waits for the swap
                    if constexpr (lockrun) {
                        int r;
                        for (;;) {
                            r = chain_pass<NCH, PERSIST, F32, MK>(f, cs, sh, rg, s_sx, s_sy, s_sz, c, p, iter, lane, launch,
                                                              have_prev && rolep_on, first, xw, base_used);
                            if (r != kPassRestart) break;
                            // the swap took its judge draw from THIS rank's stream: every step of the iteration starts elsewhere
                            base_used = sh.base;
                            p = hop_ahead(rg, base_used, c);
                            xw = -1;
                            if (have_prev && rolep_on) drop_disproved_orders(cs, sh, c, p, iter, lane);   // (the pass left after its wait for role P)
                        }
                        if (r == kPassAbort) break;                             // the job stops after the iteration before
                        p = r;
                    } else {
                        p = chain_pass<NCH, PERSIST, F32, MK>(f, cs, sh, rg, s_sx, s_sy, s_sz, c, p, iter, lane, launch,
                                                          have_prev && rolep_on, first, -1, 0);
                    }
                    have_p = true;
                    if (NW > 1 && c + NW < nc) p += rg.hop[(p & rg.mask) * kHops + NW - 2];   // skip NW-1 steps
                }
                // what wave_P staged during the previous roles phase becomes part of the window now: sh.fill is
                // stable between two barriers A, so every role sees the same window
                if (wave == wave_P && lane == 0 && fill_next > sh.fill) sh.fill = fill_next;
#ifdef HTM_STAMPS
                if (lane == 0 && cs.stamps && first) {
                    const unsigned long long ta = __builtin_amdgcn_s_memtime();
                    const bool job = nc > wave && sh.prop[wave].need_full != 0;
                    atomicAdd(&sh.stamp_acc[48 + wave + (job ? 8 : 0)], ta - t_top);
                    atomicAdd(&sh.stamp_acc[64 + wave + (job ? 8 : 0)], 1ull);
                }
#endif
                __syncthreads();                                            // ---- barrier A
                STAMP(2);   // passes
#ifdef HTM_STAMPS
                const unsigned long long t_roles0 = __builtin_amdgcn_s_memtime();
#define RSTAMP(k) do { if (lane == 0 && cs.stamps) atomicAdd(&cs.stamps[120 + (k)], __builtin_amdgcn_s_memtime() - t_roles0); } while (0)
#else
#define RSTAMP(k) do { } while (0)
#endif
                if (lockrun && xpend) {
                    // every wave has heard about the swap of the iteration before; if that iteration was the last one
                    // (a rank asked for a stop, or failed) no step of this one was taken: leave
                    xpend = false;
                    if (sh.c.stop || sh.c.err) { leave = true; __syncthreads(); break; }
                }
                if (wave == wave_P && first) {
                    // one wave without a role keeps the LDS window of the stream ahead while the roles run: 64 more
                    // positions per iteration; the memory latency hides behind the roles phase
                    PfRegs pf;
                    pf_load(pf, cs, sh, sh.fill + lane, pf_limit);
                    pf_store(pf, rg);
                    fill_next = max(sh.fill, min(sh.fill + 64, pf_limit));
                    RSTAMP(3);   // window extension
                }
                if (wave == 0 || wave == wave_R || wave == wave_W) {
                    CsRef cs = rebase(cs_);
                    const Valid v = validate<PERSIST>(sh, nc, lane);
                    int pos = 0, i1, i2;
                    double sr, slr;
                    const bool ok_plan = v.bad ? true : swap_plan(cs, sh, rg, lockstep, v.base + v.total, pos, i1, i2, sr, slr);
                    const bool done = !v.bad && ok_plan && v.mf == 0;       // iteration complete in this launch
                    if (wave == 0) {                                        // role V: validation + bookkeeping
                        const bool in = lane < nc;
                        if (v.bad) {
                            const int my_cnt = in ? sh.cnt[lane] : 0;
                            const int exact = v.base + wave_incl_scan(my_cnt) - my_cnt;
                            if (in) sh.start_fix[lane] = exact;
                            if (lane == 0) sh.redo = __ffsll((long long)v.bad) - 1;
                        } else if (!ok_plan) {
                            // nothing has been decided for good: undo and retry in the next launch
                            if (lane == 0) { sh.redo = -2; if (lockstep) sh.c.err = -7; else sh.c.stop = 2; }
                        } else {
                            const unsigned long long mp = __ballot(in && sh.prop[lane].prior_ok != 0 && sh.prop[lane].need_full == 0);
                            const unsigned long long mfull = __ballot(in && sh.prop[lane].need_full != 0);
                            if (!PERSIST && in && ((v.mf >> lane) & 1ull)) {   // k_full's work order, chain order
                                const Proposal &pr = sh.prop[lane];
                                const int vz = opaque_zero();
                                const int slot = __popcll(v.mf & ((1ull << lane) - 1ull));
                                FullEntry *en = &cs.desc->e[slot];
                                en->chain = lane; en->type = pr.type; en->idx = pr.idx; en->pad = 0;
                                en->x_new = pr.x_new;
                                en->beta = pr.type == 1 ? pr.x_new : ld_state(cs.xall + lane, vz);
                                en->q = pr.type == 3 ? pr.x_new : ld_state(cs.xall + (cs.n_chains + cs.n_chains * cs.S) + lane, vz);
                            }
                            if (lane == 0) {
                                sh.c.swap_i1 = i1; sh.c.swap_i2 = i2; sh.c.swap_r = sr; sh.c.swap_logr = slr;
                                sh.c.spos = sh.origin + pos;     // RNG commit: draws consumed so far
                                sh.base = pos;
                                if constexpr (lockrun) {
                                    // rank 0 drew the pair itself: it knows already whether the judge draw will be its own
                                    if (cs.rank == 0 && i1 >= 0 && i1 / nc == 0) sh.base = pos + 1;
                                    // (role R is adding this iteration's records meanwhile: at most nc of each)
                                    const bool my_stop = sh.c.n_lik + 3 * nc > cs.cap_lik || sh.c.n_smp + 3 * nc > cs.cap_smp ||
                                                         sh.avail < pos + 4 * wmax;
                                    sh.xstop = my_stop ? 1 : 0;
                                    sh.xspec = (!my_stop && iter < sh.c.iter_target && sh.c.err == 0 && v.mf == 0) ? 1 : 0;
                                }
                                sh.redo = -1;
                                sh.catchup = (sh.fill < pos + wmax) ? 1 : 0;
                                sh.c.n_full = __popcll(v.mf);
                                if (!PERSIST && v.mf) cs.desc->n = __popcll(v.mf);
                                sh.c.n_full_evals += __popcll(mfull);
                                sh.c.n_partial_evals += __popcll(mp);
                                if (v.mf == 0) {
                                    if (lockstep) sh.c.stage = ST_WAIT_SWAP;
                                    else { sh.c.iter_done = iter; sh.c.stage = ST_IDLE; }
                                }
                            }
                        }
                    }
                    if (wave == 0) RSTAMP(0);   // role V
                    if (wave == wave_R && done) { role_records(cs, sh, iter, lockstep, lane, rec); RSTAMP(1); }
                    if (wave == wave_W && done) { role_swap(cs, sh, iter, lockstep, lane, i1, i2, sr, slr, rec); RSTAMP(2); }
                }
                __syncthreads();                                            // ---- barrier B
                STAMP(3);   // validation + plan + records + swap
                if (sh.redo < 0) break;
                redo = sh.redo;
                first = false;
            }
            if (leave) break;
            if (sh.redo == -2) {                 // aborted: take everything back, retry in the next launch
                for (int c = wave; c < nc; c += NW)
                    if (lane == 0) undo_chain(cs, sh, rg, c);
                break;
            }
            if (sh.c.n_full > 0) {
                if constexpr (PERSIST) {
                    if (tid == 0) sh.c.err = -9;     // cannot happen: a chain wave evaluates its own order
                    __syncthreads();
                    break;
                } else {                         // hand over to k_full; the next launch resumes at P0
                    for (int c = tid; c < nc; c += blockDim.x) cs.prop[c] = sh.prop[c];
                    if (tid == 0) sh.c.stage = ST_WAIT_FULL;
                    break;
                }
            }
        } else {
            // the iteration that waited for k_full: only its end is left (decisions were taken in P0)
            if (wave == wave_R) role_records(cs, sh, iter, lockstep, lane, rec);
            if (wave == wave_W) role_swap(cs, sh, iter, lockstep, lane, sh.c.swap_i1, sh.c.swap_i2, sh.c.swap_r, sh.c.swap_logr, rec);
            if (tid == 0) {
                if (lockstep) sh.c.stage = ST_WAIT_SWAP;
                else { sh.c.iter_done = iter; sh.c.stage = ST_IDLE; }
            }
            __syncthreads();
            resume = false;
        }
        // ---------------- after the barrier: swap + records of the chains this wave owns ---------------
        if constexpr (lockrun) {
            // this rank's record goes out first (it is what the other ranks wait for).  Then either the next iteration
            // starts at once and learns the swap at its first decision (xspec, the usual case), or -- last iteration of the
            // launch -- wave 0 waits for the records here and everybody meets at barrier C.
            const bool spec = sh.xspec != 0;
            if (wave == 0) {
                exchange_post(cs, sh, iter, lane, sh.xstop != 0);
                if (!spec) exchange_finish(cs, sh, s_gath, iter, lane);
            }
            for (int c = wave; c < nc; c += NW) post_chain(cs, sh, c, iter, lane);
            if (!spec) __syncthreads();                           // ---- barrier C
            xpend = spec;
        } else {
            for (int c = wave; c < nc; c += NW) post_chain(cs, sh, c, iter, lane);
        }
        it_local = iter;
        have_prev = true;
        STAMP(4);   // post
        if (lockstep && !lockrun) break;
    }
    __syncthreads();
    for (int k = tid; k < 7 * nc; k += blockDim.x) {          // flush this launch's counters
        if (sh.np[k]) atomicAdd(&cs.n_propose[k], sh.np[k]);
        if (sh.na[k]) atomicAdd(&cs.n_accept[k], sh.na[k]);
    }
    if (tid == 0) *cs.ctrl = sh.c;
    STAMP(5);
#ifdef HTM_STAMPS
    __syncthreads();
    if (cs.stamps)
        for (int k = tid; k < 96; k += blockDim.x)
            if (k < 20 || k >= 32) { if (sh.stamp_acc[k]) atomicAdd(&cs.stamps[k], sh.stamp_acc[k]); }
#endif
}

template <int NCH, bool F32 = false>
__global__ __launch_bounds__(512) void k_step(FwdDev f, ChainsDev cs, int mode, int target_arg,
                                               const double *gathered, int ring_size, int wmax)
{
    const KArgLayout __attribute__((address_space(4))) &ka = *(const KArgLayout __attribute__((address_space(4))) *)__builtin_amdgcn_kernarg_segment_ptr();
    step_body<NCH, false, F32, -1>(ka.f, ka.cs, mode, target_arg, gathered, ring_size, wmax, 0ull);
}

// Worker block of a k_mcmc launch: waits for work orders of its own launch and evaluates its event tile of
// every model in the order (same arithmetic as k_full: wave <-> event, lane <-> station).  Block w, wave v
// take events (w*8 + v) + k * 8 * W.  The immutable inputs of its first event (observation rows, station
// coordinates) are loaded once per launch and stay in registers.  Hand-off in both directions by tagged
// granules (data is the flag); chain state is read with agent-scope loads after the job word has been seen
// (the master drained its stores before publishing it).  Leaves when the master has finished (PSync::quit)
// or after a bounded wait.
// NWV = waves per worker block (8: the master's block shape).  PIPE: the launch's master is the pipelined one (htm_pipe.hpp), whose
// orders name a second left-out event and want the left-out events' sums reported on their own -- compiled only into those
// launches: in the workers' event loop it costs the others 13 % at 10 000 x 128 x 16 fp64 (profiles/r04_h_worker_pipe_ab.txt).
template <int NCH, bool F32, int NWV = 8, bool PIPE = false>
__device__ __forceinline__ void worker_body(FwRef f_, CsRef cs_, unsigned long long launch, int w)
{
    CsRef cs = rebase(cs_);
    FwRef f = rebase(f_);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *s_red = reinterpret_cast<double *>(smem);          // [NWV <= 16]
    unsigned *s_tag = reinterpret_cast<unsigned *>(smem + 128);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int W = cs.n_workers;
    const int nc = cs.n_chains;
    const int ev0 = w * NWV + wave;

    constexpr int N = NCH > 0 ? NCH : 1;
    StaRegs<N> st;
    ObsRegs<N, F32> ob0;
    if constexpr (NCH > 0) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int j = lane + 64 * c;
            const bool valid = j < f.S;
            st.sx[c] = valid ? f.sx[j] : 0.0; st.sy[c] = valid ? f.sy[j] : 0.0; st.sz[c] = valid ? f.sz[j] : 0.0;
            st.tc[c] = 0.0; st.ac[c] = 0.0;
        }
        if (ev0 < f.E) load_obs_regs<NCH, F32>(ob0, f, ev0, lane);
    }

    int *s_chain = reinterpret_cast<int *>(smem + 136);
    // wave 0: lane 8k + g holds granule g of chain (8 j + k)'s slot; lane 8k remembers the last tag served
    constexpr int kGroups = (kMaxChains + 7) / 8;
    unsigned last_tag[kGroups];
#pragma unroll
    for (int j = 0; j < kGroups; ++j) last_tag[j] = 0;
    // An order whose named commit does not show within kDeferTicks is put aside -- its tag stays in last_tag, so the polls
    // skip it -- and becomes eligible again kDeferTicks later (if the slot still holds it): a wait that the master's
    // signals should have ended (void_slot) must never keep the worker from the other chains' orders.
    constexpr unsigned long long kDeferTicks = 2000ull;       // 20 us of the 100 MHz clock; legitimate waits are a few us
    // (kept in LDS, wave 0 only: scalar registers are what this kernel is short of)
    int *s_defer_chain = reinterpret_cast<int *>(smem + 192);
    unsigned long long *s_defer_until = reinterpret_cast<unsigned long long *>(smem + 200);
    if (threadIdx.x == 0) { *s_defer_chain = -1; *s_defer_until = 0ull; }
    // [0] type | idx << 3, [1] x_new high, [2] x_new low, [3] committed element or ~0, [4] its value high, [5] low,
    // [6] element (+1) whose value, as seen by this evaluation, is to be reported (two-ahead orders), 0 = none
    unsigned *s_job = reinterpret_cast<unsigned *>(smem + 160);
    const unsigned long long *slots = cs.slots + (size_t)(w % cs.slot_rep) * cs.slot_stride;
    const int npoll = cs.npoll;
    for (;;) {
        if (wave == 0) {
            unsigned tag = 0;
            int chain = -1;
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();          // 100 MHz
            // one poll = the slots of all chains (one load per 8 chains) + the quit word; npoll polls in flight
            constexpr int kPolls = 3;
            unsigned long long x[kPolls][kGroups], qw[kPolls];
            auto issue = [&](int b) __attribute__((always_inline)) {
#pragma unroll
                for (int j = 0; j < kGroups; ++j) {
                    x[b][j] = 0ull;
                    if (8 * j < nc && 8 * j + (lane >> 3) < nc) x[b][j] = ld_agent(slots + (size_t)(8 * j) * kGranPerSlot + lane);
                }
                qw[b] = lane == 63 ? ld_agent(&cs.ps->quit) : 0ull;
            };
            // returns 1: an order was taken (tag, chain, s_job set), -1: the master has quit, 0: nothing yet
            auto check = [&](int b) __attribute__((always_inline)) -> int {
#pragma unroll
                for (int j = 0; j < kGroups; ++j) {
                    if (8 * j >= nc) continue;
                    const unsigned t = (unsigned)(x[b][j] >> 32), pay = (unsigned)x[b][j];
                    const unsigned tq = (unsigned)__builtin_amdgcn_update_dpp(0, (int)t, 0x00, 0xF, 0xF, false);    // quad_perm [0,0,0,0]
                    const unsigned t4 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)t, 0x114, 0xF, 0xF, false);   // row_shr:4
                    bool ok = t == tq && t != 0u && 8 * j + (lane >> 3) < nc;
                    if ((lane & 7) == 4) ok = ok && t == t4;                 // second quad agrees with the first
                    if ((lane & 7) == 0) ok = ok && pay == (unsigned)launch && t != last_tag[j];
                    unsigned long long m = __ballot(ok);
                    m &= m >> 1; m &= m >> 2; m &= m >> 4; m &= 0x0101010101010101ull;
                    if (m) {
                        const int l0 = __ffsll((long long)m) - 1;           // lane of granule 0 of the chosen chain
                        chain = 8 * j + (l0 >> 3);
                        tag = (unsigned)__builtin_amdgcn_readlane((int)t, l0);
                        if (lane == l0) last_tag[j] = tag;
                        unsigned pv[7];
#pragma unroll
                        for (int q = 0; q < 7; ++q) pv[q] = (unsigned)__builtin_amdgcn_readlane((int)pay, l0 + 1 + q);
                        if (lane == 0) {
#pragma unroll
                            for (int q = 0; q < 7; ++q) s_job[q] = pv[q];
                        }
                        return 1;
                    }
                }
                const unsigned qlo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)qw[b], 63);
                const unsigned qhi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(qw[b] >> 32), 63);
                return ((((unsigned long long)qhi << 32) | qlo) > launch) ? -1 : 0;
            };
#pragma unroll
            for (int b = 0; b < kPolls; ++b) if (b < npoll) issue(b);
            auto rearm = [&](int dc) __attribute__((always_inline)) {       // the order put aside may be taken again
#pragma unroll
                for (int j = 0; j < kGroups; ++j)
                    if (j == (dc >> 3) && lane == 8 * (dc & 7)) last_tag[j] = 0;
                if (lane == 0) *s_defer_chain = -1;
            };
            for (bool spin = true; spin;) {
                {
                    const int dc = *s_defer_chain;
                    if (__builtin_expect(dc >= 0, 0)) { if (__builtin_amdgcn_s_memrealtime() > *s_defer_until) rearm(dc); }
                }
#pragma unroll
                for (int b = 0; b < kPolls; ++b) {
                    if (b >= npoll) continue;
#ifdef HTM_STAMPS
                    if (cs.stamps && w == 0 && lane == 0) cs.stamps[27] += 1;      // polls of worker 0
#endif
                    int r = spin ? check(b) : 0;
                    if (r == 1 && s_job[3] != 0xffffffffu) {
                        // The order was sent ahead of the chain's latest commit: wait until that store is what memory returns
                        // (the other waves read the chain's state only after this wave has seen it).  If the master takes the
                        // order back meanwhile -- the chain repeated a pass and rewrote the element, so the value never shows --
                        // it says so by rewriting the slot (void_slot, or a newer order): the order is dropped and the worker
                        // keeps polling.  A wait that lasts longer than any commit takes to land puts the order ASIDE for a while
                        // (kDeferTicks): it is neither dropped nor answered on the clock's say-so, the worker merely serves the other
                        // chains' orders meanwhile and looks again later.  The further bounds are fail-stops (master gone, 30 s).
                        const unsigned long long want = ((unsigned long long)s_job[4] << 32) | s_job[5];
                        const unsigned long long tw0 = __builtin_amdgcn_s_memrealtime();
                        for (unsigned spins = 0;; ++spins) {
                            if ((unsigned long long)__double_as_longlong(ld_agent(cs.xall + s_job[3])) == want) break;
                            const unsigned now_tag = (unsigned)(ld_agent(slots + (size_t)chain * kGranPerSlot) >> 32);
                            if (now_tag != tag) {
#ifdef HTM_STAMPS
                                if (lane == 0 && w == 0 && cs.stamps) {
                                    cs.stamps[112] += 1; cs.stamps[113] = chain; cs.stamps[114] = tag; cs.stamps[115] = s_job[3]; cs.stamps[116] = want;
                                    cs.stamps[117] = (unsigned long long)__double_as_longlong(ld_agent(cs.xall + s_job[3])); cs.stamps[118] = s_job[0]; cs.stamps[119] = now_tag;
                                }
#endif
                                r = 0; tag = 0; chain = -1;      // taken back: keep polling
                                break;
                            }
                            if ((spins & 7u) == 7u && __builtin_amdgcn_s_memrealtime() - tw0 > kDeferTicks) {
                                // not now: back to the polls, this order aside (it stays in last_tag until it is re-armed)
                                if (w == 0 && lane == 0) cs.diag[24] += 1;          // (reported with error -8)
                                { const int dc = *s_defer_chain; if (dc >= 0) rearm(dc); }
                                if (lane == 0) { *s_defer_chain = chain; *s_defer_until = __builtin_amdgcn_s_memrealtime() + kDeferTicks; }
                                r = 0; tag = 0; chain = -1;
                                break;
                            }
                            if ((spins & 63u) == 63u) {          // fail-stops only
                                if (w == 0 && lane == 0 && cs.diag[16] == 0 && __builtin_amdgcn_s_memrealtime() - t0 > 100000000ull) {
                                    // a second in this wait: leave a note for the host's error message (once)
                                    unsigned long long *dg = cs.diag;
                                    dg[17] = chain; dg[18] = tag; dg[19] = s_job[3]; dg[20] = want;
                                    dg[21] = (unsigned long long)__double_as_longlong(ld_agent(cs.xall + s_job[3])); dg[22] = s_job[0]; dg[23] = s_job[6];
                                    dg[16] = 1;
                                }
                                if (ld_agent(&cs.ps->quit) > launch) { r = -1; tag = 0; chain = -1; break; }
                                if (__builtin_amdgcn_s_memrealtime() - t0 > 3000000000ull) { r = -1; tag = 0; chain = -1; break; }
                            }
                            __builtin_amdgcn_s_sleep(1);
                        }
                    }
                    if (r != 0) spin = false;
                    if (spin) issue(b);
                }
                if (spin && __builtin_amdgcn_s_memrealtime() - t0 > 3000000000ull) spin = false;   // never spin forever (30 s)
            }
            if (lane == 0) { *s_tag = tag; *s_chain = chain; }
        }
        __syncthreads();
        const unsigned tag = *s_tag;
        const int m = *s_chain;
        if (tag == 0) return;
#ifdef HTM_STAMPS
        const bool wstamp = cs.stamps && w == 0 && tid == 0;
        if (wstamp) cs.stamps[21] += __builtin_amdgcn_s_memrealtime();
#endif
        {
            // (bit 31 of the order word: an order of the pipelined master, htm_pipe.hpp -- no commit is named, and the word after
            // it names a SECOND event the workers leave out)
            const bool pipe_order = PIPE && (s_job[0] >> 31) != 0u;
            const int type = (int)(s_job[0] & 7u), idx = (int)((s_job[0] & 0x7fffffffu) >> 3);
            const double ov_val = __longlong_as_double((long long)(((unsigned long long)s_job[1] << 32) | s_job[2]));
            // vs and qs of the evaluated model: chain state unless they are the proposal (same round of loads as
            // the corrections and the event coordinates below)
            const GroupOff go = group_offsets(cs);
            const double beta_c = ld_agent(cs.xall + m), q_c = ld_agent(cs.xall + go.qs + m);
            const double beta = type == 1 ? ov_val : beta_c, q = type == 3 ? ov_val : q_c;
            // (once per order, not once per event: cls_forward.f90:118, :204 divide per station; event_misfit multiplies)
            const double rbeta_j = 1.0 / beta, katt_j = (kPi * kFreq) / (q * beta);
            int ov_kind = 0, ov_idx = -1, ov_evt = -1, ov_cmp = 0;
            const int r_off = (int)s_job[6] - 1 - (go.hy + m * go.nh);
            const int r_evt = s_job[6] ? r_off / 3 : -1;       // two-ahead orders: the event left to the chain's own wave
            const int r_off2 = (int)s_job[4] - 1 - (go.hy + m * go.nh);
            const int r_evt2 = (pipe_order && s_job[4]) ? r_off2 / 3 : -1;
            // An order of the pipelined master: the wave that holds a left-out event evaluates it at BOTH candidate positions --
            // where the state has it, and with the component that the hypocentre step in between proposes (its value comes in
            // tagged granules beside the order) -- under this order's parameters, and reports the two sums on their own: the
            // master adds the one that step's decision selects (htm_pipe.hpp).
            unsigned long long lgv0 = 0, lgv1 = 0, lgv2 = 0, lgv3 = 0;
            if (pipe_order && (r_evt >= 0 || r_evt2 >= 0)) {
                const unsigned long long *lgi = cs.lo_gran + (size_t)m * 16;
                lgv0 = ld_agent(lgi); lgv1 = ld_agent(lgi + 1); lgv2 = ld_agent(lgi + 2); lgv3 = ld_agent(lgi + 3);
            }
            auto leftout = [&](const auto &ob_, double x, double y, double z, int e, const auto &st_, double rb_, double ka_) __attribute__((always_inline)) {
                if constexpr (NCH > 0) {
                    const int k = (e == r_evt) ? 0 : 1;
                    const int cmpk = k == 0 ? r_off - 3 * r_evt : r_off2 - 3 * r_evt2;
                    unsigned long long *lg = cs.lo_gran + (size_t)m * 16;
                    unsigned long long a = k == 0 ? lgv0 : lgv2, b = k == 0 ? lgv1 : lgv3;      // (requested with the order's state loads)
                    for (unsigned spins = 0; spins < (1u << 22); ++spins) {
                        if ((unsigned)(a >> 32) == tag && (unsigned)(b >> 32) == tag) break;
                        __builtin_amdgcn_s_sleep(1);
                        a = ld_agent(lg + 2 * k); b = ld_agent(lg + 2 * k + 1);
                    }
                    if ((unsigned)(a >> 32) != tag || (unsigned)(b >> 32) != tag) return;      // (never seen: the master's collector reports the unanswered order)
                    const double xn = gran_f64(a, b);
                    const double px[2] = {x, cmpk == 0 ? xn : x};
                    const double py[2] = {y, cmpk == 1 ? xn : y};
                    const double pz[2] = {z, cmpk == 2 ? xn : z};
                    double o2[2];
                    event_misfit<(NCH > 0 ? NCH : 1), 2, F32, true>(f, ob_, lane, st_, px, py, pz, rb_, ka_, o2);
                    wave_sum<2>(o2);
                    if (lane == 0) { st_gran_f64(lg + 8 + 4 * k, tag, o2[0]); st_gran_f64(lg + 8 + 4 * k + 2, tag, o2[1]); }
                }
            };
            if (type == 2 || type == 4) { ov_kind = type; ov_idx = idx; }
            else if (type >= 5) { ov_evt = idx / 3; ov_cmp = idx - 3 * ov_evt; }
            const double *hyp = cs.xall + go.hy + (size_t)m * go.nh;
            const double *tc = cs.xall + go.tc + (size_t)m * cs.S, *ac = cs.xall + go.ac + (size_t)m * cs.S;
            double lane_acc = 0.0;
            if constexpr (NCH > 0) {
#pragma unroll
                for (int c = 0; c < NCH; ++c) {                 // chain state: agent-scope loads
                    const int j = lane + 64 * c;
                    const bool valid = j < f.S;
                    st.tc[c] = valid ? ld_agent(tc + j) : 0.0;
                    st.ac[c] = valid ? ld_agent(ac + j) : 0.0;
                    if (valid && ov_idx == j) { if (ov_kind == 2) st.tc[c] = ov_val; if (ov_kind == 4) st.ac[c] = ov_val; }
                }
                if constexpr (!F32) {
                    // Events ev0, ev0 + 8W, ...: the first one's observation rows are resident (ob0); every further event's
                    // rows and coordinates are requested BEFORE the current event is evaluated (software pipeline, one
                    // event of look-ahead).  fp64: an event's arithmetic covers the next one's loads, the deeper form below
                    // measured -1 % at 1 000 and 10 000 events x 64 stations, and two stations per lane do not fit the 256
                    // registers of a wave with three buffers.
                    const bool full_rows64 = f.S == 64 * NCH && f.use_time != 0 && f.use_amp != 0;      // (loads without selects or branches)
                    ObsRegs<NCH, F32> ob_cur = ob0, ob_nxt = ob0;
                    double cx = 0.0, cy = 0.0, cz = 0.0, nx = 0.0, ny = 0.0, nz = 0.0;
                    if (ev0 < f.E) { cx = ld_agent(hyp + 3 * ev0); cy = ld_agent(hyp + 3 * ev0 + 1); cz = ld_agent(hyp + 3 * ev0 + 2); }
                    for (int ev = ev0; ev < f.E; ev += NWV * W) {
                        const int evn = ev + NWV * W;
                        if (evn < f.E) {
                            if (full_rows64) load_obs_regs_nobranch<NCH, F32, HTM_NT_WORKERS != 0, HTM_VRPS_WORKERS != 0, true>(ob_nxt, f, evn, lane);
                            else load_obs_regs<NCH, F32, HTM_NT_WORKERS != 0, HTM_VRPS_WORKERS != 0>(ob_nxt, f, evn, lane);
                            nx = ld_agent(hyp + 3 * evn); ny = ld_agent(hyp + 3 * evn + 1); nz = ld_agent(hyp + 3 * evn + 2);
                        }
                        double ex = cx, ey = cy, ez = cz;      // (the proposed component of a first-iteration order: a rare branch)
                        if (__builtin_expect(ev == ov_evt, 0)) { ex = ov_cmp == 0 ? ov_val : cx; ey = ov_cmp == 1 ? ov_val : cy; ez = ov_cmp == 2 ? ov_val : cz; }
                        const double px[1] = {ex};
                        const double py[1] = {ey};
                        const double pz[1] = {ez};
                        double out[1];
                        if (__builtin_expect(pipe_order && (ev == r_evt || ev == r_evt2), 0)) leftout(ob_cur, cx, cy, cz, ev, st, rbeta_j, katt_j);
                        else {
                            event_misfit<NCH, 1, F32, true>(f, ob_cur, lane, st, px, py, pz, rbeta_j, katt_j, out);
                            // two-ahead order: the event of the step in between is left to the chain's own wave (it may be mid-commit)
                            lane_acc += (ev == r_evt) ? 0.0 : out[0];
                        }
                        ob_cur = ob_nxt; cx = nx; cy = ny; cz = nz;
                    }
                } else {
                    // Events ev0, ev0 + 8W, ...: the first one's observation rows are resident (ob0).  Three register buffers in
                    // rotation: the rows and coordinates of the event TWO places ahead are requested before the current event is
                    // evaluated, with a fixed number of loads per request (load_obs_regs_nobranch), so that at E >> 8W two
                    // events' loads fly under the arithmetic of a third: the fp32 forward's arithmetic is too short to cover an
                    // event's loads (configs[4] shape: 837 -> 902 k steps/s).  The last <= 4 events of a wave take the plain path.
                    const int s8 = NWV * W;
                    ObsRegs<NCH, F32> b0 = ob0, b1 = ob0, b2 = ob0;
                    double x0 = 0.0, y0 = 0.0, z0 = 0.0, x1 = 0.0, y1 = 0.0, z1 = 0.0, x2 = 0.0, y2 = 0.0, z2 = 0.0;
                    // (every lane has a station in both chunks and both data types are used: the loads need no selects)
                    const bool full_rows = (f.S & 63) == 0 && f.S == 64 * NCH && f.use_time != 0 && f.use_amp != 0;
                    auto fetch = [&](ObsRegs<NCH, F32> &b, double &x, double &y, double &z, int e) __attribute__((always_inline)) {
                        if (full_rows) load_obs_regs_nobranch<NCH, F32, HTM_NT_WORKERS != 0, HTM_VRPS_WORKERS != 0, true>(b, f, e, lane);
                        else load_obs_regs_nobranch<NCH, F32, HTM_NT_WORKERS != 0, HTM_VRPS_WORKERS != 0>(b, f, e, lane);
                        x = ld_agent(hyp + 3 * e); y = ld_agent(hyp + 3 * e + 1); z = ld_agent(hyp + 3 * e + 2);
                    };
                    auto eval = [&](const ObsRegs<NCH, F32> &b, double x, double y, double z, int e) __attribute__((always_inline)) {
                        // (the proposed hypocentre component of an order of the first iteration: a rare branch, not three selects per event)
                        double ex = x, ey = y, ez = z;
                        if (__builtin_expect(e == ov_evt, 0)) { ex = ov_cmp == 0 ? ov_val : x; ey = ov_cmp == 1 ? ov_val : y; ez = ov_cmp == 2 ? ov_val : z; }
                        const double px[1] = {ex};
                        const double py[1] = {ey};
                        const double pz[1] = {ez};
                        double out[1];
                        if (__builtin_expect(pipe_order && (e == r_evt || e == r_evt2), 0)) leftout(b, x, y, z, e, st, rbeta_j, katt_j);
                        else {
                            event_misfit<NCH, 1, F32, true>(f, b, lane, st, px, py, pz, rbeta_j, katt_j, out);
                            // two-ahead order: the event of the step in between is left to the chain's own wave (it may be mid-commit)
                            lane_acc += (e == r_evt) ? 0.0 : out[0];
                        }
                    };
                    int ev = ev0;
                    if (ev < f.E) {
                        x0 = ld_agent(hyp + 3 * ev); y0 = ld_agent(hyp + 3 * ev + 1); z0 = ld_agent(hyp + 3 * ev + 2);
                        if (ev + s8 < f.E) fetch(b1, x1, y1, z1, ev + s8);
                    }
                    while (ev + 4 * s8 < f.E) {            // b0 <- event ev, b1 <- ev + s8 on entry and again after the trip
                        fetch(b2, x2, y2, z2, ev + 2 * s8); eval(b0, x0, y0, z0, ev);
                        fetch(b0, x0, y0, z0, ev + 3 * s8); eval(b1, x1, y1, z1, ev + s8);
                        fetch(b1, x1, y1, z1, ev + 4 * s8); eval(b2, x2, y2, z2, ev + 2 * s8);
                        ev += 3 * s8;
                    }
                    if (ev < f.E) { if (ev + 2 * s8 < f.E) fetch(b2, x2, y2, z2, ev + 2 * s8); eval(b0, x0, y0, z0, ev); }
                    if (ev + s8 < f.E) { if (ev + 3 * s8 < f.E) fetch(b0, x0, y0, z0, ev + 3 * s8); eval(b1, x1, y1, z1, ev + s8); }
                    if (ev + 2 * s8 < f.E) eval(b2, x2, y2, z2, ev + 2 * s8);
                    if (ev + 3 * s8 < f.E) eval(b0, x0, y0, z0, ev + 3 * s8);
                }
            } else {
                // generic station count: corrections are read through plain loads after an agent acquire
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                for (int ev = ev0; ev < f.E; ev += NWV * W) {
                    const bool ov = ev == ov_evt;
                    const double hx = ld_agent(hyp + 3 * ev), hy = ld_agent(hyp + 3 * ev + 1), hz = ld_agent(hyp + 3 * ev + 2);
                    const double px[1] = {(ov && ov_cmp == 0) ? ov_val : hx};
                    const double py[1] = {(ov && ov_cmp == 1) ? ov_val : hy};
                    const double pz[1] = {(ov && ov_cmp == 2) ? ov_val : hz};
                    double out[1];
                    event_misfit_generic<1>(f, ev, lane, f.sx, f.sy, f.sz, tc, ac, ov_kind, ov_idx, ov_val, px, py, pz,
                                            beta, q, out);
                    lane_acc += out[0];
                }
            }
            const double tot = wave_sum1(lane_acc);
#ifdef HTM_STAMPS
            if (wstamp) cs.stamps[22] += __builtin_amdgcn_s_memrealtime();
#endif
            if (lane == 0) s_red[wave] = tot;
            __syncthreads();
#ifdef HTM_STAMPS
            if (wstamp) cs.stamps[23] += __builtin_amdgcn_s_memrealtime();
#endif
            if (tid == 0) {
                double tot8 = ((s_red[0] + s_red[1]) + (s_red[2] + s_red[3])) + ((s_red[4] + s_red[5]) + (s_red[6] + s_red[7]));
                if constexpr (NWV == 12) tot8 = tot8 + ((s_red[8] + s_red[9]) + (s_red[10] + s_red[11]));
                st_gran_f64(cs.pgran + ((size_t)m * cs.n_wg + w) * cs.pgran_stride, tag, tot8);
            }
            __syncthreads();
        }
    }
}

// Probe of the peer-mapped inboxes (htm_chains_xchg_probe): one wave writes a token record into every rank's inbox and
// waits (bounded, `ticks` of the 100 MHz clock) until the tokens of all ranks have arrived in its own -- the same
// stores, loads and scopes exchange_post / exchange_finish use, so a mapping whose writes are not visible to a polling kernel is
// found at set-up, not inside a run.  Token tags have the top bit set: no iteration number ever matches them.
__global__ __launch_bounds__(64) void k_xchg_probe(ChainsDev cs, unsigned token, unsigned long long ticks, int *result)
{
    const int lane = threadIdx.x, np = cs.n_procs, G = cs.xg;
    const unsigned tag = 0x80000000u | token;
    if (lane < 2)
        for (int q = 0; q < np; ++q)
            st_sys(ld_const(cs.outbox + q) + (size_t)(0 * np + cs.rank) * G + lane, ((unsigned long long)tag << 32) | (unsigned)cs.rank);
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    bool ok = false;
    for (;;) {
        bool mine = true;
        for (int r = lane >> 1; r < np; r += 32) {
            const unsigned long long v = ld_sys(cs.inbox + (size_t)r * G + (lane & 1));
            mine = mine && (unsigned)(v >> 32) == tag && (unsigned)v == (unsigned)r;
        }
        if (__all(mine)) { ok = true; break; }
        if (__builtin_amdgcn_s_memrealtime() - t0 > ticks) break;
        __builtin_amdgcn_s_sleep(8);
    }
    if (lane == 0) *result = ok ? 1 : 0;
}

}  // namespace htm
