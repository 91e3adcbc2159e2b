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

#ifndef HTM_TURN_SLEEP
#define HTM_TURN_SLEEP 1  // s_sleep argument in the turn's wait (64 clocks each)
#endif
#ifndef HTM_FAIR
#define HTM_FAIR 1        // diagnostics: 0 = no priority hand-over between the two waves of a SIMD
#endif
#ifndef HTM_FAIR_LOCK
#define HTM_FAIR_LOCK 1   // ... on a lock-step rank too (one rank: 5.87 -> 5.26 us per iteration; ranks sharing a GPU: no difference)
#endif

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
    // 1 / vs and pi f / (qs vs) of every chain (what event_misfit multiplies by: cls_forward.f90:118, :204), renewed by the chain's
    // own wave when it commits a new vs or qs: a partial update reads them instead of dividing twice (28 instructions)
    double rbeta[kMaxChains], katt[kMaxChains];
    // a lock-step rank (k_mcmc<.., 4>): the bet that judge_swap's draw is not this rank's is settled for the swaps of iterations
    // <= xanch (the next iteration re-anchored if it was lost); xanch_claim: the iteration a wave has taken on to settle it for
    int xanch, xanch_claim;
    int xctl4[4];                          // [i & 3] {i << 2 | control word} of this rank's own header of iteration i (its own waves read it here)
    // type | event << 3 of the chain's PREVIOUS step (0: none yet; kept across launches in ChainsDev::prev_mid): if it was a
    // hypocentre step, a full evaluation leaves its event to the chain's own wave -- whenever the order went out (flow_step)
    int pv_mid[kMaxChains];
};

// a wave-uniform value that reached a vector register (read from LDS) back in a scalar one
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
// lane l of a 64-bit word, as a wave-uniform value
__device__ __forceinline__ unsigned long long rl_u64(unsigned long long v, int l)
{
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, l);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), l);
    return ((unsigned long long)hi << 32) | lo;
}

// LDS words shared between the waves: relaxed accesses in program order.  LDS operations of a wave are executed in
// issue order and the LDS is one serialisation point for the workgroup, so "release" and "acquire" are compiler
// barriers here, not waits for outstanding memory operations.
__device__ __forceinline__ int lds_ld(const int *p) { const int v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); asm volatile("" ::: "memory"); return v; }
__device__ __forceinline__ unsigned long long lds_ld(const unsigned long long *p) { const unsigned long long v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); asm volatile("" ::: "memory"); return v; }
__device__ __forceinline__ void lds_st(int *p, int v) { asm volatile("" ::: "memory"); __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_st(unsigned long long *p, unsigned long long v) { asm volatile("" ::: "memory"); __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// ---- SEVERAL MASTER WORKGROUPS in one launch (k_mcmc<.., 7>; block b runs chains 8 b .. 8 b + 7, a wave each) -----------------
// One CU issues the instructions of eight chain waves and no more: more chains per rank than eight used to take
// turns on the same waves.  Here the words the chains of a rank share -- what FlowShared holds in LDS for one workgroup -- live in
// memory, written and read with agent-scope accesses, every one of them self-validating (a single 8-byte word, or tagged granules
// {tag : 32, payload : 32} whose tag says which iteration / epoch they belong to), so that no ordering between stores to
// different addresses is relied upon:
//   prog[c]       as FlowShared::prog: {epoch << 1 | prior rejected, key} of chain c's latest checked step
//   crec[i & 3][c]   chain c after iteration i: {i, T hi | T lo | L hi | L lo} -- also what FlowShared::done says
//   swrec[i & 3]     the swap the stream holds after iteration i: {i, Eof | i1 | i2 | draws | r hi | r lo | log r hi | lo}
//   anch[e & 1]      the anchor of epoch e: {e, key}, {e, pos}
//   word[]           0 epoch, 1 last iteration of the launch, 2 failure word, 3 stop code, 4 / 5 likelihood / sample records
//                    written, 6 master blocks that have finished, 7 -
// A step looks at all of it ONCE, with one load instruction per lane issued when its own check is published -- the round trip
// (~1 us) runs under the evaluation -- and again only while its turn has not come.  Same job, same stream order, same results
// as one workgroup: the protocol is FlowShared's (tools/flow_protocol_sim.py), only the medium differs.
struct MbShared {
    unsigned long long prog[kMaxChains];
    unsigned long long crec[4][kMaxChains][4];
    unsigned long long swrec[4][8];
    unsigned long long anch[2][2];
    unsigned long long word[8];
    unsigned long long n_full, n_part;
};
enum { MW_EPOCH = 0, MW_LAST = 1, MW_ERR = 2, MW_STOP = 3, MW_NLIK = 4, MW_NSMP = 5, MW_FIN = 6 };

struct MbWave {                   // what a chain wave of a multi-block launch saw with its latest look at MbShared
    int epoch, last_iter, err, n_lik, n_smp;
    unsigned long long pv;        // lane < n_chains: prog[lane]
#ifdef HTM_MB_DIAG
    unsigned hist;                // diagnostics (tools/mb_steplog.py): a nibble per event of the current step -- 0kee entered in epoch ee (k: start known from the look-ahead), 10ee adopted ee at the top, 11ee adopted ee in the turn
#endif
};
#ifdef HTM_MB_DIAG
#define MB_HIST(mw_, nib_) do { (mw_).hist = ((mw_).hist << 4) | (unsigned)(nib_); } while (0)
#else
#define MB_HIST(mw_, nib_) do { } while (0)
#endif
// one look: lane < nc: prog[lane]; lanes 59..63: words 4, 5 (records written), 0 (epoch), 1 (last iteration), 2 (failure)
__device__ __forceinline__ unsigned long long mb_look(const MbShared *g, int nc, int lane)
{
    unsigned long long v = 0ull;
    if (lane < nc) v = ld_agent(&g->prog[lane]);
    else if (lane >= 59) { const int k = lane == 59 ? MW_NLIK : lane == 60 ? MW_NSMP : lane - 61; v = ld_agent(&g->word[k]); }
    return v;
}
__device__ __forceinline__ void mb_take(MbWave &mw, unsigned long long v)
{
    mw.pv = v;
    mw.n_lik = (int)(unsigned)rl_u64(v, 59); mw.n_smp = (int)(unsigned)rl_u64(v, 60);
    mw.epoch = (int)(unsigned)rl_u64(v, 61); mw.last_iter = (int)(unsigned)rl_u64(v, 62); mw.err = (int)(unsigned)rl_u64(v, 63);
}
// `n` granules from `base`, all tagged `tag`: polls until they are (false: gave up, sh.c.err set)
__device__ __forceinline__ bool mb_wait(FlowShared &sh, const unsigned long long *base, int n, unsigned tag, int lane, unsigned long long &v, int code = -12,
                                        unsigned mask = 0xffffffffu)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (unsigned spin = 0;; ++spin) {
        v = lane < n ? ld_agent(base + lane) : ((unsigned long long)tag << 32);
        if (__all(((unsigned)(v >> 32) & mask) == (tag & mask)) && __all((unsigned)(v >> 32) == (unsigned)(rl_u64(v, 0) >> 32) || lane >= n)) return true;
        if ((spin & 7u) == 7u) {
            if (sh.c.err != 0) return false;
            if (__builtin_amdgcn_s_memrealtime() - t0 > 500000000ull) { if (lane == 0) sh.c.err = code; return false; }
        }
        __builtin_amdgcn_s_sleep(1);
    }
}

struct FlowWave {                 // a wave's predictions (wave-uniform)
    int epoch, akey;              // the epoch they are made in, that epoch's anchor key
    int rc, rpos;                 // "chain rc's step of the current iteration starts at rpos" (rc = 0: the iteration's base)
    int rc1, rpos1;               // the same for the next iteration
    int B2;                       // base of the iteration after that (orders sent ahead only); -1 = not known
};

#ifdef HTM_STAMPS
// diagnostic cycle accounting of the free-running master (tools/flow_stamps.py): per wave, [k] ticks of phase k summed over its
// partial-update steps (0 front: loads issued, 1 proposal + check published, 2 evaluation, 3 turn, 4 swap + decision + commit,
// 5 records + orders), 6 ticks of its full-evaluation steps, 7 / 8 the two counts, 9 ticks between steps (loop top), 10 wait part of 6
#define FSTAMP(k) do { if (lane == 0 && cs.stamps) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); st_acc[k] += n_ - t_last; t_last = n_; } } while (0)
#else
#define FSTAMP(k) do { } while (0)
#endif

// What a wave with ONE chain knows about its next step before the step begins: looked up from the stream window while the
// current step is evaluated (hop table -> start position -> decoded proposal, Gaussian, judge draw: two dependent LDS round
// trips that used to open every step, and the same again for the orders sent ahead and for the base of the iteration after
// next), so that the LDS latencies run under the evaluation instead of in front of it.
struct FlowNext {
    int p, it, c, epoch;          // start position of step (it, c), predicted in `epoch`; p < 0: nothing known
    int type, idx, evt, dec_w;
    double g, r, logr;
    int b3;                       // base of iteration it + 2 (what the wave's B2 becomes when it moves on); -1: not known
};

// What a step reads from LDS before its loads go out, requested in ONE batch at the top of the loop (they used to be five
// dependent round trips -- epoch, end of launch, the other wave's progress, the chain's order book, its log-likelihood -- each
// behind a branch on the one before; LDS operations return in order, the first use waits for all of them once).
struct FlowTop {
    int epoch, last_iter, err, pk;
    int book_pos, book_mode, book_mid, pv_mid;
    unsigned book_tag;
    double L;
    double rbeta, katt;           // the chain's 1 / vs and pi f / (qs vs) (FlowShared)
};

constexpr int kFlowRestart = -1;  // flow_step: the step's position was disproved before its turn came: run it again
constexpr int kFlowAbort = -2;    // flow_step: a wait gave up (sh.c.err is set)
constexpr int kFlowStop = -3;     // flow_step (lock-step rank): the job stops after the iteration before: this step is not taken

// select_pair + the judge_swap draw starting at E (cls_parallel.f90:226-230, :163): pair, draws used in all (single rank:
// this rank draws both).  The stream service has the usual case precomputed (sw ring); more than 12 redraws follow the stream.
__device__ __forceinline__ bool flow_swap_at(CsRef cs_, const StepShared &sh, const Ring &rg, int E, int limit, int &i1, int &i2, int &nd)
{
    CsRef cs = rebase(cs_);
    const int n_all = cs.n_procs * cs.n_chains;
    i1 = -1; i2 = -1; nd = 0;
    if (n_all <= 1) return true;
    if (E + 2 >= limit) return false;
    if (rg.lock && cs.rank != 0) return true;     // lock-step: the pair is rank 0's; this rank draws only if the pair's first chain is its own (a bet: not)
    const i32x4 sw = reinterpret_cast<const i32x4 *>(rg.sw)[E & rg.mask];
    if (sw.z > 0) {
        i1 = sw.x; i2 = sw.y;
        // (single rank: this rank draws judge_swap's number too; lock-step rank 0: only if the pair's first chain is its own)
        nd = sw.z + ((!rg.lock || i1 / cs.n_chains == 0) ? 1 : 0);
        return E + nd < limit;
    }
    int pos = E;
    i1 = (int)(rg.U[pos & rg.mask] * cs.n_procs * cs.n_chains);
    pos++;
    for (;;) {
        if (pos + 1 >= limit) return false;
        i2 = (int)(rg.U[pos & rg.mask] * cs.n_procs * cs.n_chains);
        pos++;
        if (i1 != i2) break;
    }
    nd = pos - E + ((!rg.lock || i1 / cs.n_chains == 0) ? 1 : 0);
    return true;
}

// base of the next iteration when n chain steps of this one remain from pos; -1 if the window does not cover it
// (BOUNDED: chained table entries -- n > kHops -- are read only inside the window.  Two master workgroups keep a shorter lead
// than the three iterations of one workgroup, whose look-ups have never come near the window's end.)
template <bool BOUNDED = false>
__device__ __forceinline__ int flow_next_base(CsRef cs, const StepShared &sh, const Ring &rg, int pos, int n, int limit)
{
    if (pos < 0) return -1;
    // (chained table entries when n > kHops: each one read must lie inside the window -- limit -- like the result)
    int E = pos;
    if (!BOUNDED || n <= kHops) E = hop_ahead(rg, pos, n);      // (one entry, at a position inside the window)
    else {
        int m = n;
        while (m > kHops) { if (E + 16 >= limit) return -1; E += rg.hop[(E & rg.mask) * kHops + kHops - 1]; m -= kHops; }
        if (m > 0) { if (E + 16 >= limit) return -1; E += rg.hop[(E & rg.mask) * kHops + m - 1]; }
    }
    int i1, i2, nd;
    if (E + 16 >= limit || !flow_swap_at(cs, sh, rg, E, limit, i1, i2, nd)) return -1;
    return E + nd;
}

// the anchor as (iteration, chain, position of that chain's step).  A rejected LAST step of an iteration leaves pos = the
// end of that iteration's chain steps: chain 0 of the next iteration starts after the swap's draws.
__device__ __forceinline__ void flow_from_anchor(CsRef cs_, const FlowShared &sh, const Ring &rg, unsigned long long a, int &ia, int &ca, int &ap)
{
    CsRef cs = rebase(cs_);
    const int nc = cs.n_chains;
    const int key = (int)(unsigned)(a >> 32), pos = (int)(unsigned)a;
    ia = sh.i0 + key / nc; ca = key - (key / nc) * nc; ap = pos;
    if (ca == 0) {
        int i1, i2, nd;
        flow_swap_at(cs, sh, rg, pos, 1 << 30, i1, i2, nd);     // (the rejected step's own wave read these positions: covered)
        ap = pos + nd;
    }
}

// every order this wave has out for its chains is void (their positions were predicted in another epoch): the workers are told
__device__ __forceinline__ void flow_void_books(CsRef cs, FlowShared &sh, int wave, int NW, int nc, int lane)
{
    for (int c = wave; c < nc; c += NW) {
        if (sh.ob_pos[c] != -1) {
            if (lane == 0) { void_slot(cs, c); sh.ob_pos[c] = -1; }
        }
    }
}

// this wave adopts epoch e (read from sh.epoch a moment ago).  `standing`: its current step (it, c) has passed its check and lies
// before the anchor -- it stands, and everything the wave runs after it starts at or after the anchor; else the current
// step itself starts at or after the anchor.  Returns false if the epoch moved on meanwhile (the caller looks again).
// (g != nullptr: a multi-block launch -- epoch and anchor are MbShared's; false: the anchor of epoch e is not there (yet, or
// any more): the caller looks at the epoch again)
template <bool MB = false>
__device__ __forceinline__ bool flow_adopt(CsRef cs_, FlowShared &sh, const Ring &rg, FlowWave &W, int e, int it, int c, bool in_turn, bool &stands,
                                           const MbShared *g = nullptr)
{
    CsRef cs = rebase(cs_);
    const int nc = cs.n_chains;
    unsigned long long a;
    if constexpr (MB) {
        const unsigned long long a0 = ld_agent(&g->anch[e & 1][0]), a1 = ld_agent(&g->anch[e & 1][1]);
        if ((int)(unsigned)(a0 >> 32) != e || (int)(unsigned)(a1 >> 32) != e) return false;
        a = (a0 << 32) | (a1 & 0xffffffffull);
    } else {
        a = lds_ld(&sh.anch[e & 1]);
        if (lds_ld(&sh.epoch) != e) return false;
    }
    const int akey = (int)(unsigned)(a >> 32);
    const int key = (it - sh.i0) * nc + c;
    int ia, ca, ap;
    flow_from_anchor(cs, sh, rg, a, ia, ca, ap);
    const int limit = sh.fill;
    stands = in_turn && key < akey;
    W.epoch = e; W.akey = akey;
    if (!stands) {
        // the current step starts at or after the anchor: same iteration, or the anchor sits in the iteration before
        if (ia == it) { W.rc = ca; W.rpos = ap; }
        else { W.rc = 0; W.rpos = flow_next_base<MB>(cs, sh, rg, ap, nc - ca, 1 << 30); }
        W.rc1 = 0; W.rpos1 = flow_next_base<MB>(cs, sh, rg, W.rpos, nc - W.rc, limit);
        W.B2 = flow_next_base<MB>(cs, sh, rg, W.rpos1, nc, limit);
    } else if (ia == it) {
        W.rc = ca; W.rpos = ap;                         // (the wave's later chains of this iteration)
        W.rc1 = 0; W.rpos1 = flow_next_base<MB>(cs, sh, rg, ap, nc - ca, limit);
        W.B2 = flow_next_base<MB>(cs, sh, rg, W.rpos1, nc, limit);
    } else {                                            // the anchor is a step of the next iteration
        W.rc1 = ca; W.rpos1 = ap;
        W.B2 = flow_next_base<MB>(cs, sh, rg, ap, nc - ca, limit);
    }
    return true;
}

// ---- a lock-step rank (k_mcmc<.., 4>, MODE_LOCKRUN): the temperature swap between chains of ANY two ranks (cls_parallel.f90:
// ---- 100-216) without a meeting of the rank's waves.
// The swap after iteration j needs: the pair (rank 0's select_pair, :226-230), judge_swap's draw (from the stream of the rank
// that owns the pair's first chain, :163) and (T, L) of the two chains -- nothing else.  So every CHAIN posts its own (T, L) when it
// commits its step of iteration j, and the wave of the rank's LAST chain posts a HEADER {pair (rank 0), the draw this rank would
// take, stop / failure word} -- tagged 8-byte granules {iteration : 32, payload : 32} written with system-scope stores into every
// rank's inbox (fine-grained memory, over xGMI for a peer).  A chain wave at its decision of iteration j + 1 looks at rank 0's
// header (the pair); only if its own chain is one of the two does it wait for the partner's record (and the draw's owner's
// header) -- the waves of the other chains go on with the temperature they have.  What the free-running loop does inside one
// rank through LDS (L4 / T4 / done) is done between ranks through the inboxes; a partner on the same rank is read from LDS.
// A STOP (record buffers or produced stream nearly used up on some rank) asked for in the headers of iteration j ends the job
// after iteration j + 2 on every rank: every wave reads the headers of iteration i - 2 before it commits a step of iteration i
// (two iterations old: they are there), and all ranks read the same headers.  A FAILURE travels the same way (or as a timeout).
// Ranks other than 0 bet that judge_swap's draw is not theirs (flow_swap_at); the first wave that learns the pair settles the
// bet for the rank (xanch): lost, it anchors the next iteration one position later under a new epoch -- the very mechanism
// of a prior rejection -- before any wave of the rank commits a step of that iteration.
// Inbox: region A [2][n_procs][xg] is the barrier loop's (exchange_post / exchange_finish, htm_step.hpp); region B
// [kXSlots][n_procs][xg], slot = iteration & 7, is this loop's.  Per (slot, rank): granules 0..6 = header {i1, i2, draw hi,
// lo, its log hi, lo, control word}, 8 + 4 c .. 8 + 4 c + 3 = chain c {T hi, lo, L hi, lo}.  Eight slots: a wave still reading
// the headers of iteration j - 4 (its decision of j - 2) may see a peer post those of j (tools/flow_protocol_sim.py found the
// deadlock of a four-slot ring); a tag newer than the one waited for is a fail-stop (-15).
constexpr int kXSlots = 8, kXHdr = 8;
__device__ __forceinline__ size_t flow_xoff(int np, int G, int iter, int rank) { return ((size_t)(2 + (iter & (kXSlots - 1))) * np + rank) * G; }

// chain c's (T, L) after iteration `iter` into every rank's inbox
// (xout: the ranks' inboxes as mapped here, cached in LDS by flow_body; this rank's own waves read its records from LDS, so
// nothing is posted to its own inbox -- one rank alone has no memory traffic for the swap at all)
__device__ __forceinline__ void flow_post_chain(CsRef cs_, unsigned long long *const *xout, int iter, int c, double T, double L, int lane)
{
    CsRef cs = rebase(cs_);
    const int np = cs.n_procs, G = cs.xg;
    if (np == 1) return;
    if (__builtin_expect(cs.dbg_xfail_iter > 0 && iter >= cs.dbg_xfail_iter, 0)) return;      // (fault injection: nobody sees this rank's records)
    const unsigned long long tb = (unsigned long long)__double_as_longlong(T), lb = (unsigned long long)__double_as_longlong(L);
    for (int k = lane; k < 4 * np; k += 64) {
        const int q = k >> 2, g = k & 3;
        const unsigned pay = g == 0 ? (unsigned)(tb >> 32) : g == 1 ? (unsigned)tb : g == 2 ? (unsigned)(lb >> 32) : (unsigned)lb;
        if (q != cs.rank) st_sys(xout[q] + flow_xoff(np, G, iter, cs.rank) + kXHdr + 4 * c + g, ((unsigned long long)(unsigned)iter << 32) | pay);
    }
}
// the rank's header of iteration `iter` (by the wave of its last chain, at its commit: E = where the chain steps end)
__device__ __forceinline__ void flow_post_header(CsRef cs_, FlowShared &sh, unsigned long long *const *xout, int iter, int wmax, int lane)
{
    CsRef cs = rebase(cs_);
    const int np = cs.n_procs, G = cs.xg, nc = cs.n_chains, k4 = iter & 3;
    // (pair and draw as this wave published them with the iteration's end position when its step passed its check)
    const int i1 = sh.sw_i1[k4], i2 = sh.sw_i2[k4];
    const unsigned long long ub = (unsigned long long)__double_as_longlong(sh.sw_r[k4]), lb = (unsigned long long)__double_as_longlong(sh.sw_logr[k4]);
    // this rank asks everybody to stop: record buffers or produced stream nearly used up.  It takes effect two iterations on: until
    // then the rank's other chains may still write their records of this iteration and all chains those of the next two (< 3 n_chains)
    const bool my_stop = sh.c.n_lik + 3 * nc > cs.cap_lik || sh.c.n_smp + 3 * nc > cs.cap_smp || sh.avail < sh.Eof[k4] + 8 * wmax;
    const unsigned ctl = (my_stop ? 1u : 0u) | (sh.c.err ? 2u : 0u);
    if (lane == 0) lds_st(&sh.xctl4[k4], (int)(((unsigned)iter << 2) | ctl));
    if (np == 1) return;
    if (__builtin_expect(cs.dbg_xfail_iter > 0 && iter >= cs.dbg_xfail_iter, 0)) return;
    for (int k = lane; k < 8 * np; k += 64) {
        const int q = k >> 3, g = k & 7;
        const unsigned pay = g == 0 ? (unsigned)i1 : g == 1 ? (unsigned)i2 : g == 2 ? (unsigned)(ub >> 32) : g == 3 ? (unsigned)ub
                           : g == 4 ? (unsigned)(lb >> 32) : g == 5 ? (unsigned)lb : ctl;
        if (g < 7 && q != cs.rank) st_sys(xout[q] + flow_xoff(np, G, iter, cs.rank) + g, ((unsigned long long)(unsigned)iter << 32) | pay);
    }
}
// What a step of iteration `iter` reads from the rank's inbox before its decision, requested when its evaluation is done (a load
// from fine-grained memory takes ~1 us and vector loads return in order: in front of the step's own loads it would hold them up;
// here it runs under the turn): lane q < n_procs, q another rank: rank q's control word of iteration iter - 2; lanes 62, 63, on a
// rank other than 0: the pair in rank 0's header of iteration iter - 1.
__device__ __forceinline__ unsigned long long flow_xload(CsRef cs_, int iter, int lane)
{
    CsRef cs = rebase(cs_);
    const int np = cs.n_procs, G = cs.xg;
    unsigned long long v = 0ull;
    if (lane < np) { if (lane != cs.rank) v = ld_sys(cs.inbox + flow_xoff(np, G, iter - 2, lane) + 6); }
    else if (lane >= 62 && cs.rank != 0) v = ld_sys(cs.inbox + flow_xoff(np, G, iter - 1, 0) + (lane - 62));
    return v;
}
// `n` granules (n <= 64) from `base` in this rank's inbox, all tagged `iter`: polls until they are (false: gave up, sh.c.err set)
__device__ __forceinline__ bool flow_xwait(CsRef cs_, FlowShared &sh, const unsigned long long *base, int n, int iter, int lane, unsigned long long &v)
{
    CsRef cs = rebase(cs_);
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (unsigned spin = 0;; ++spin) {
        v = lane < n ? ld_sys(base + lane) : ((unsigned long long)(unsigned)iter << 32);
        const unsigned tg = (unsigned)(v >> 32);
        if (__all(tg == (unsigned)iter)) return true;
        if (__any(tg != 0u && (int)tg > iter)) { if (lane == 0) sh.c.err = -15; return false; }      // overwritten: the ring is too short
        if ((spin & 7u) == 7u) {
            if (lds_ld(&sh.c.err) != 0) return false;
            if (__builtin_amdgcn_s_memrealtime() - t0 > cs.xwait_ticks) { if (lane == 0) sh.c.err = -10; return false; }
        }
        __builtin_amdgcn_s_sleep(2);
    }
}
// the swap of iteration `it` as chain c of this rank sees it (cls_parallel.f90:121-136, :285-302): is the chain one of the pair,
// and if so does it take the partner's temperature?  T, rT: the chain's temperature during `it` in, after the swap out; L: its
// log-likelihood after `it`.  Both chains' waves (on whichever ranks) evaluate the same expression on the same values.
__device__ __forceinline__ bool flow_lock_swap(CsRef cs_, FlowShared &sh, int it, int c, int i1, int i2, double L, double &T, double &rT, int lane)
{
    CsRef cs = rebase(cs_);
    const int nc = cs.n_chains, np = cs.n_procs, G = cs.xg, par = it & 3;
    const int g = cs.rank * nc + c;
    if (g != i1 && g != i2) return true;
    const int gp = g == i1 ? i2 : i1, rp = gp / nc, cp = gp - rp * nc, r1 = i1 / nc;
    double Tp, rTp, Lp, u, logu;
    if (rp == cs.rank) {
        const int want = (it - sh.i0) * nc + cp;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        for (unsigned spin = 0; lds_ld(&sh.done[cp]) < want; ++spin) {
            if ((spin & 15u) == 15u) {
                if (sh.c.err != 0) return false;
                if (__builtin_amdgcn_s_memrealtime() - t0 > 500000000ull) { if (lane == 0) sh.c.err = -12; return false; }
            }
            __builtin_amdgcn_s_sleep(1);
        }
        Tp = sh.T4[par][cp]; rTp = sh.rT4[par][cp]; Lp = sh.L4[par][cp];
    } else {
        unsigned long long v;
        if (!flow_xwait(cs, sh, cs.inbox + flow_xoff(np, G, it, rp) + kXHdr + 4 * cp, 4, it, lane, v)) return false;
        Tp = gran_f64(rl_u64(v, 0), rl_u64(v, 1)); Lp = gran_f64(rl_u64(v, 2), rl_u64(v, 3));
        rTp = 1.0 / Tp;
    }
    if (r1 == cs.rank) { u = sh.sw_r[par]; logu = sh.sw_logr[par]; }      // (this rank's own draw, published with the iteration's end)
    else {
        unsigned long long v;
        if (!flow_xwait(cs, sh, cs.inbox + flow_xoff(np, G, it, r1) + 2, 4, it, lane, v)) return false;
        u = gran_f64(rl_u64(v, 0), rl_u64(v, 1)); logu = gran_f64(rl_u64(v, 2), rl_u64(v, 3));
    }
    const double rT1 = g == i1 ? rT : rTp, rT2 = g == i1 ? rTp : rT, L1 = g == i1 ? L : Lp, L2 = g == i1 ? Lp : L;
    const double del_s = (L2 - L1) * (rT1 - rT2);                              // :292
    if (u >= kEps && logu <= del_s) { T = Tp; rT = rTp; }                      // :131-136
    return true;
}

// After the loop (wave 0; every step up to the last iteration is committed): the swap of the launch's LAST iteration applied to
// this rank's own chains, the rank's stream position and iteration counter settled -- the state the next launch (or the host) finds.
__device__ __forceinline__ void flow_lock_finish(CsRef cs_, FlowShared &sh, int lane)
{
    CsRef cs = rebase(cs_);
    const int nc = cs.n_chains, np = cs.n_procs, G = cs.xg, i0 = sh.i0;
    const int last = min(sh.last_iter, sh.c.iter_target);
    if (last <= i0) return;
    const int par = last & 3, E = sh.Eof[par];
    int i1 = -1, i2 = -1, extra = 0;
    if (np * nc > 1) {
        if (cs.rank == 0) { i1 = uni(sh.sw_i1[par]); i2 = uni(sh.sw_i2[par]); }
        else {
            unsigned long long v;
            if (!flow_xwait(cs, sh, cs.inbox + flow_xoff(np, G, last, 0), 2, last, lane, v)) return;
            i1 = (int)(unsigned)rl_u64(v, 0); i2 = (int)(unsigned)rl_u64(v, 1);
        }
        for (int k = 0; k < 2; ++k) {
            const int g = k == 0 ? i1 : i2;
            if (g / nc != cs.rank) continue;
            const int c = g - cs.rank * nc;
            double T = sh.T4[par][c], rT = sh.rT4[par][c];      // (the partner, if it is this rank's too, is read from the same rings: its value before the swap)
            if (!flow_lock_swap(cs, sh, last, c, i1, i2, sh.L4[par][c], T, rT, lane)) return;
            if (lane == 0) { sh.temp[c] = T; sh.rtemp[c] = rT; }
        }
        extra = (cs.rank != 0 && i1 / nc == cs.rank) ? 1 : 0;       // judge_swap's draw came from this rank's stream (rank 0 counted it: sw_nd)
    }
    if (lane == 0) {
        sh.c.swap_i1 = i1; sh.c.swap_i2 = i2; sh.c.swap_r = sh.sw_r[par]; sh.c.swap_logr = sh.sw_logr[par];
        sh.c.spos = sh.origin + E + sh.sw_nd[par] + extra;
        sh.c.iter_done = last;
        sh.c.stage = ST_IDLE;
    }
}

// One chain step from its front to its commit and the orders of the chain's coming full evaluations (the free-running
// counterpart of chain_pass).  All 64 lanes execute with identical (uniform) values; lane <-> station only inside
// event_misfit.  `ext`: this wave keeps the LDS window of the stream ahead (chain 0's wave, one round of <= 64 positions
// per step, in flight under the step's own loads).  Returns the stream position after the step, kFlowRestart or kFlowAbort.
template <int NCH, bool F32, bool LOCK, bool MB = false>
__device__ __forceinline__ int flow_step(FwRef f_, CsRef cs_, FlowShared &sh, const Ring &rg, FlowWave &W, FlowNext &nx,
                                         MbShared *g_mb, MbWave &mw, double *s_gath, int wmax,
                                         const double *s_sx, const double *s_sy, const double *s_sz, int c, int p, int iter,
                                         int lane, int wave, int NW, unsigned long long launch, bool ext, int look, int back,
                                         bool rec_now, const FlowTop &tp)
{
    CsRef cs = rebase(cs_);
    FwRef f = rebase(f_);
    const int M = rg.mask;
    p = __builtin_amdgcn_readfirstlane(p);
#ifdef HTM_STAMPS
    unsigned long long st_acc[7] = {0, 0, 0, 0, 0, 0, 0};      // ([6]: a lock-step rank's look at the inbox, the bet and the pair's wait)
    unsigned long long t_last = __builtin_amdgcn_s_memtime(), t_wait = 0;
    const unsigned long long t_step0 = t_last;
#endif
    // ---- the LDS window of the stream rings, one round per step of chain 0's wave: requested first, stored behind the
    // ---- step's own loads (which return after it: vector-memory operations complete in order)
    PfRegs pf, pf2;
    pf.p = -1; pf2.p = -1;
    int fill_to = 0;
    if (ext) {
        const int fl = sh.fill;
        // (this step is chain c's: the iteration's base lies 4 c .. 6 c positions back; the other waves may still read `back`
        // positions behind it, and want `look` positions ahead of it)
        fill_to = min(min(p - 4 * c + look, sh.avail), p - 6 * c - back + M + 1);
        // (several master workgroups: the wave has ONE step per iteration for all the draws of the rank's chains -- two rounds)
        if (fill_to > fl + (MB ? 128 : 64)) fill_to = fl + (MB ? 128 : 64);
        if (fill_to > fl) pf_load(pf, cs, sh, fl + lane, fill_to);
        if constexpr (MB) { if (fill_to > fl + 64) pf_load(pf2, cs, sh, fl + 64 + lane, fill_to); }
    }
    const double *xall_ = cs.xall;
    const PriorRec *prior_ = cs.prior;
    const int nc_ = cs.n_chains, S_ = cs.S, nh = 3 * cs.E, psame_ = cs.prior_same;
    asm volatile("" : "+s"(xall_), "+s"(prior_));
    // the decoded proposal (htm_stream.hpp), its Gaussian and its judge draw: looked up during the step before (FlowNext), or here
    int type, idx, evt, dec_w;
    double g, r_ring, logr_ring;
    if (nx.p == p && nx.it == iter && nx.c == c && nx.epoch == W.epoch) {
        type = nx.type; idx = nx.idx; evt = nx.evt; dec_w = nx.dec_w; g = nx.g; r_ring = nx.r; logr_ring = nx.logr;
    } else {
        const i32x4 dec = reinterpret_cast<const i32x4 *>(rg.dec)[p & M];
        // wave-uniform by construction: keep them in scalar registers (addresses and selects become SALU work)
        type = __builtin_amdgcn_readfirstlane(dec.x); idx = __builtin_amdgcn_readfirstlane(dec.y);
        evt = __builtin_amdgcn_readfirstlane(dec.z); dec_w = __builtin_amdgcn_readfirstlane(dec.w);
        g = rg.pg[p & M]; r_ring = rg.pr[p & M]; logr_ring = rg.plogr[p & M];
    }
    nx.p = -1;
    const bool partial = evt > 0 && iter > 1;       // hypo_tremor_mcmc.f90:246
    const int off_tc = nc_, off_qs = nc_ + nc_ * S_, off_ac = 2 * nc_ + nc_ * S_, off_hy = 2 * nc_ + 2 * nc_ * S_;
    const int goff = type == 1 ? 0 : type == 2 ? off_tc : type == 3 ? off_qs : type == 4 ? off_ac : off_hy;
    const int gnx = (type == 1 || type == 3) ? 1 : (type == 2 || type == 4) ? S_ : nh;
    const int o = goff + c * gnx + idx;             // element of the rank's parameter vector this step perturbs
    const int ev = partial ? evt - 1 : 0;
    const int o_h = off_hy + c * nh + 3 * ev;
    int goffs = o;
    goffs = lane == 1 ? o_h : goffs; goffs = lane == 2 ? o_h + 1 : goffs; goffs = lane == 3 ? o_h + 2 : goffs;
    goffs = lane == 4 ? c : goffs; goffs = lane == 5 ? off_qs + c : goffs;
    const double gathered_v = xall_[goffs];
    const PriorRec prr = ld_prior(prior_ + (psame_ ? o - c * gnx : o));      // (one 32-byte scalar load: PriorRec)
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
    // the book of this chain: is this step's order out already, and how
    const int book_pos = tp.book_pos, book_mode = tp.book_mode, book_mid = tp.book_mid;
    const unsigned book_tag = tp.book_tag;
    const bool pre = !partial && book_pos == p;
    const int pre_mode = pre ? book_mode : 0;
    // A full evaluation whose chain's PREVIOUS step was a hypocentre step leaves that step's event to this wave: the workers sum
    // the other events, the wave adds this one (below) -- always, whether the order went out two steps ahead (the workers could
    // not know that step's outcome), one step ahead or only now.  The association of the sum is then a function of the stream
    // alone, not of when the order happened to go out: two runs of a job give the same bits.  The event's inputs are requested
    // now, into the registers a partial update would use.  (pv_mid and an order's book_mid name the same step.)
    int d_e = 0;
    double d_ex = 0.0, d_ey = 0.0, d_ez = 0.0;
    const int mid = pre ? book_mid : tp.pv_mid;
    const bool own_evt = NCH > 0 && !partial && type <= 4 && (mid & 7) >= 5;
    if constexpr (NCH > 0) {
        if (__builtin_expect(own_evt, 0)) {
            d_e = __builtin_amdgcn_readfirstlane(mid >> 3) - 1;
            const int vzd = opaque_zero();
            const double *hypd = xall_ + off_hy + c * nh + 3 * d_e;
            d_ex = ld_state(hypd, vzd); d_ey = ld_state(hypd + 1, vzd); d_ez = ld_state(hypd + 2, vzd);
            load_sta_regs<NCH>(st, f.S, lane, s_sx, s_sy, s_sz, tc, ac, 0, -1, 0.0);
            load_obs_regs<NCH, F32>(ob, f, d_e, lane);
        }
    }
    FSTAMP(0);
    const double x_old = rl_f64(gathered_v, 0);
    const double hx = rl_f64(gathered_v, 1), hy = rl_f64(gathered_v, 2), hz = rl_f64(gathered_v, 3);
    const double beta = rl_f64(gathered_v, 4), q = rl_f64(gathered_v, 5);
    const double L_cur = tp.L;
    const double x_new = x_old + g * step;                      // cls_model.f90:172
    const double da = x_new - mu, db = x_old - mu;
    double lpr = -(da * da - db * db) * rs2;                    // :175-177 (rs2 = 1 / (2 sigma^2), formed once on the host)
    int ok = 1;
    if (ptype == 1) {                                           // :178-187
        if (x_new <= mu) { lpr = (double)-1.0e+30f; ok = 0; }
        else lpr = lpr + htm_log(x_new - mu) - htm_log(x_old - mu);       // (:184-185; htm_device.hpp: < 0.75 ulp, a third of the library routine's instructions)
    }
    const double r = ok ? r_ring : 0.0, logr = ok ? logr_ring : 0.0;
    const int cnt = dec_w - 1 + ok;                             // the judge draw happens only if prior_ok
    const int key = (iter - sh.i0) * nc_ + c;
    // ---- the step has passed (or failed) its prior check: the later steps may go ahead on it
    if (c == nc_ - 1) {                                         // where this iteration's swap starts, and what it draws there
        const int E = p + cnt, k4 = iter & 3;
        int i1, i2, nd;
        flow_swap_at(cs, sh, rg, E, 1 << 30, i1, i2, nd);
        if (lane == 0) {
            sh.sw_i1[k4] = i1; sh.sw_i2[k4] = i2; sh.sw_nd[k4] = nd;
            if constexpr (LOCK) {
                // the draw THIS rank takes if the pair's first chain is its own (rank 0 knows: then it is the last of the nd draws;
                // another rank peeks at the position after its chain steps): what goes into the rank's header
                const int own = (cs.rank == 0 && i1 >= 0 && i1 / nc_ == 0) ? 1 : 0;
                sh.sw_r[k4] = rg.U[(E + nd - own) & M]; sh.sw_logr[k4] = rg.LOGU[(E + nd - own) & M];
            } else
            if (nd > 0) { sh.sw_r[k4] = rg.U[(E + nd - 1) & M]; sh.sw_logr[k4] = rg.LOGU[(E + nd - 1) & M]; }
            lds_st(&sh.Eof[k4], E);
        }
        if constexpr (MB) {
            // (the same, for the chains of the other workgroups: eight granules tagged with the iteration)
            const int je = nd > 0 ? (E + nd - 1) & M : 0;
            const unsigned long long rb = nd > 0 ? (unsigned long long)__double_as_longlong(rg.U[je]) : 0ull;
            const unsigned long long lb = nd > 0 ? (unsigned long long)__double_as_longlong(rg.LOGU[je]) : 0ull;
            if (lane < 8) {
                const unsigned pay = lane == 0 ? (unsigned)E : lane == 1 ? (unsigned)i1 : lane == 2 ? (unsigned)i2 : lane == 3 ? (unsigned)nd
                                   : lane == 4 ? (unsigned)(rb >> 32) : lane == 5 ? (unsigned)rb : lane == 6 ? (unsigned)(lb >> 32) : (unsigned)lb;
                // (tag = epoch << 24 | iteration: a step run again under a new epoch posts a new version; the reader knows which it wants)
                st_agent(&g_mb->swrec[k4][lane], ((unsigned long long)((((unsigned)W.epoch & 0xffu) << 24) | ((unsigned)iter & 0xffffffu)) << 32) | pay);
            }
        }
    }
    {
        const unsigned long long pword = ((unsigned long long)(((unsigned)W.epoch << 1) | (ok ? 0u : 1u)) << 32) | (unsigned)key;
        if (lane == 0) lds_st(&sh.prog[c], pword);                       // (the chains of this workgroup read it here)
        if constexpr (MB) { if (lane == 0) st_agent(&g_mb->prog[c], pword); }      // (the chains of the other workgroups)
    }
    // several master workgroups: this step's look at what the chains share (MbShared), in flight under the evaluation; with it
    // the swap the stream holds after the iteration before (read for good when the turn has come)
    unsigned long long mbv = 0ull, mbs = 0ull;
    if constexpr (MB) {
        mbv = mb_look(g_mb, nc_, lane);
        if (lane < 8) mbs = ld_agent(&g_mb->swrec[(iter - 1) & 3][lane]);
    }
    if (ext && fill_to > sh.fill) {                             // (the window's loads were issued before the step's: they are there)
        pf_store(pf, rg);
        if constexpr (MB) pf_store(pf2, rg);
        if (lane == 0) lds_st(&sh.fill, fill_to);
    }

    FSTAMP(1);
    double L_new = 0.0;
    int need_full = 0;
    int la = 0, la_epoch = 0, hA = 0, hB = 0, hE = 0, la_p1 = 0, la_p2 = 0, la_E2 = 0;
    i32x4 d1v = {0, 0, 0, 0}, d2v = {0, 0, 0, 0}, swv = {0, 0, 0, 0};
    double la_g = 0.0, la_r = 0.0, la_logr = 0.0;
    if (__builtin_expect(ok != 0, 1)) {
        if (__builtin_expect(partial, 1)) {
            const int cmp = idx - 3 * ev;        // 0 x, 1 y, 2 z of event ev (selects: see htm_step.hpp)
            const double px[2] = {hx, cmp == 0 ? x_new : hx};
            const double py[2] = {hy, cmp == 1 ? x_new : hy};
            const double pz[2] = {hz, cmp == 2 ? x_new : hz};
            double out[2];
            // look-ahead, first round trip (a wave with one chain): the hop-table entries that give the start of this chain's
            // next step, of its step after that, and the end of the iteration after next -- issued here, used behind the evaluation
            // (the hop tables reach kHops steps: with more chains than that -- several master workgroups -- the positions are looked
            // up at the top of the next step instead)
            if (NW >= nc_ && W.rpos1 >= 0 && W.B2 >= 0 && c >= W.rc1 && iter + 1 <= sh.c.iter_target && rg.mir_n > 0) {
                la = 1; la_epoch = W.epoch;
                const int n1 = c - W.rc1;
                // (one workgroup: a wave has one chain only with <= 8 = kHops chains, the tables reach in one entry)
                if (!MB || nc_ <= kHops) {
                    if (n1 > 0) hA = rg.hop[(W.rpos1 & M) * kHops + n1 - 1];
                    if (c > 0) hB = rg.hop[(W.B2 & M) * kHops + c - 1];
                    hE = rg.hop[(W.B2 & M) * kHops + nc_ - 1];
                } else {
                    // (more chains than the hop tables reach in one entry -- several master workgroups: the positions by
                    // chained look-ups, here; what the stream holds there is still read under the evaluation)
                    // (every table entry read must lie inside the window: a position past it holds another iteration's numbers)
                    const int limh = uni(sh.fill) - 16;
                    auto hops = [&](int pos, int n) __attribute__((always_inline)) {
                        int q = pos;
                        while (n > kHops) { if (q >= limh) return -1; q += rg.hop[(q & M) * kHops + kHops - 1]; n -= kHops; }
                        if (n > 0) { if (q >= limh) return -1; q += rg.hop[(q & M) * kHops + n - 1]; }
                        return q;
                    };
                    la_p1 = hops(W.rpos1, n1); la_p2 = hops(W.B2, c); la_E2 = hops(W.B2, nc_);
                    la = (la_p1 >= 0 && la_p2 >= 0 && la_E2 >= 0) ? 3 : 0;
                }
            }
            if constexpr (NCH > 0) event_misfit<NCH, 2, F32, true>(f, ob, lane, st, px, py, pz, tp.rbeta, tp.katt, out);
            else event_misfit_generic<2>(f, ev, lane, s_sx, s_sy, s_sz, tc, ac, 0, -1, 0.0, px, py, pz, beta, q, out);
            L_new = L_cur + wave_sum1(out[0] - out[1]);
            // look-ahead, second round trip: what the stream holds at those positions -- used after the commit
            if (la) {
                if (!MB || la == 1) { la_p1 = W.rpos1 + uni(hA); la_p2 = W.B2 + uni(hB); la_E2 = W.B2 + uni(hE); }
                else { la_p1 = uni(la_p1); la_p2 = uni(la_p2); la_E2 = uni(la_E2); }
                const int lim = uni(sh.fill) - 16;
                if (la_p1 < lim && la_p2 < lim && la_E2 + 16 < lim) {
                    la = 2;
                    d1v = reinterpret_cast<const i32x4 *>(rg.dec)[la_p1 & M];
                    d2v = reinterpret_cast<const i32x4 *>(rg.dec)[la_p2 & M];
                    swv = reinterpret_cast<const i32x4 *>(rg.sw)[la_E2 & M];
                    la_g = rg.pg[la_p1 & M]; la_r = rg.pr[la_p1 & M]; la_logr = rg.plogr[la_p1 & M];
                }
            }
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
            if (!pre && lane < cs.slot_rep * kGranPerSlot) {
                const int gi = lane & 7;
                const unsigned long long xb = (unsigned long long)__double_as_longlong(x_new);
                const unsigned pay = gi == 0 ? (unsigned)launch : gi == 1 ? ((unsigned)type | ((unsigned)idx << 3))
                                   : gi == 2 ? (unsigned)(xb >> 32) : gi == 3 ? (unsigned)xb
                                   : gi == 4 ? 0xffffffffu                // no commit to wait for (drained above)
                                   : gi == 7 ? (own_evt ? (unsigned)(off_hy + c * nh + 3 * d_e) + 1u : 0u)      // the previous step's event is this wave's
                                   : 0u;
                st_gran(cs.slots + (size_t)(lane >> 3) * cs.slot_stride + c * kGranPerSlot + gi, tag, pay);
            }
            // ---- the workers' partial sums: tagged granules, fixed summation order; two rounds of loads in flight --
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
            // an order sent ahead (one or two steps) may have been answered already: its granules are requested now, under the
            // evaluation of the wave's own event
            const bool early = pre;
            if (early) issue(0);
            // The event of the chain's previous step, if that was a hypocentre step: the workers LEFT IT OUT (an order sent two
            // steps ahead was evaluated while that step may or may not have committed), this wave adds its misfit -- at the
            // position the event has now, under this step's proposed parameters.  The result depends neither on when the
            // workers looked nor on when the order went out.
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
                    event_misfit<NCH, 1, F32>(f, ob, lane, st, pxd, pyd, pzd, type == 1 ? x_new : beta, type == 3 ? x_new : q, outd);
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
                if (j * 64 < cs.n_wg)
                    part += (j * 64 + lane < cs.n_wg) ? (which == 0 ? gran_f64(hi[0][j], lo[0][j]) : gran_f64(hi[1][j], lo[1][j])) : 0.0;
            L_new = -wave_sum1(part + own_lane) - f.const_sum;       // cls_forward.f90:277-300
        }
    }
    FSTAMP(2);
    // a lock-step rank: the other ranks' stop words of two iterations ago and rank 0's pair of the iteration before (flow_xload)
    unsigned long long xv = 0ull;
    if constexpr (LOCK) { if (cs.n_procs > 1) xv = flow_xload(cs, iter, lane); }

    // ---- the step's turn: every step before it in stream order has passed its check in this epoch (or lies before the
    // ---- epoch's anchor: checked earlier, final).  Lanes <-> chains.
    {
        const int key_i = (iter - sh.i0) * nc_, key_m = key_i - nc_;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        for (unsigned spin = 0;; ++spin) {
            unsigned long long pv;
            int e;
            if constexpr (MB) {
                // the chains of this workgroup: LDS, as with one workgroup; the others' and the shared words: the look at memory
                // (the first round uses the one issued before the evaluation; a constant lag between the workgroups is all the
                // round trip costs: the steps of the OTHER workgroups a turn waits for are mostly an iteration old)
                if (spin > 0) mbv = mb_look(g_mb, nc_, lane);
                mb_take(mw, mbv);
                pv = lane < nc_ ? ((lane >> 3) == (c >> 3) ? lds_ld(&sh.prog[lane]) : mbv) : 0ull; e = mw.epoch;
                if (__builtin_expect(mw.err != 0, 0)) { if (lane == 0 && sh.c.err == 0) sh.c.err = mw.err; return kFlowAbort; }
                if (__builtin_expect(iter > mw.last_iter, 0)) return kFlowStop;      // (the launch ends before this iteration: the step is not taken)
            } else {
                pv = lane < nc_ ? lds_ld(&sh.prog[lane]) : 0ull;
                e = lds_ld(&sh.epoch);
            }
            if (__builtin_expect(e != W.epoch, 0)) {
                bool stands = false;
                if (!flow_adopt<MB>(cs, sh, rg, W, e, iter, c, true, stands, g_mb)) { if (MB) __builtin_amdgcn_s_sleep(1); continue; }
                if constexpr (MB) MB_HIST(mw, 0xC | (e & 3));
                flow_void_books(cs, sh, wave, NW, nc_, lane);
                if (!stands) return kFlowRestart;
                continue;
            }
            const int need = (lane < c ? key_i : key_m) + lane;
            const int pk = (int)(unsigned)pv;
            const unsigned ph = (unsigned)(pv >> 32);
            const bool okl = lane >= nc_ || lane == c || pk > need ||
                             (pk == need && (pk < W.akey || (ph == ((unsigned)W.epoch << 1))));
            if constexpr (MB) {
                // A check published under an epoch this wave has not seen yet: the epoch word (memory, a round trip old) lags the
                // checks (LDS for this workgroup's chains, or simply a later store) -- a chain that is already past the anchor
                // must not be taken for "ahead" before the anchor has been adopted.  Look again until the epoch shows.
                if (__any(lane < nc_ && lane != c && (int)(ph >> 1) > W.epoch)) {
                    if ((spin & 15u) == 15u && __builtin_amdgcn_s_memrealtime() - t0 > 500000000ull) { if (lane == 0) sh.c.err = -16; return kFlowAbort; }
                    __builtin_amdgcn_s_sleep(1);
                    continue;
                }
            }
            if (__all(okl)) break;
            if constexpr (LOCK) {
                // the job may have stopped after the iteration before (a wave that learnt it from the swap records has left
                // without taking its step of this iteration -- whose check this turn would wait for)
                if (iter > lds_ld(&sh.last_iter)) return kFlowStop;
            }
            if ((spin & 15u) == 15u) {
                if (sh.c.err != 0) return kFlowAbort;
                if (__builtin_amdgcn_s_memrealtime() - t0 > 500000000ull) {
                    if constexpr (MB) {      // (what the turn was waiting for: the host puts it into its message)
                        const unsigned long long bad = __ballot(!okl);
                        if (lane == 0) { unsigned long long *dg = cs.diag; dg[1] = c; dg[2] = iter; dg[3] = bad; dg[4] = W.epoch; dg[5] = mw.epoch; dg[6] = rl_u64(pv, __ffsll((long long)bad) - 1); dg[0] = 2; }
                    }
                    if (lane == 0) sh.c.err = MB ? -16 : -12;
                    return kFlowAbort;
                }
            }
            __builtin_amdgcn_s_sleep(HTM_TURN_SLEEP);
        }
    }
    FSTAMP(3);
    // ---- the temperature of this iteration: the swap of the iteration before (cls_parallel.f90:121-136, :285-302), decided
    // ---- here by the waves of the two chains it concerns -- each evaluates the same expression on the same values
    const int par = iter & 3, ppar = (iter - 1) & 3;
    double T = sh.T4[par][c], rT = sh.rT4[par][c];        // (first iteration of a launch: written by the prologue)
    if constexpr (LOCK) {
        // a lock-step rank (see flow_post_chain): the stop words of iteration iter - 2, the pair of iteration iter - 1, the bet on
        // judge_swap's draw; then -- only if this chain is one of the pair -- the partner's (T, L)
        const int np = cs.n_procs, i0_ = sh.i0;
        // (what the block reads from LDS, requested in one batch: each of these used to be a round trip of its own behind a branch)
        T = sh.temp[c]; rT = sh.rtemp[c];
        const int own = sh.xctl4[(iter - 2) & 3];
        int l_it = sh.last_iter, ep = sh.epoch;
        const int si1 = sh.sw_i1[ppar], si2 = sh.sw_i2[ppar], xa = sh.xanch;
        asm volatile("" ::: "memory");
        if (np > 1) {
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            for (unsigned spin = 0;; ++spin) {
                const unsigned tg = (unsigned)(xv >> 32);
                bool okl = true;
                if (lane < np) okl = iter - 2 <= i0_ || lane == cs.rank || tg == (unsigned)(iter - 2);
                else if (lane >= 62) okl = iter - 1 <= i0_ || cs.rank == 0 || tg == (unsigned)(iter - 1);
                if (__all(okl)) break;
                if (__any(((lane < np && lane != cs.rank) || (lane >= 62 && cs.rank != 0)) && tg != 0u && (int)tg > iter)) { if (lane == 0) sh.c.err = -15; return kFlowAbort; }
                if ((spin & 7u) == 7u) {
                    if (lds_ld(&sh.c.err) != 0) return kFlowAbort;
                    if (iter > lds_ld(&sh.last_iter)) return kFlowStop;
                    if (__builtin_amdgcn_s_memrealtime() - t0 > cs.xwait_ticks) { if (lane == 0) sh.c.err = -10; return kFlowAbort; }
                }
                __builtin_amdgcn_s_sleep(2);
                xv = flow_xload(cs, iter, lane);
            }
        }
        if (iter - 2 > i0_) {
            // (this rank's own word: LDS.  Its last chain's wave finished iteration iter - 2 before it began the step of iter - 1
            // whose check this step's turn has seen)
            if (__builtin_expect((own >> 2) != iter - 2, 0)) { if (lane == 0) sh.c.err = -12; return kFlowAbort; }
            const unsigned ctl = lane == cs.rank ? (unsigned)own & 3u : lane < np ? (unsigned)xv : 0u;
            if (__any((ctl & 2u) != 0u)) { if (lane == 0 && sh.c.err == 0) sh.c.err = -11; return kFlowAbort; }      // a peer reported a failure
            if (__any((ctl & 1u) != 0u)) {      // somebody asked for a stop two iterations ago: this iteration is the job's last (on every rank)
                if (lane == 0) { atomicMin(&sh.last_iter, iter); sh.c.stop = 3; }
                l_it = min(l_it, iter);
            }
        }
        if (iter > uni(l_it)) return kFlowStop;
        if (iter - 1 > i0_ && np * nc_ > 1) {
            int i1, i2;
            if (cs.rank == 0) { i1 = uni(si1); i2 = uni(si2); }      // (this step's turn has seen the last chain's check)
            else { i1 = (int)(unsigned)rl_u64(xv, 62); i2 = (int)(unsigned)rl_u64(xv, 63); }
            if (cs.rank != 0 && uni(xa) < iter - 1) {
                // the bet that the draw is not this rank's, settled once per iteration by whoever learns the pair first
                int mine = 0;
                if (lane == 0) mine = atomicMax(&sh.xanch_claim, iter - 1) < iter - 1 ? 1 : 0;
                if (uni(mine)) {
                    if (lane == 0) {
                        if (i1 / nc_ == cs.rank) {
                            // lost: the next iteration starts one position later (anchor of a step 0: flow_from_anchor adds what this
                            // rank predicts the swap to draw there -- nothing)
                            const int e1 = sh.epoch + 1;
                            lds_st(&sh.anch[e1 & 1], ((unsigned long long)(unsigned)((iter - i0_) * nc_) << 32) | (unsigned)(sh.Eof[ppar] + 1));
                            lds_st(&sh.epoch, e1);
                        }
                        lds_st(&sh.xanch, iter - 1);
                    }
                } else {
                    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                    for (unsigned spin = 0; lds_ld(&sh.xanch) < iter - 1; ++spin) {
                        if ((spin & 15u) == 15u) {
                            if (sh.c.err != 0) return kFlowAbort;
                            if (__builtin_amdgcn_s_memrealtime() - t0 > 500000000ull) { if (lane == 0) sh.c.err = -12; return kFlowAbort; }
                        }
                        __builtin_amdgcn_s_sleep(1);
                    }
                }
                ep = lds_ld(&sh.epoch);
            }
            if (uni(ep) != W.epoch) {
                bool stands = false;
                while (!flow_adopt(cs, sh, rg, W, lds_ld(&sh.epoch), iter, c, true, stands)) { }
                flow_void_books(cs, sh, wave, NW, nc_, lane);
                if (!stands) return kFlowRestart;
            }
#ifdef HTM_STAMPS_SPLIT
            FSTAMP(6);
#endif
            const double T_was = T;
            if (!flow_lock_swap(cs, sh, iter - 1, c, i1, i2, L_cur, T, rT, lane)) return kFlowAbort;
            if (T != T_was && lane == 0) { sh.temp[c] = T; sh.rtemp[c] = rT; }
        }
        // the rank's header of this iteration, from the wave of its last chain: where the chain steps end is final (the turn), the bet
        // on the draw of the iteration before is settled (it moves a rank's stream, and with it the draw the rank would take)
        if (c == nc_ - 1) flow_post_header(cs, sh, reinterpret_cast<unsigned long long *const *>(s_gath), iter, wmax, lane);
#ifdef HTM_STAMPS_SPLIT      // (diagnostics: [6] = the block up to the pair's wait; the wait and the header go to phase 4)
        FSTAMP(4);
#else
        FSTAMP(6);
#endif
    } else if (MB && iter - 1 > sh.i0) {
        // several master workgroups: the pair and the draw from MbShared::swrec (requested before the evaluation), the other
        // chain's (T, L) from its record -- this chain's own from LDS, where its wave keeps them
        T = sh.T4[ppar][c]; rT = sh.rT4[ppar][c];
        if (nc_ > 1) {
            const bool sw_local = ((nc_ - 1) >> 3) == (c >> 3);      // the rank's last chain is this workgroup's: its LDS has the swap
            int i1, i2;
            double sr, slr;
            if (sw_local) { i1 = uni(lds_ld(&sh.sw_i1[ppar])); i2 = uni(sh.sw_i2[ppar]); sr = sh.sw_r[ppar]; slr = sh.sw_logr[ppar]; }
            else {
                // WHICH version of the record: the one posted with the last chain's check of iteration iter - 1 as this step's
                // turn has seen it -- its epoch is in that check's word.  (The record read before the evaluation may be the one
                // of an earlier run of that step.)  The last chain already past that step: its record is final and long
                // there -- read again now.
                const unsigned long long lv = rl_u64(mbv, nc_ - 1);
                const int needL = (iter - 1 - sh.i0) * nc_ + nc_ - 1;
                if ((int)(unsigned)lv == needL) {
                    const unsigned want = ((((unsigned)(lv >> 33)) & 0xffu) << 24) | ((unsigned)(iter - 1) & 0xffffffu);
                    if (!__all(lane >= 8 || (unsigned)(mbs >> 32) == want)) {
                        if (!mb_wait(sh, &g_mb->swrec[ppar][0], 8, want, lane, mbs, -17)) return kFlowAbort;
                    }
                } else {
                    if (!mb_wait(sh, &g_mb->swrec[ppar][0], 8, (unsigned)(iter - 1) & 0xffffffu, lane, mbs, -17, 0xffffffu)) return kFlowAbort;
                }
                i1 = (int)(unsigned)rl_u64(mbs, 1); i2 = (int)(unsigned)rl_u64(mbs, 2);
                sr = gran_f64(rl_u64(mbs, 4), rl_u64(mbs, 5)); slr = gran_f64(rl_u64(mbs, 6), rl_u64(mbs, 7));
            }
            if (c == i1 || c == i2) {
                const int o2 = c == i1 ? i2 : i1;
                double To, Lo, rTo;
                if ((o2 >> 3) == (c >> 3)) {      // the other chain is this workgroup's too: LDS (as with one workgroup)
                    const int want = (iter - 1 - sh.i0) * nc_ + o2;
                    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                    for (unsigned spin = 0; lds_ld(&sh.done[o2]) < want; ++spin) {
                        if ((spin & 15u) == 15u) {
                            if (sh.c.err != 0) return kFlowAbort;
                            if (__builtin_amdgcn_s_memrealtime() - t0 > 500000000ull) { if (lane == 0) sh.c.err = -12; return kFlowAbort; }
                        }
                        __builtin_amdgcn_s_sleep(1);
                    }
                    To = sh.T4[ppar][o2]; rTo = sh.rT4[ppar][o2]; Lo = sh.L4[ppar][o2];
                } else {
                    unsigned long long cv;
                    if (!mb_wait(sh, &g_mb->crec[ppar][o2][0], 4, (unsigned)(iter - 1), lane, cv, -18)) return kFlowAbort;
                    To = gran_f64(rl_u64(cv, 0), rl_u64(cv, 1)); Lo = gran_f64(rl_u64(cv, 2), rl_u64(cv, 3)); rTo = 1.0 / To;
                }
                const double rT1 = c == i1 ? rT : rTo, rT2 = c == i1 ? rTo : rT, L1 = c == i1 ? L_cur : Lo, L2 = c == i1 ? Lo : L_cur;
                const double del_s = (L2 - L1) * (rT1 - rT2);                            // :292
                if (sr >= kEps && slr <= del_s) { T = To; rT = rTo; }                    // :131-136
            }
        }
    } else if (iter - 1 > sh.i0) {
        T = sh.T4[ppar][c]; rT = sh.rT4[ppar][c];
        if (cs.n_procs * nc_ > 1) {
            // (written by the last chain's wave before it published its check; this step's turn has seen that check)
            const int i1 = lds_ld(&sh.sw_i1[ppar]), i2 = sh.sw_i2[ppar];
            if (c == i1 || c == i2) {
                const int o2 = c == i1 ? i2 : i1;
                const int want = (iter - 1 - sh.i0) * nc_ + o2;
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
    const int acc = (ok != 0 && metropolis(L_new, L_cur, rT, lpr, r, logr)) ? 1 : 0;       // cls_mcmc.f90:193-203
    // this wave's chain-state stores of EARLIER steps have landed before this step's commit goes out: an order sent after
    // the commit names only the commit itself for the workers to wait for
    drain_vmem();
    const int cool = (T < 1.0 + kEps) ? 1 : 0;
    const double L_post = acc ? L_new : L_cur;
    if (lane == 0) {
        sh.T4[par][c] = T; sh.rT4[par][c] = rT;
        if (ok != 0) atomicAdd(need_full ? &sh.n_full_w : &sh.n_part_w, 1ull);      // (a step that is run again left before this point)
        if (cool) sh.np[c * 7 + type - 1] += 1;                 // cls_mcmc.f90:186-189
        if (acc) {                                              // :207-219
            st_agent(cs.xall + o, x_new);
            if (o < rg.mir_n) rg.mx[o] = x_new;
            sh.L[c] = L_new;
            if (cool) sh.na[c * 7 + type - 1] += 1;
            if (__builtin_expect(type == 1 || type == 3, 0)) {      // a new vs or qs: the chain's two reciprocals with it
                const double b_ = type == 1 ? x_new : beta, q_ = type == 3 ? x_new : q;
                sh.rbeta[c] = 1.0 / b_; sh.katt[c] = (kPi * kFreq) / (q_ * b_);
            }
        }
        sh.L4[par][c] = L_post;
        lds_st(&sh.done[c], key);
    }
    if constexpr (LOCK) {
        // this chain's (T, L) after the iteration (the rank's header went out before the decision, from the wave of its last chain)
        unsigned long long *const *xout = reinterpret_cast<unsigned long long *const *>(s_gath);
        flow_post_chain(cs, xout, iter, c, T, L_post, lane);
    }
    if constexpr (MB) {
        // this chain after the iteration, for the chains of the other workgroups (the pair of the swap that follows)
        const unsigned long long tb = (unsigned long long)__double_as_longlong(T), lb = (unsigned long long)__double_as_longlong(L_post);
        if (lane < 4) {
            const unsigned pay = lane == 0 ? (unsigned)(tb >> 32) : lane == 1 ? (unsigned)tb : lane == 2 ? (unsigned)(lb >> 32) : (unsigned)lb;
            st_agent(&g_mb->crec[par][c][lane], ((unsigned long long)(unsigned)iter << 32) | pay);
        }
    }
    FSTAMP(4);
    // ---- a rejected prior: this step was one draw shorter than the hop tables assume.  Everything after it starts
    // ---- elsewhere: new epoch, anchored at the step after this one
    if (__builtin_expect(ok == 0, 0)) {
        const int e1 = W.epoch + 1;
        bool stands = false;
        if constexpr (MB) {
            if (lane == 0) {
                st_agent(&g_mb->anch[e1 & 1][0], ((unsigned long long)(unsigned)e1 << 32) | (unsigned)(key + 1));
                st_agent(&g_mb->anch[e1 & 1][1], ((unsigned long long)(unsigned)e1 << 32) | (unsigned)(p + cnt));
                st_agent(&g_mb->word[MW_EPOCH], (unsigned long long)(unsigned)e1);
            }
            // (this wave's own view: from what it has just written; a later rejection finds it at its next look)
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            while (!flow_adopt<true>(cs, sh, rg, W, e1, iter, c, true, stands, g_mb)) {
                if (__builtin_amdgcn_s_memrealtime() - t0 > 500000000ull) { if (lane == 0) sh.c.err = -19; return kFlowAbort; }
                __builtin_amdgcn_s_sleep(1);
            }
            mw.epoch = e1;
        } else {
        if (lane == 0) {
            lds_st(&sh.anch[e1 & 1], ((unsigned long long)(unsigned)(key + 1) << 32) | (unsigned)(p + cnt));
            lds_st(&sh.epoch, e1);
        }
        // (this wave's own view: as any wave whose step stands before the anchor; a later rejection may already have moved on)
        while (!flow_adopt(cs, sh, rg, W, lds_ld(&sh.epoch), iter, c, true, stands)) { }
        }
        flow_void_books(cs, sh, wave, NW, nc_, lane);
    }
    // ---- records of this step (hypo_tremor_mcmc.f90:270-280): slots by LDS atomics, put in order on the host
    if (__builtin_expect(sh.c.slog_cap > 0, 0)) {
        const int row = sh.c.slog_n + (iter - sh.i0 - 1) * nc_ + c;
        if (lane == 0 && row < sh.c.slog_cap) {
            int32_t *ir = cs.slog_i + 8 * (size_t)row;
            double *dr = cs.slog_d + 4 * (size_t)row;
            ir[0] = iter; ir[1] = c; ir[2] = type; ir[3] = idx + 1; ir[4] = ok; ir[5] = acc; ir[6] = need_full;
            ir[7] = 0;
#ifdef HTM_MB_DIAG
            // (several master workgroups, diagnostic build: where the step started, the epoch it was committed in, the step's history
            // of epoch adoptions and its distance from the wave's reference position -- tools/mb_steplog.py)
            if constexpr (MB) { ir[7] = (int)(((unsigned)p & 0xffffffu) | (((unsigned)W.epoch & 0xffu) << 24)); }
            if constexpr (MB) { ir[6] = need_full | (int)((mw.hist & 0xfffffu) << 4) | (int)(((unsigned)(p - W.rpos) & 0x7fu) << 24); }
#endif
            dr[0] = x_new; dr[1] = L_new; dr[2] = L_post; dr[3] = T;
        }
    }
    if (__builtin_expect(rec_now && cool, 0)) {
        int sl = 0, ss = -1;
        if (lane == 0) {
            if constexpr (MB) {      // (the record slots of all workgroups' chains: counted in memory)
                sl = (int)atomicAdd(&g_mb->word[MW_NLIK], 1ull);
                if (iter > cs.n_burn) ss = (int)atomicAdd(&g_mb->word[MW_NSMP], 1ull);
            } else {
            sl = atomicAdd(&sh.c.n_lik, 1);
            if (iter > cs.n_burn) ss = atomicAdd(&sh.c.n_smp, 1);
            }
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
    // ---- orders of this chain's coming full evaluations (what role P does for all chains in step_body): the next step's
    // ---- if it needs one (one step ahead), else -- that step being a hypocentre step -- the one after it (two steps ahead:
    // ---- the workers leave the event of the step in between out, this wave adds it).  Positions are predictions; a step
    // ---- uses an order only if it starts exactly where the order was written for, and an epoch change voids the book.
    // (the look-ahead stands if no epoch change came between: then the next step's start and proposal go to the next call)
    const bool la_ok = la == 2 && la_epoch == W.epoch && ok != 0;
    struct { int x, y, z; } d1 = {0, 0, 0};
    if (la_ok) {
        d1.x = uni(d1v.x); d1.y = uni(d1v.y); d1.z = uni(d1v.z);
        nx.p = la_p1; nx.it = iter + 1; nx.c = c; nx.epoch = W.epoch;
        nx.type = d1.x; nx.idx = d1.y; nx.evt = d1.z; nx.dec_w = uni(d1v.w);
        nx.g = la_g; nx.r = la_r; nx.logr = la_logr;
        const int swz = uni(swv.z);
        // (what the swap at la_E2 draws on THIS rank: flow_swap_at's rule)
        if (cs.n_procs * nc_ <= 1 || (rg.lock && cs.rank != 0)) nx.b3 = la_E2;
        else nx.b3 = swz > 0 ? la_E2 + swz + ((!rg.lock || uni(swv.x) / nc_ == 0) ? 1 : 0) : -1;
    }
    if (rg.mir_n > 0 && sh.ob_pos[c] == -1 && iter + 1 <= sh.c.iter_target) {
        const int lim = sh.fill - 8;
        int p1 = -1, mode = 0, pj = -1, jt = 0, ji = 0;
        bool w1 = false, job1 = false;
        if (la_ok) {
            p1 = la_p1; w1 = true;
            job1 = d1.x >= 1 && d1.x <= 4;
            mode = job1 ? 1 : 0; pj = p1; jt = d1.x; ji = d1.y;
            const int d2x = uni(d2v.x);
            if (HTM_ALLOW2 && NCH > 0 && !job1 && iter + 2 <= sh.c.iter_target && d2x >= 1 && d2x <= 4) { mode = 2; pj = la_p2; jt = d2x; ji = uni(d2v.y); }
        } else {
            // this chain's next step is its step of the next iteration (this wave's other chains of this iteration come first)
            p1 = (W.rpos1 >= 0 && c >= W.rc1) ? hop_ahead(rg, W.rpos1, c - W.rc1) : -1;
            w1 = p1 >= 0 && p1 < lim;
            const i32x4 d1r = reinterpret_cast<const i32x4 *>(rg.dec)[(w1 ? p1 : 0) & M];
            d1.x = d1r.x; d1.y = d1r.y; d1.z = d1r.z;
            job1 = w1 && d1.x >= 1 && d1.x <= 4;
            mode = job1 ? 1 : 0; pj = p1; jt = d1.x; ji = d1.y;
            if (HTM_ALLOW2 && NCH > 0 && w1 && !job1 && iter + 2 <= sh.c.iter_target && W.B2 >= 0) {
                const int p2 = hop_ahead(rg, W.B2, c);
                if (p2 < lim) {
                    const i32x4 d2 = reinterpret_cast<const i32x4 *>(rg.dec)[p2 & M];
                    if (d2.x >= 1 && d2.x <= 4) { mode = 2; pj = p2; jt = d2.x; ji = d2.y; }
                }
            }
        }
        if (mode) {
            const int jgoff = jt == 1 ? 0 : jt == 2 ? nc_ : jt == 3 ? nc_ + nc_ * S_ : 2 * nc_ + nc_ * S_;
            const int jo = jgoff + c * ((jt == 1 || jt == 3) ? 1 : S_) + ji;
            const double jx_old = rg.mx[jo];                                  // LDS mirror, kept current by this wave's commits
            const int jop = cs.prior_same ? jo - c * ((jt == 1 || jt == 3) ? 1 : S_) : jo;
            const double jstep = rg.mir_steps ? rg.mstep[jo] : ld_const(&cs.prior[jop].step);
            const double jx_new = jx_old + rg.pg[pj & M] * jstep;             // cls_model.f90:172, as the step will compute it
            if (cs.rayleigh14) {                                              // a Rayleigh prior among vs/qs/corrections (:178-187)
                if (ld_const(&cs.prior[jop].ptype) == 1 && jx_new <= ld_const(&cs.prior[jop].mu)) mode = 0;      // prior rejects: no evaluation
            }
            // two ahead: the workers wait for this step's commit by reading its value back; the step in between must not
            // be able to overwrite that very element before they look
            const int o_mid = off_hy + c * nh + d1.y;
            if (mode == 2 && acc && o == o_mid) mode = 0;
            if (mode) {
                unsigned long long tk = 0;
                if (lane == 0) {
                    tk = (atomicAdd(&sh.c.jobs_total, 1ull) + 1ull) & 0x7fffffffull;
                    if (tk == 0) tk = 0x7fffffffull;
                    // (the step right before the order's: the one in between, or -- one ahead -- this very step)
                    sh.ob_pos[c] = pj; sh.ob_tag[c] = (unsigned)tk; sh.ob_mode[c] = mode;
                    sh.ob_mid[c] = mode == 2 ? (d1.x | (d1.z << 3)) : (type >= 5 ? (type | (evt << 3)) : type);
                }
                const unsigned tag = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)tk);
                if (lane < cs.slot_rep * kGranPerSlot) {
                    const int gi = lane & 7;
                    const unsigned long long xb = (unsigned long long)__double_as_longlong(jx_new);
                    const unsigned long long cb = (unsigned long long)__double_as_longlong(x_new);
                    const unsigned pay = gi == 0 ? (unsigned)launch : gi == 1 ? ((unsigned)jt | ((unsigned)ji << 3))
                                       : gi == 2 ? (unsigned)(xb >> 32) : gi == 3 ? (unsigned)xb
                                       : gi == 4 ? (acc ? (unsigned)o : 0xffffffffu)              // the commit the workers must see
                                       : gi == 5 ? (unsigned)(cb >> 32) : gi == 6 ? (unsigned)cb
                                       : (mode == 2 ? (unsigned)o_mid + 1u                         // element of the step right before the order's (+1;
                                          : type >= 5 ? (unsigned)o + 1u : 0u);                      // 0 = not a hypocentre step): its event is left out
                    st_gran(cs.slots + (size_t)(lane >> 3) * cs.slot_stride + c * kGranPerSlot + gi, tag, pay);
                }
            }
        }
    }
    if (lane == 0) sh.pv_mid[c] = type >= 5 ? (type | (evt << 3)) : type;      // (what the chain's next full evaluation leaves to this wave)
#ifdef HTM_STAMPS
    FSTAMP(5);
    if (lane == 0 && cs.stamps) {
        unsigned long long *a = sh.stamp_acc + 12 * (wave & 7);
        if (need_full) { a[6] += t_last - t_step0; a[8] += 1; a[10] += t_wait; }
        else { for (int k = 0; k < 6; ++k) a[k] += st_acc[k]; a[9] += st_acc[6]; a[7] += 1; }
    }
#endif
    return p + cnt;
}

// block 0 of a k_mcmc<NCH, F32, 0> launch when the host selects the free-running master (htm_hip.hip: flow_ok)
// (MB: one of several master workgroups of the launch -- block b runs chains 8 b .. 8 b + 7; returns true in the workgroup that
// finishes last, which has written the launch's end state and releases the workers)
template <int NCH, bool F32, bool LOCK = false, bool MB = false>
__device__ __forceinline__ bool flow_body(FwRef f_, CsRef cs_, int target_arg, int ring_size, int wmax, unsigned long long launch)
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
    rg.lock = LOCK ? 1 : 0;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NW = min((int)(blockDim.x >> 6), 8);      // chain waves
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
    if constexpr (LOCK) {      // where the peers' inboxes are mapped (flow_post_chain / flow_post_header)
        for (int q = tid; q < cs.n_procs && q < kGathStage; q += blockDim.x) reinterpret_cast<unsigned long long **>(s_gath)[q] = cs.outbox[q];
    }
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
        sh.xdone = sh.c.iter_done; sh.xanch = sh.c.iter_done; sh.xanch_claim = sh.c.iter_done;
        sh.xctl4[0] = sh.xctl4[1] = sh.xctl4[2] = sh.xctl4[3] = 0;
        sh.xctl = 0u; sh.xctl_iter = -1;
    }
    __syncthreads();
    const int i0 = sh.i0;
    for (int c = tid; c < kMaxChains; c += blockDim.x) {
        sh.ob_pos[c] = -1;
        sh.pv_mid[c] = c < nc ? cs.prev_mid[c] : 0;
        sh.prog[c] = (unsigned long long)(unsigned)c;          // key(i0, c), epoch 0, prior ok
        sh.done[c] = c;
        if (c < nc) {
            const double T = cs.temp[c], L = cs.L[c];
            sh.L[c] = L; sh.temp[c] = T; sh.rtemp[c] = 1.0 / T;
            sh.T4[(i0 + 1) & 3][c] = T; sh.T4[i0 & 3][c] = T; sh.L4[i0 & 3][c] = L;
            sh.rT4[(i0 + 1) & 3][c] = 1.0 / T; sh.rT4[i0 & 3][c] = 1.0 / T;
            const double b_ = cs.xall[c], q_ = cs.xall[nc + nc * cs.S + c];      // vs, qs of the chain ([vs | t_corr | qs | ...])
            sh.rbeta[c] = 1.0 / b_; sh.katt[c] = (kPi * kFreq) / (q_ * b_);
        }
    }
    MbShared *g_mb = MB ? cs.mb : nullptr;
    const int mb_b = MB ? (int)blockIdx.x : 0;                // this master workgroup; its chains are 8 b + wave
    if (sh.c.iter_done >= sh.c.iter_target || sh.c.stop || sh.c.err) return !MB || mb_b == 0;      // (uniform; every master block reads the same control block)
    if (sh.avail < 3 * wmax) {                                 // the produced stream does not cover a safe stretch: the host refills
        __syncthreads();
        if (tid == 0 && mb_b == 0) { if (LOCK) sh.c.err = -7; else sh.c.stop = 2; *cs.ctrl = sh.c; }      // (lock-step ranks leave a launch only together: the host feeds the stream first)
        return !MB || mb_b == 0;
    }
    // the draws an iteration can take: 6 per chain step + select_pair's and judge_swap's
    const int wd = 6 * nc + 16;
    const int look = 3 * wd + 24, back = wd + 8;
    prefetch_all(cs, sh, rg, min(look, ring_size - back));       // ends with a barrier
    const int n_int = cs.n_interval;
    int rec_phase = (i0 + 1) % n_int;                            // iteration % n_interval, kept by increments

    FlowWave W;
    W.epoch = 0; W.akey = 0; W.rc = 0; W.rpos = 0;
    W.rc1 = 0; W.rpos1 = flow_next_base<MB>(cs, sh, rg, 0, nc, sh.fill);
    W.B2 = flow_next_base<MB>(cs, sh, rg, W.rpos1, nc, sh.fill);
    FlowNext nx;
    nx.p = -1; nx.it = 0; nx.c = 0; nx.epoch = 0; nx.type = 5; nx.idx = 0; nx.evt = 1; nx.dec_w = 6; nx.g = 0.0; nx.r = 0.0; nx.logr = 0.0; nx.b3 = -1;
    int iter = i0 + 1;
    int c = MB ? 8 * mb_b + wave : wave;
    bool alive = MB ? (wave < 8 && c < nc) : (wave < nc && wave < NW);
    MbWave mw;
#ifdef HTM_MB_DIAG
    mw.hist = 0u;
#endif
    mw.epoch = 0; mw.last_iter = sh.c.iter_target; mw.err = 0; mw.n_lik = sh.c.n_lik; mw.n_smp = sh.c.n_smp; mw.pv = 0ull;
#ifdef HTM_STAMPS
    const unsigned long long t_loop0 = __builtin_amdgcn_s_memtime();
#endif
    while (alive) {
        // ---- top of a step: the epoch its position is predicted in
        FlowTop tp;
        {
            const int co = (HTM_FAIR && (!LOCK || HTM_FAIR_LOCK) && (c ^ 4) < nc) ? (c ^ 4) : c;
            if constexpr (MB) {      // (what the wave saw with its latest look at MbShared: a step old, checked again in the turn)
                tp.epoch = mw.epoch; tp.last_iter = mw.last_iter; tp.err = mw.err != 0 ? mw.err : lds_ld(&sh.c.err);
                tp.pk = (int)(unsigned)rl_u64(mw.pv, co);
            } else {
            tp.epoch = lds_ld(&sh.epoch); tp.last_iter = lds_ld(&sh.last_iter); tp.err = lds_ld(&sh.c.err);
            tp.pk = (int)(unsigned)lds_ld(&sh.prog[co]);
            }
            tp.book_pos = sh.ob_pos[c]; tp.book_mode = sh.ob_mode[c]; tp.book_mid = sh.ob_mid[c]; tp.book_tag = sh.ob_tag[c];
            tp.pv_mid = sh.pv_mid[c];
            tp.L = sh.L[c];
            tp.rbeta = sh.rbeta[c]; tp.katt = sh.katt[c];
            asm volatile("" ::: "memory");
        }
        {
            const int e = tp.epoch;
            if (__builtin_expect(e != W.epoch, 0)) {
                bool stands = false;
                if (!flow_adopt<MB>(cs, sh, rg, W, e, iter, c, false, stands, g_mb)) {
                    if constexpr (MB) { mb_take(mw, mb_look(g_mb, nc, lane)); }      // (the anchor of that epoch is gone or not there yet: look again)
                    continue;
                }
                flow_void_books(cs, sh, MB ? c : wave, MB ? 64 : NW, nc, lane);
                if constexpr (MB) MB_HIST(mw, 0x8 | (e & 3));
                continue;      // (the book read above is void with it: from the top)
            }
        }
        if (iter > tp.last_iter || tp.err != 0) break;      // (a lock-step rank: the swap of its last iteration is applied after the loop)
        if (__builtin_expect(W.rpos1 < 0 || W.B2 < 0, 0)) {      // predictions the window did not cover when they were made
            if (W.rpos1 < 0) { W.rpos1 = flow_next_base<MB>(cs, sh, rg, W.rpos, nc - W.rc, sh.fill); W.rc1 = 0; }
            if (W.B2 < 0 && W.rpos1 >= 0) W.B2 = flow_next_base<MB>(cs, sh, rg, W.rpos1, nc - W.rc1, sh.fill);
        }
        // where the step starts: known from the step before (FlowNext), or from the hop table now
        const bool known = nx.p >= 0 && nx.it == iter && nx.c == c && nx.epoch == W.epoch;
        // (more than kHops steps from the reference position take chained table entries: every entry read must lie inside the
        // window -- a position past it holds another iteration's numbers, and nothing downstream would notice)
        auto hops_in_window = [&](int pos, int n) __attribute__((always_inline)) {
            const int lim = lds_ld(&sh.fill) - 16;
            int q = pos;
            while (n > kHops) { if (q >= lim) return -1; q += rg.hop[(q & rg.mask) * kHops + kHops - 1]; n -= kHops; }
            if (n > 0) { if (q >= lim) return -1; q += rg.hop[(q & rg.mask) * kHops + n - 1]; }
            return q;
        };
        // (up to kHops steps: ONE entry, at a position that was inside the window when it became the wave's reference -- no look at
        // the window's extent on the way to every step)
        int p = known ? nx.p : ((!MB || c - W.rc <= kHops) ? hop_ahead(rg, W.rpos, c - W.rc) : hops_in_window(W.rpos, c - W.rc));
        if (!known) nx.p = -1;
        if (!LOCK && wave == 0 && c == 0 && lane == 0) {      // (a lock-step rank asks the others through its swap record: flow_post_record)
            // chain 0's wave decides where the launch ends: record buffers or produced stream nearly used up.  Everybody
            // learns it before committing a step of this iteration (its turn waits for chain 0's check)
            int code = 0;
            if constexpr (MB) {
                // (the records are counted in memory; this wave's latest look is a step old: a margin of one more iteration)
                if (mw.n_lik + 5 * nc > cs.cap_lik || mw.n_smp + 5 * nc > cs.cap_smp) code = 1;
                else if (sh.avail < p + 3 * wd + 32) code = 2;
                if (code && mw.last_iter > iter) {
                    st_agent(&g_mb->word[MW_STOP], (unsigned long long)(unsigned)code);
                    st_agent(&g_mb->word[MW_LAST], (unsigned long long)(unsigned)iter);
                }
            } else {
            if (sh.c.n_lik + 3 * nc > cs.cap_lik || sh.c.n_smp + 3 * nc > cs.cap_smp) code = 1;
            else if (sh.avail < p + 3 * wd + 32) code = 2;
            if (code && lds_ld(&sh.last_iter) > iter) { sh.stop_code = code; lds_st(&sh.last_iter, iter); }
            }
        }
        if constexpr (MB) {
            // (chain 0's wave, all lanes: its own view of the launch's end at once -- its next look may still return the old word)
            if (wave == 0 && c == 0 && mw.last_iter > iter &&
                (mw.n_lik + 5 * nc > cs.cap_lik || mw.n_smp + 5 * nc > cs.cap_smp || sh.avail < p + 3 * wd + 32)) mw.last_iter = iter;
        }
        // the window covers this step (chain 0's wave keeps it 3 iterations ahead); a fail-stop, never expected to wait
        if (__builtin_expect(!known && (p < 0 || p + 16 >= lds_ld(&sh.fill)), 0)) {      // (a known start was checked against the window when it was looked up)
            if (wave == 0) { if (lane == 0) sh.c.err = -13; break; }
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            bool dead = false;
            while (p < 0 || p + 16 >= lds_ld(&sh.fill)) {
                if (sh.c.err != 0 || __builtin_amdgcn_s_memrealtime() - t0 > 500000000ull) { dead = true; break; }
                __builtin_amdgcn_s_sleep(4);
                if (p < 0) p = hops_in_window(W.rpos, c - W.rc);
            }
            if (dead) { if (lane == 0 && sh.c.err == 0) sh.c.err = -13; break; }
        }
        // Two chain waves share a SIMD, and the hardware issues the older one first: left alone, waves 0-3 finish a step in
        // ~10 k cycles and wait ~2.7 k in their turns for waves 4-7, which need ~12 k.  The wave that is BEHIND its SIMD's
        // other wave (a step or more, by the checks published) asks for priority; the one ahead gives it up.
        // (Not between lock-step ranks: there the hand-over costs 2-3 %, profiles/r03_n_fair.txt.)
        if (HTM_FAIR && (!LOCK || HTM_FAIR_LOCK) && (c ^ 4) < nc) {      // (chain c ^ 4 is the corresponding chain of the SIMD's other wave, whatever the number of chains per wave)
            const int pk = tp.pk;
            if (pk >= (iter - i0) * nc + (c ^ 4)) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
        }
        // (several master workgroups: a wave has one chain -- "wave" c of as many waves as there are chains)
        if constexpr (MB) MB_HIST(mw, (known ? 4 : 0) | (W.epoch & 3));
        const int r = flow_step<NCH, F32, LOCK, MB>(f, cs, sh, rg, W, nx, g_mb, mw, s_gath, wmax, s_sx, s_sy, s_sz, c, p, iter, lane, MB ? c : wave,
                                              MB ? 64 : NW, launch, wave == 0, look, back, rec_phase == 1, tp);
        if (r == kFlowRestart) continue;
        if (r == kFlowAbort || r == kFlowStop) break;
        // ---- this wave's next step
#ifdef HTM_MB_DIAG
        mw.hist = 0u;
#endif
        c += MB ? nc : NW;
        if (c >= nc) {
            c = MB ? 8 * mb_b + wave : wave;
            iter += 1;
            rec_phase = rec_phase + 1 == n_int ? 0 : rec_phase + 1;
            if (W.rpos1 < 0) {    // (the window did not cover the prediction when it was made: it does now)
                W.rpos1 = flow_next_base<MB>(cs, sh, rg, W.rpos, nc - W.rc, 1 << 30); W.rc1 = 0;
                W.B2 = -1;
            }
            W.rc = W.rc1; W.rpos = W.rpos1;
            W.rc1 = 0; W.rpos1 = W.B2 >= 0 ? W.B2 : flow_next_base<MB>(cs, sh, rg, W.rpos, nc - W.rc, sh.fill);
            // (the base of the iteration after next: looked up during the step, or from the tables now)
            if (nx.p >= 0 && nx.epoch == W.epoch && nx.it == iter && nx.b3 >= 0 && W.rpos1 >= 0) W.B2 = nx.b3;
            else W.B2 = flow_next_base<MB>(cs, sh, rg, W.rpos1, nc, sh.fill);
        }
    }
#ifdef HTM_STAMPS
    if (lane == 0 && cs.stamps && wave < 8) sh.stamp_acc[12 * wave + 11] += __builtin_amdgcn_s_memtime() - t_loop0;
#endif
    __syncthreads();
#ifdef HTM_STAMPS
    if (cs.stamps)
        for (int k = tid; k < 96; k += blockDim.x) if (sh.stamp_acc[k]) atomicAdd(&cs.stamps[32 + k], sh.stamp_acc[k]);      // (the workers' own stamps sit at 20..28)
#endif
    // ---- every step up to last_iter is committed: the swap of the last iteration, counters, the launch's end state
    if constexpr (LOCK) {
        if (wave == 0 && sh.c.err == 0) flow_lock_finish(cs, sh, lane);
        __syncthreads();
    }
    if constexpr (MB) {
        // Several master workgroups: each writes its own chains' end state and adds its counters; the one that finishes LAST
        // (a ticket) applies the swap of the last iteration -- from the records in MbShared -- and writes the control block.
        const int nb = (nc + 7) / 8;
        if (tid == 0) {
            if (sh.c.err != 0) st_agent(&g_mb->word[MW_ERR], (unsigned long long)(unsigned)sh.c.err);      // (the other workgroups leave at their next look)
            sh.last_iter = (int)(unsigned)ld_agent(&g_mb->word[MW_LAST]);
        }
        __syncthreads();
        const int last = min(sh.last_iter, sh.c.iter_target);
        if (tid < 8 && 8 * mb_b + tid < nc && last > i0) {
            const int k = 8 * mb_b + tid;
            cs.L[k] = sh.L[k]; cs.temp[k] = sh.T4[last & 3][k]; cs.prev_mid[k] = sh.pv_mid[k];
        }
        for (int k = tid; k < 7 * nc; k += blockDim.x) {          // flush this launch's counters (this workgroup's chains)
            if (sh.np[k]) atomicAdd(&cs.n_propose[k], sh.np[k]);
            if (sh.na[k]) atomicAdd(&cs.n_accept[k], sh.na[k]);
        }
        if (tid == 0) {
            atomicAdd(&g_mb->n_full, sh.n_full_w); atomicAdd(&g_mb->n_part, sh.n_part_w);
            atomicMax(&g_mb->word[7], sh.c.jobs_total);
        }
        __threadfence();
        __syncthreads();
        if (tid == 0) sh.xdone = atomicAdd(&g_mb->word[MW_FIN], 1ull) == (unsigned long long)(nb - 1) ? 1 : 0;
        __syncthreads();
        if (sh.xdone == 0) return false;      // (uniform)
        __threadfence();
        if (tid == 0) {
            const int gerr = (int)(unsigned)ld_agent(&g_mb->word[MW_ERR]);
            if (sh.c.err == 0 && gerr != 0) sh.c.err = gerr;
            if (sh.c.err == 0 && last > i0) {
                const int par = last & 3;
                unsigned sw[8];
                for (int k = 0; k < 8; ++k) { const unsigned long long v = ld_agent(&g_mb->swrec[par][k]); sw[k] = (unsigned)v; if (((unsigned)(v >> 32) & 0xffffffu) != ((unsigned)last & 0xffffffu)) sh.c.err = -20; }
                const int E = (int)sw[0], i1 = (int)sw[1], i2 = (int)sw[2], nd = (int)sw[3];
                if (sh.c.err == 0 && nc > 1) {
                    const double sr = __longlong_as_double((long long)(((unsigned long long)sw[4] << 32) | sw[5]));
                    const double slr = __longlong_as_double((long long)(((unsigned long long)sw[6] << 32) | sw[7]));
                    double TL[2][2];
                    for (int k = 0; k < 2; ++k) {
                        const unsigned long long *r = &g_mb->crec[par][k == 0 ? i1 : i2][0];
                        const unsigned long long a0 = ld_agent(r), a1 = ld_agent(r + 1), a2 = ld_agent(r + 2), a3 = ld_agent(r + 3);
                        if ((int)(unsigned)(a0 >> 32) != last || (int)(unsigned)(a3 >> 32) != last) sh.c.err = -21;
                        TL[k][0] = gran_f64(a0, a1); TL[k][1] = gran_f64(a2, a3);
                    }
                    const double del_s = (TL[1][1] - TL[0][1]) * (1.0 / TL[0][0] - 1.0 / TL[1][0]);      // cls_parallel.f90:292
                    if (sr >= kEps && slr <= del_s) { cs.temp[i1] = TL[1][0]; cs.temp[i2] = TL[0][0]; }   // :131-136
                    sh.c.swap_i1 = i1; sh.c.swap_i2 = i2; sh.c.swap_r = sr; sh.c.swap_logr = slr;
                }
                if (sh.c.err == 0) {
                    sh.c.spos = sh.origin + E + nd;
                    sh.c.iter_done = last;
                    sh.c.stage = ST_IDLE;
                    if (sh.c.slog_cap > 0) sh.c.slog_n = min(sh.c.slog_cap, sh.c.slog_n + (last - i0) * nc);
                    sh.c.n_full_evals += (long long)ld_agent(&g_mb->n_full);
                    sh.c.n_partial_evals += (long long)ld_agent(&g_mb->n_part);
                    sh.c.n_lik = (int)(unsigned)ld_agent(&g_mb->word[MW_NLIK]); sh.c.n_smp = (int)(unsigned)ld_agent(&g_mb->word[MW_NSMP]);
                    sh.c.jobs_total = ld_agent(&g_mb->word[7]);
                    if (last < sh.c.iter_target) sh.c.stop = (int)(unsigned)ld_agent(&g_mb->word[MW_STOP]);
                }
            }
            *cs.ctrl = sh.c;
        }
        __syncthreads();
        return true;
    }
    if (tid == 0 && sh.c.err == 0) {
        if constexpr (LOCK) {
            // (iteration counter, stream position and temperatures were settled with the last swap: flow_lock_finish)
            const int last = sh.c.iter_done;
            for (int k = 0; k < nc; ++k) { cs.L[k] = sh.L[k]; cs.temp[k] = sh.temp[k]; }
            if (sh.c.slog_cap > 0) sh.c.slog_n = min(sh.c.slog_cap, sh.c.slog_n + (last - i0) * nc);
            sh.c.n_full_evals += (long long)sh.n_full_w;
            sh.c.n_partial_evals += (long long)sh.n_part_w;
        } else {
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
    }
    __syncthreads();
    for (int k = tid; k < 7 * nc; k += blockDim.x) {          // flush this launch's counters
        if (sh.np[k]) atomicAdd(&cs.n_propose[k], sh.np[k]);
        if (sh.na[k]) atomicAdd(&cs.n_accept[k], sh.na[k]);
    }
    for (int k = tid; k < nc; k += blockDim.x) cs.prev_mid[k] = sh.pv_mid[k];
    if (tid == 0) *cs.ctrl = sh.c;
    return true;
}

// before a launch with several master workgroups (one wave): MbShared as the first step finds it
__global__ __launch_bounds__(64) void k_mb_init(ChainsDev cs, int target_arg)
{
    MbShared *g = cs.mb;
    const int lane = threadIdx.x;
    unsigned long long *w = reinterpret_cast<unsigned long long *>(g);
    for (int k = lane; k < (int)(sizeof(MbShared) / sizeof(unsigned long long)); k += 64) w[k] = 0ull;
    __syncthreads();
    for (int c = lane; c < kMaxChains; c += 64) g->prog[c] = (unsigned long long)(unsigned)c;          // key(i0, c), epoch 0, prior ok
    if (lane == 0) {
        const Ctrl c = *cs.ctrl;
        g->word[MW_LAST] = (unsigned long long)(unsigned)(target_arg >= 0 ? target_arg : c.iter_target);
        g->word[MW_NLIK] = (unsigned long long)(unsigned)c.n_lik; g->word[MW_NSMP] = (unsigned long long)(unsigned)c.n_smp;
        g->word[7] = c.jobs_total;
    }
}

}  // namespace htm
