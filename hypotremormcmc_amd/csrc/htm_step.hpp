// htm_step.hpp -- k_step: one workgroup drives all chains of a rank through MCMC iterations.
//
// Roles inside the workgroup
//   chain waves 0..NW-1   wave <-> chain (chains c = wave, wave+NW, ...), lane <-> station.  Per iteration a
//                         chain wave finds its position in the rank's random stream, builds the proposal
//                         (cls_mcmc.f90:134-165, cls_model.f90:162-190), evaluates the single-event partial
//                         update (cls_forward.f90:307-362) and takes the Metropolis decision
//                         (cls_mcmc.f90:193-203) without any workgroup barrier in between.
//   producer wave NW      keeps an LDS ring of the rank's xorshift128 stream filled AHEAD of the chains:
//                         raw draws (serial, scalar ALU) and, per stream position, everything a consumer may
//                         need from it: U = rand_u, log U, the Box-Muller value starting there, and the number
//                         of draws a chain step starting there consumes.  None of this depends on chain state,
//                         so it runs concurrently with the chain waves and is off the critical path.
//   wave 0, lanes <-> chains  validates the optimistic stream positions (a Rayleigh-prior rejection makes a
//                         step one draw shorter, cls_mcmc.f90:193), plans the temperature swap
//                         (cls_parallel.f90:226-230,:292-299), assigns record slots and builds the list of
//                         chains that need k_full -- all lane-parallel (ballot / DPP scan), no serial loops.
// Two workgroup barriers per iteration.  The kernel loops over iterations until a chain needs a full
// evaluation (hand-over to k_full, resume in the next launch), the target is reached or a buffer is full.
#pragma once
#include "htm_device.hpp"

namespace htm {

struct StepShared {
    Proposal prop[kMaxChains];
    double temp[kMaxChains], L[kMaxChains];
    int start[kMaxChains];        // stream position at which each chain step starts (optimistic until validated)
    int cnt[kMaxChains];          // draws the step consumed (judge draw included iff prior_ok)
    int slot_l[kMaxChains], slot_s[kMaxChains];
    uint32_t init_state[4];       // generator state at stream position 0 of this launch
    uint32_t gen[4];              // producer's generator state, after raw position fill_raw-1
    int fill_raw, fill_tr;        // raws valid for positions < fill_raw, transforms for positions < fill_tr
    int base;                     // stream position at which the current iteration starts
    int redo;                     // >= 0: chains >= redo repeat their pass with corrected positions
    int end_pos;
    Ctrl c;
};

struct Ring {                     // LDS ring over stream positions, index = position & mask
    uint32_t *raw;
    int *next;
    double *U, *LOGU, *G;
    int mask;
};

__device__ __forceinline__ ModelDev pick_model(const ChainsDev &cs, int type)
{
    ModelDev m;
    const bool v = type == 1, t = type == 2, q = type == 3, a = type == 4;
    m.x = v ? cs.vs.x : t ? cs.tc.x : q ? cs.qs.x : a ? cs.ac.x : cs.hypo.x;
    m.mu = v ? cs.vs.mu : t ? cs.tc.mu : q ? cs.qs.mu : a ? cs.ac.mu : cs.hypo.mu;
    m.sigma = v ? cs.vs.sigma : t ? cs.tc.sigma : q ? cs.qs.sigma : a ? cs.ac.sigma : cs.hypo.sigma;
    m.step = v ? cs.vs.step : t ? cs.tc.step : q ? cs.qs.step : a ? cs.ac.step : cs.hypo.step;
    m.ptype = v ? cs.vs.ptype : t ? cs.tc.ptype : q ? cs.qs.ptype : a ? cs.ac.ptype : cs.hypo.ptype;
    m.nx = (v || q) ? 1 : (t || a) ? cs.S : cs.hypo.nx;
    return m;
}

// draws a chain step starting with a_select = a consumes when its prior is fine (judge draw included)
__device__ __forceinline__ int draws_if_ok(const ChainsDev &cs, double a)
{
    if (a < cs.th1) return 4;     // vs:      a, g(2), r
    if (a < cs.th2) return 5;     // t_corr:  a, id, g(2), r
    if (a < cs.th3) return 4;     // qs
    if (a < cs.th4) return 5;     // a_corr
    return 6;                     // hypo:    a, id, icmp, g(2), r
}

// word k of the sequence [state at position 0 (x,y,z,w), output 0, output 1, ...]
__device__ __forceinline__ uint32_t stream_word(const StepShared &sh, const Ring &rg, int k)
{
    return k < 4 ? sh.init_state[k] : rg.raw[(k - 4) & rg.mask];
}
// mod_random state after n draws
__device__ __forceinline__ void state_at(const StepShared &sh, const Ring &rg, int n, uint32_t (&out)[4])
{
#pragma unroll
    for (int j = 0; j < 4; ++j) out[j] = stream_word(sh, rg, n + j);
}

__device__ __forceinline__ void ring_transform(const ChainsDev &cs, const Ring &rg, int p)
{
    const uint32_t r0 = rg.raw[p & rg.mask], r1 = rg.raw[(p + 1) & rg.mask];
    const double u = u_of(r0);
    rg.U[p & rg.mask] = u;
    rg.LOGU[p & rg.mask] = log(u);
    rg.G[p & rg.mask] = g_of(r0, r1);
    rg.next[p & rg.mask] = draws_if_ok(cs, u);
}

// producer wave (all 64 lanes): fill the ring until transforms cover positions < target
__device__ __forceinline__ void produce_until(const ChainsDev &cs, StepShared &sh, const Ring &rg, int target,
                                              int lane)
{
    int fr = __builtin_amdgcn_readfirstlane(sh.fill_raw);
    int ft = __builtin_amdgcn_readfirstlane(sh.fill_tr);
    const int base = __builtin_amdgcn_readfirstlane(sh.base);
    if (ft >= target) return;
    uint32_t x = __builtin_amdgcn_readfirstlane(sh.gen[0]), y = __builtin_amdgcn_readfirstlane(sh.gen[1]);
    uint32_t z = __builtin_amdgcn_readfirstlane(sh.gen[2]), w = __builtin_amdgcn_readfirstlane(sh.gen[3]);
    while (ft < target && fr + 64 - (base - 4) <= rg.mask + 1) {
        uint32_t mine = 0;
#pragma unroll 8
        for (int k = 0; k < 64; ++k) {          // serial recurrence on the scalar ALU, mod_random.f90:63-71
            const uint32_t r = xs128_next(x, y, z, w);
            mine = (lane == k) ? r : mine;
        }
        rg.raw[(fr + lane) & rg.mask] = mine;
        fr += 64;
        const int p = ft + lane;                 // fr-1 is the last raw; position p needs raw p+1
        if (p < fr - 1) ring_transform(cs, rg, p);
        ft = fr - 1;
    }
    if (lane == 0) {
        sh.gen[0] = x; sh.gen[1] = y; sh.gen[2] = z; sh.gen[3] = w;
        sh.fill_raw = fr; sh.fill_tr = ft;
    }
}

// one lane, producer idle: make position p available (rare: long select_pair redraw runs)
__device__ inline bool ring_ensure_serial(const ChainsDev &cs, StepShared &sh, const Ring &rg, int p)
{
    while (p >= sh.fill_tr) {
        if (sh.fill_raw + 1 - (sh.base - 4) > rg.mask + 1) return false;
        uint32_t x = sh.gen[0], y = sh.gen[1], z = sh.gen[2], w = sh.gen[3];
        rg.raw[sh.fill_raw & rg.mask] = xs128_next(x, y, z, w);
        sh.gen[0] = x; sh.gen[1] = y; sh.gen[2] = z; sh.gen[3] = w;
        sh.fill_raw += 1;
        if (sh.fill_raw >= 2) { ring_transform(cs, rg, sh.fill_raw - 2); sh.fill_tr = sh.fill_raw - 1; }
    }
    return true;
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

__device__ __forceinline__ bool metropolis(double L_new, double L_cur, double T, double lpr, double r,
                                           double logr)
{
    double ratio = (L_new - L_cur) / T;     // cls_mcmc.f90:194-195
    ratio = ratio + lpr;
    return r >= kEps && logr <= ratio;      // :198-199
}

// One chain step up to (and including, for hypocentre proposals) the Metropolis decision.
// All 64 lanes execute with identical (uniform) values; lane <-> station only inside event_misfit.
// Returns the stream position after this step.
template <int NCH>
__device__ __forceinline__ int chain_pass(const FwdDev &f, const ChainsDev &cs, StepShared &sh, const Ring &rg,
                                          const double *s_sx, const double *s_sy, const double *s_sz, int c,
                                          int p, int iter, int lane)
{
    const int M = rg.mask;
    const double a = rg.U[p & M];
    int type, idx, evt = -999, gpos;
    if (a < cs.th1) { type = 1; idx = 0; gpos = p + 1; }
    else if (a < cs.th2) { type = 2; idx = (int)(rg.U[(p + 1) & M] * cs.S); gpos = p + 2; }
    else if (a < cs.th3) { type = 3; idx = 0; gpos = p + 1; }
    else if (a < cs.th4) { type = 4; idx = (int)(rg.U[(p + 1) & M] * cs.S); gpos = p + 2; }
    else {
        const int id = (int)(rg.U[(p + 1) & M] * cs.E) + 1;
        const int icmp = (int)(rg.U[(p + 2) & M] * 3);
        idx = 3 * id - icmp - 1; type = 5 + icmp; evt = id; gpos = p + 3;
    }
    const bool partial = evt > 0 && iter > 1;       // hypo_tremor_mcmc.f90:246
    const ModelDev Mo = pick_model(cs, type);
    const size_t o = (size_t)c * Mo.nx + idx;
    // every global load of the step is issued here, before any dependent arithmetic
    const double x_old = Mo.x[o], mu = Mo.mu[o], sigma = Mo.sigma[o], step = Mo.step[o];
    const int ptype = Mo.ptype[o];
    double hx = 0.0, hy = 0.0, hz = 0.0, beta = 1.0, q = 1.0;
    const int ev = partial ? evt - 1 : 0;
    const double *tc = cs.tc.x + (size_t)c * cs.S, *ac = cs.ac.x + (size_t)c * cs.S;
    StaRegs<(NCH > 0 ? NCH : 1)> st;
    if (partial) {
        const double *hyp = cs.hypo.x + (size_t)c * cs.hypo.nx + 3 * ev;
        hx = hyp[0]; hy = hyp[1]; hz = hyp[2];
        beta = cs.vs.x[c]; q = cs.qs.x[c];
        if constexpr (NCH > 0) load_sta_regs<NCH>(st, f.S, lane, s_sx, s_sy, s_sz, tc, ac, 0, -1, 0.0);
    }
    const double T = sh.temp[c], L_cur = sh.L[c];

    const double x_new = x_old + rg.G[gpos & M] * step;        // cls_model.f90:172
    const double da = x_new - mu, db = x_old - mu;
    double lpr = -(da * da - db * db) / (2.0 * sigma * sigma);  // :175-177
    int ok = 1;
    if (ptype == 1) {                                           // :178-187
        if (x_new <= mu) { lpr = (double)-1.0e+30f; ok = 0; }
        else lpr = lpr + log(x_new - mu) - log(x_old - mu);
    }
    const int jpos = gpos + 2;
    const double r = ok ? rg.U[jpos & M] : 0.0, logr = ok ? rg.LOGU[jpos & M] : 0.0;
    const int cnt = (jpos - p) + ok;

    double L_new = 0.0;
    int need_full = 0, acc = 0;
    if (ok) {
        if (partial) {
            const int cmp = idx - 3 * ev;        // 0 x, 1 y, 2 z of event ev
            const double px[2] = {hx, cmp == 0 ? x_new : hx};
            const double py[2] = {hy, cmp == 1 ? x_new : hy};
            const double pz[2] = {hz, cmp == 2 ? x_new : hz};
            double out[2];
            if constexpr (NCH > 0) event_misfit<NCH, 2>(f, ev, lane, st, px, py, pz, beta, q, out);
            else event_misfit_generic<2>(f, ev, lane, s_sx, s_sy, s_sz, tc, ac, 0, -1, 0.0, px, py, pz, beta, q, out);
#ifdef HTM_STAMPS
            if (cs.stamps && c == 0 && iter == 4 && lane == 0) {
                double *dbg = reinterpret_cast<double *>(cs.stamps) + 16;
                dbg[0] = px[0]; dbg[1] = px[1]; dbg[2] = py[0]; dbg[3] = py[1]; dbg[4] = pz[0]; dbg[5] = pz[1];
                dbg[6] = out[0]; dbg[7] = out[1]; dbg[8] = x_old; dbg[9] = x_new; dbg[10] = (double)cmp; dbg[11] = (double)ev;
                dbg[12] = beta; dbg[13] = q; dbg[14] = (double)idx; dbg[15] = (double)type;
            }
#endif
            L_new = L_cur + wave_sum1(out[0] - out[1]);
            acc = metropolis(L_new, L_cur, T, lpr, r, logr) ? 1 : 0;
        } else {
            need_full = 1;
        }
    }
    if (lane == 0) {
        Proposal &pr = sh.prop[c];
        pr.type = type; pr.idx = idx; pr.evt = evt; pr.prior_ok = ok; pr.need_full = need_full; pr.accepted = acc;
        pr.cool = (T < 1.0 + kEps) ? 1 : 0; pr.pad_ = 0;
        pr.x_new = x_new; pr.lpr = lpr; pr.r_judge = r; pr.logr_judge = logr; pr.L_new = L_new;
        sh.start[c] = p; sh.cnt[c] = cnt;
    }
    return p + cnt;
}

// wave 0, lanes <-> chains: record slots (hypo_tremor_mcmc.f90:270-280), step log, swap or record export.
// Needs every chain's decision in sh.prop.  Runs between two barriers; the producer is idle meanwhile.
__device__ __forceinline__ void finish_iteration(const ChainsDev &cs, StepShared &sh, int iter, bool lockstep,
                                                 int lane)
{
    const int nc = cs.n_chains, n_all = cs.n_procs * nc;
    const bool in = lane < nc;
    const int c = in ? lane : 0;
    const Proposal pr = sh.prop[c];
    const double T = sh.temp[c];
    const double L_post = pr.accepted ? pr.L_new : sh.L[c];
    const bool cool = T < 1.0 + kEps;
    const bool rec_l = in && cool && (iter % cs.n_interval) == 1;
    const bool rec_s = rec_l && iter > cs.n_burn;
    const unsigned long long ml = __ballot(rec_l), ms = __ballot(rec_s);
    const unsigned long long below = (1ull << lane) - 1ull;
    if (in) {
        sh.slot_l[c] = rec_l ? sh.c.n_lik + __popcll(ml & below) : -1;
        sh.slot_s[c] = rec_s ? sh.c.n_smp + __popcll(ms & below) : -1;
        const int row = sh.c.slog_n + c;
        if (row < sh.c.slog_cap) {
            int32_t *ir = cs.slog_i + 8 * (size_t)row;
            double *dr = cs.slog_d + 4 * (size_t)row;
            ir[0] = iter; ir[1] = c; ir[2] = pr.type; ir[3] = pr.idx + 1; ir[4] = pr.prior_ok;
            ir[5] = pr.accepted; ir[6] = pr.need_full; ir[7] = 0;
            dr[0] = pr.x_new; dr[1] = pr.L_new; dr[2] = L_post; dr[3] = T;
        }
        if (lockstep) { cs.swap_rec[4 + 2 * c] = T; cs.swap_rec[5 + 2 * c] = L_post; }
    }
    if (lane == 0) {
        sh.c.n_lik += __popcll(ml); sh.c.n_smp += __popcll(ms);
        if (sh.c.slog_n < sh.c.slog_cap) sh.c.slog_n += nc;
        if (lockstep) {      // the swap itself happens in MODE_APPLY from the all-gathered records
            cs.swap_rec[0] = (double)sh.c.swap_i1; cs.swap_rec[1] = (double)sh.c.swap_i2;
            cs.swap_rec[2] = sh.c.swap_r; cs.swap_rec[3] = (double)iter;
            sh.c.stage = ST_WAIT_SWAP;
        } else {
            if (n_all > 1) {  // both chains live on this rank: cls_parallel.f90:121-136 + :285-302
                const int c1 = sh.c.swap_i1, c2 = sh.c.swap_i2;
                const double T1 = sh.temp[c1], T2 = sh.temp[c2];
                const double L1 = sh.prop[c1].accepted ? sh.prop[c1].L_new : sh.L[c1];
                const double L2 = sh.prop[c2].accepted ? sh.prop[c2].L_new : sh.L[c2];
                const double del_s = (L2 - L1) * (1.0 / T1 - 1.0 / T2);
                if (sh.c.swap_r >= kEps && sh.c.swap_logr <= del_s) {
                    sh.temp[c1] = T2; sh.temp[c2] = T1;
                    cs.temp[c1] = T2; cs.temp[c2] = T1;
                }
            }
            sh.c.iter_done = iter;
            sh.c.stage = ST_IDLE;
        }
    }
}

// chain wave: apply the decision (cls_mcmc.f90:186-189,:207-219) and write records.  The counters use the
// temperature test of judge time (prop.cool), not the possibly just-swapped sh.temp.
__device__ __forceinline__ void commit_chain(const ChainsDev &cs, StepShared &sh, int c, int iter, int lane)
{
    const Proposal pr = sh.prop[c];
    const bool cool = pr.cool != 0;
    if (lane == 0) {
        if (cool) atomicAdd(&cs.n_propose[c * 7 + pr.type - 1], 1);
        if (pr.accepted) {
            const ModelDev Mo = pick_model(cs, pr.type);
            Mo.x[(size_t)c * Mo.nx + pr.idx] = pr.x_new;
            sh.L[c] = pr.L_new;
            cs.L[c] = pr.L_new;
            if (cool) atomicAdd(&cs.n_accept[c * 7 + pr.type - 1], 1);
        }
    }
    const int sl = sh.slot_l[c], ss = sh.slot_s[c];
    if (sl >= 0 && lane == 0) {
        cs.lik_iter[sl] = iter; cs.lik_chain[sl] = c;
        cs.lik_val[sl] = pr.accepted ? pr.L_new : sh.L[c];
    }
    if (ss >= 0) {
        const int nh = cs.hypo.nx, S = cs.S, rec = nh + 2 * S + 2;
        double *dst = cs.smp_data + (size_t)ss * rec;
        const double *hx = cs.hypo.x + (size_t)c * nh;
        const bool acc = pr.accepted != 0;
        // the element accepted in this very step is taken from the proposal, not re-read from memory
        for (int k = lane; k < nh; k += 64) dst[k] = (acc && pr.type >= 5 && k == pr.idx) ? pr.x_new : hx[k];
        for (int k = lane; k < S; k += 64) {
            const double t = cs.tc.x[(size_t)c * S + k], a = cs.ac.x[(size_t)c * S + k];
            dst[nh + k] = (acc && pr.type == 2 && k == pr.idx) ? pr.x_new : t;
            dst[nh + S + k] = (acc && pr.type == 4 && k == pr.idx) ? pr.x_new : a;
        }
        if (lane == 0) {
            dst[nh + 2 * S] = (acc && pr.type == 1) ? pr.x_new : cs.vs.x[c];
            dst[nh + 2 * S + 1] = (acc && pr.type == 3) ? pr.x_new : cs.qs.x[c];
            cs.smp_iter[ss] = iter; cs.smp_chain[ss] = c;
        }
    }
}

#ifdef HTM_STAMPS
#define STAMP(k)                                                                                   \
    do {                                                                                           \
        if (tid == 0 && cs.stamps) {                                                               \
            const unsigned long long now_ = __builtin_amdgcn_s_memtime();                          \
            cs.stamps[k] += now_ - stamp_last_;                                                    \
            stamp_last_ = now_;                                                                    \
        }                                                                                          \
    } while (0)
#else
#define STAMP(k) do { } while (0)
#endif

template <int NCH>
__global__ __launch_bounds__(576) void k_step(FwdDev f, ChainsDev cs, int mode, int target_arg,
                                               const double *gathered, int ring_size, int wmax)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    StepShared &sh = *reinterpret_cast<StepShared *>(smem);
    char *carve = smem + ((sizeof(StepShared) + 15) & ~size_t(15));
    Ring rg;
    rg.mask = ring_size - 1;
    rg.U = reinterpret_cast<double *>(carve);          carve += sizeof(double) * ring_size;
    rg.LOGU = reinterpret_cast<double *>(carve);       carve += sizeof(double) * ring_size;
    rg.G = reinterpret_cast<double *>(carve);          carve += sizeof(double) * ring_size;
    rg.raw = reinterpret_cast<uint32_t *>(carve);      carve += sizeof(uint32_t) * ring_size;
    rg.next = reinterpret_cast<int *>(carve);          carve += sizeof(int) * ring_size;
    double *s_sx = reinterpret_cast<double *>(carve);
    double *s_sy = s_sx + f.S;
    double *s_sz = s_sy + f.S;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NW = (blockDim.x >> 6) - 1;          // chain waves; wave NW is the producer
    const bool producer = wave == NW;
    const int nc = cs.n_chains;
    const bool lockstep = (mode != MODE_RUN);
    const int n_all = cs.n_procs * nc;
    const int RW = 4 + 2 * nc;                     // swap-record words
#ifdef HTM_STAMPS
    unsigned long long stamp_last_ = __builtin_amdgcn_s_memtime();
#endif

    if (tid == 0) {
        sh.c = *cs.ctrl;
        if (target_arg >= 0) sh.c.iter_target = target_arg;
        for (int k = 0; k < 4; ++k) { sh.init_state[k] = sh.c.rng[k]; sh.gen[k] = sh.c.rng[k]; }
        sh.fill_raw = 0; sh.fill_tr = 0; sh.base = 0; sh.redo = -1; sh.end_pos = 0;
    }
    for (int j = tid; j < f.S; j += blockDim.x) { s_sx[j] = f.sx[j]; s_sy[j] = f.sy[j]; s_sz[j] = f.sz[j]; }
    for (int c = tid; c < nc; c += blockDim.x) { sh.temp[c] = cs.temp[c]; sh.L[c] = cs.L[c]; }
    __syncthreads();

    // ---------------- MODE_APPLY: cls_parallel.f90:118-213 from the all-gathered records -------------
    if (mode == MODE_APPLY) {
        if (tid == 0 && sh.c.stage == ST_WAIT_SWAP && sh.c.err == 0) {
            const int iter = sh.c.iter_done + 1;
            if (n_all > 1) {
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
                    if (cs.rank == rank1) cs.temp[chain1] = T2;
                    if (cs.rank == rank2) cs.temp[chain2] = T1;
                }
                if (cs.rank == rank1)
                    for (int k = 0; k < 4; ++k) sh.c.rng[k] = sh.c.rng_plus1[k];
            }
            sh.c.iter_done = iter;
            sh.c.stage = ST_IDLE;
            *cs.ctrl = sh.c;
        }
        return;
    }

    bool resume = (sh.c.stage == ST_WAIT_FULL);
    if (mode == MODE_FINISH && !resume) return;
    if (sh.c.stage == ST_WAIT_SWAP) return;                   // nothing to do until the swap is applied
    if (!resume && (sh.c.iter_done >= sh.c.iter_target || sh.c.stop || sh.c.err)) return;
    STAMP(0);   // prologue

    // ---------------- P0: first ring fill || judge of the chains that came back from k_full ---------
    if (producer) {
        produce_until(cs, sh, rg, 2 * wmax, lane);
    } else if (resume) {
        for (int c = wave; c < nc; c += NW) {
            Proposal pr = cs.prop[c];
            if (pr.need_full) {
                double acc = 0.0;
                for (int k = lane; k < cs.n_wg; k += 64) acc += cs.partial[(size_t)c * cs.n_wg + k];
                pr.L_new = -wave_sum1(acc) - f.const_sum;      // cls_forward.f90:277-300
                pr.accepted = metropolis(pr.L_new, sh.L[c], sh.temp[c], pr.lpr, pr.r_judge, pr.logr_judge) ? 1 : 0;
            }
            pr.cool = (sh.temp[c] < 1.0 + kEps) ? 1 : 0;
            if (lane == 0) sh.prop[c] = pr;
        }
    }
    __syncthreads();
    STAMP(1);   // P0

    for (;;) {
        const int iter = sh.c.iter_done + 1;
        if (!resume) {
            // ---------------- anything left to do? ---------------------------------------------------
            if (sh.c.iter_done >= sh.c.iter_target || sh.c.stop || sh.c.err) break;
            if (sh.c.n_lik + nc > cs.cap_lik || sh.c.n_smp + nc > cs.cap_smp) {
                __syncthreads();
                if (tid == 0) { if (lockstep) sh.c.err = -5; else sh.c.stop = 1; }
                __syncthreads();
                break;
            }
            // ---------------- passes: propose -> partial update -> decision, per chain wave ----------
            int redo = 0;
            bool first = true;
            for (;;) {
                if (producer) {
                    if (first) produce_until(cs, sh, rg, sh.base + 2 * wmax, lane);
                } else {
                    int p = 0;
                    bool have_p = false;
                    for (int c = wave; c < nc; c += NW) {
                        if (c < redo) continue;
                        if (!first) p = sh.start[c];                        // corrected by the validation
                        else if (!have_p) {                                 // chase: skip the chains before
                            p = sh.base;
                            for (int k = 0; k < c; ++k) p += rg.next[p & rg.mask];
                        }
                        p = chain_pass<NCH>(f, cs, sh, rg, s_sx, s_sy, s_sz, c, p, iter, lane);
                        have_p = true;
                        for (int k = 1; k < NW && c + k < nc; ++k) p += rg.next[p & rg.mask];
                    }
                }
                __syncthreads();                                            // ---- barrier A
                STAMP(2);   // passes
                if (wave == 0) {
                    // ---------------- validation: exact start = base + exclusive scan of the draw counts
                    const bool in = lane < nc;
                    const int my_cnt = in ? sh.cnt[lane] : 0, my_start = in ? sh.start[lane] : 0;
                    const int incl = wave_incl_scan(my_cnt);
                    const int exact = sh.base + incl - my_cnt;
                    const unsigned long long bad = __ballot(in && exact != my_start);
                    if (bad) {
                        if (in) sh.start[lane] = exact;
                        if (lane == 0) sh.redo = __ffsll((long long)bad) - 1;
                    } else {
                        const int total = __builtin_amdgcn_readlane(incl, 63);
                        const unsigned long long mf = __ballot(in && sh.prop[lane].need_full != 0);
                        const unsigned long long mp = __ballot(in && sh.prop[lane].prior_ok != 0 && sh.prop[lane].need_full == 0);
                        if (in && ((mf >> lane) & 1ull)) cs.full_list[__popcll(mf & ((1ull << lane) - 1ull))] = lane;
                        if (lane == 0) {
                            // ---------------- swap plan + RNG commit (cls_parallel.f90:226-230,:294) ----
                            int pos = sh.base + total;
                            sh.c.swap_i1 = -1; sh.c.swap_i2 = -1; sh.c.swap_r = 0.0; sh.c.swap_logr = 0.0;
                            if (n_all > 1) {
                                if (cs.rank == 0) {
                                    if (!ring_ensure_serial(cs, sh, rg, pos + 2)) sh.c.err = -4;
                                    const int i1 = (int)(rg.U[pos & rg.mask] * cs.n_procs * nc);
                                    pos++;
                                    int i2;
                                    for (;;) {
                                        if (!ring_ensure_serial(cs, sh, rg, pos + 2)) { sh.c.err = -4; i2 = (i1 + 1) % n_all; break; }
                                        i2 = (int)(rg.U[pos & rg.mask] * cs.n_procs * nc);
                                        pos++;
                                        if (i1 != i2) break;
                                    }
                                    sh.c.swap_i1 = i1; sh.c.swap_i2 = i2;
                                }
                                if (!ring_ensure_serial(cs, sh, rg, pos + 2)) sh.c.err = -4;
                                sh.c.swap_r = rg.U[pos & rg.mask]; sh.c.swap_logr = rg.LOGU[pos & rg.mask];
                                uint32_t st1[4];
                                state_at(sh, rg, pos + 1, st1);
                                for (int k = 0; k < 4; ++k) sh.c.rng_plus1[k] = st1[k];
                                if (!lockstep) pos++;            // single rank: this rank is always rank1
                            }
                            uint32_t st0[4];
                            state_at(sh, rg, pos, st0);
                            for (int k = 0; k < 4; ++k) sh.c.rng[k] = st0[k];
                            sh.base = pos;
                            sh.redo = -1;
                            sh.c.n_full = __popcll(mf);
                            sh.c.n_full_evals += __popcll(mf);
                            sh.c.n_partial_evals += __popcll(mp);
                        }
                        if (mf == 0) finish_iteration(cs, sh, iter, lockstep, lane);
                    }
                }
                __syncthreads();                                            // ---- barrier B
                STAMP(3);   // validation + plan + finish
                if (sh.redo < 0) break;
                redo = sh.redo;
                first = false;
            }
            if (sh.c.n_full > 0) {          // hand over to k_full; the next launch resumes after P0
                for (int c = tid; c < nc; c += blockDim.x) cs.prop[c] = sh.prop[c];
                if (tid == 0) sh.c.stage = ST_WAIT_FULL;
                break;
            }
        } else {
            if (wave == 0) finish_iteration(cs, sh, iter, lockstep, lane);
            __syncthreads();
            resume = false;
        }
        // ---------------- commit decisions + records (chain waves) ------------------------------------
        if (!producer)
            for (int c = wave; c < nc; c += NW) commit_chain(cs, sh, c, iter, lane);
        STAMP(4);   // commit
        if (lockstep) break;
    }
    __syncthreads();
    if (tid == 0) *cs.ctrl = sh.c;
    STAMP(5);
}

}  // namespace htm
