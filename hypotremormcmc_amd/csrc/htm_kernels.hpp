// htm_kernels.hpp -- the gfx950 kernels.  See DESIGN.md §3 for the per-kernel roofline notes.
//
//   k_full<NCH>   batched full log-likelihood: grid (event tiles, model groups), wave <-> event,
//                 lane <-> station; the HBM/L2-streaming kernel of the path (reference
//                 forward_calc_log_likelihood, src/cls_forward.f90:268-303).
//   k_sum_partials  deterministic second-stage reduction of k_full's per-workgroup partials.
//   k_step<NCH>   one workgroup per rank: RNG window, proposals, single-event partial updates
//                 (src/cls_forward.f90:307-362), Metropolis judge (src/cls_mcmc.f90:176-226), recording
//                 (src/hypo_tremor_mcmc.f90:270-280) and the temperature swap (src/cls_parallel.f90:100-216)
//                 for all chains of the rank; loops over iterations until a chain needs k_full.
//   k_syn / k_partial_one  the remaining `type forward` entry points.
#pragma once
#include "htm_device.hpp"

namespace htm {

// ---------------------------------------------------------------------------------------------------
struct FullJob {
    const double *hypo; long hypo_stride;
    const double *tc;   long tc_stride;
    const double *ac;   long ac_stride;
    const double *vs, *qs;          // [n_models]
    const Proposal *prop;           // chain mode: proposed override per model, else nullptr
    const int *list;                // chain mode: models needing a full evaluation, else nullptr
    const Ctrl *ctrl;               // chain mode: gate on ctrl->stage == ST_WAIT_FULL, n = ctrl->n_full
    int n_models;
    double *partial;                // [n_models][n_wg]
    int n_wg, epw;                  // event tiles; events per wave
};

template <int NCH>
__global__ __launch_bounds__(256) void k_full(FwdDev f, FullJob jb)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *s_sx = reinterpret_cast<double *>(smem);
    double *s_sy = s_sx + f.S;
    double *s_sz = s_sy + f.S;
    double *s_red = s_sz + f.S;   // 4 doubles

    int nm = jb.n_models;
    if (jb.ctrl) {
        if (jb.ctrl->stage != ST_WAIT_FULL) return;
        nm = jb.ctrl->n_full;
    }
    if (nm <= 0) return;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int j = threadIdx.x; j < f.S; j += blockDim.x) {
        s_sx[j] = f.sx[j]; s_sy[j] = f.sy[j]; s_sz[j] = f.sz[j];
    }
    __syncthreads();

    for (int k = blockIdx.y; k < nm; k += gridDim.y) {
        const int m = jb.list ? jb.list[k] : k;
        double beta = jb.vs[m], q = jb.qs[m];
        int ov_kind = 0, ov_idx = -1, ov_evt = -1, ov_cmp = 0;
        double ov_val = 0.0;
        if (jb.prop) {
            const Proposal &pr = jb.prop[m];
            ov_val = pr.x_new;
            if (pr.type == 1) beta = pr.x_new;
            else if (pr.type == 3) q = pr.x_new;
            else if (pr.type == 2 || pr.type == 4) { ov_kind = pr.type; ov_idx = pr.idx; }
            else { ov_evt = pr.idx / 3; ov_cmp = pr.idx - 3 * ov_evt; }
        }
        const double *hyp = jb.hypo + (size_t)m * jb.hypo_stride;
        const double *tc = jb.tc + (size_t)m * jb.tc_stride;
        const double *ac = jb.ac + (size_t)m * jb.ac_stride;

        double lane_acc = 0.0;
        if constexpr (NCH > 0) {
            StaRegs<NCH> st;
            load_sta_regs<NCH>(st, f.S, lane, s_sx, s_sy, s_sz, tc, ac, ov_kind, ov_idx, ov_val);
            for (int e = 0; e < jb.epw; ++e) {
                const int ev = (blockIdx.x * jb.epw + e) * 4 + wave;
                if (ev < f.E) {
                    double px[1] = {hyp[3 * ev]}, py[1] = {hyp[3 * ev + 1]}, pz[1] = {hyp[3 * ev + 2]};
                    if (ev == ov_evt) {
                        if (ov_cmp == 0) px[0] = ov_val; else if (ov_cmp == 1) py[0] = ov_val; else pz[0] = ov_val;
                    }
                    double out[1];
                    event_misfit<NCH, 1>(f, ev, lane, st, px, py, pz, beta, q, out);
                    lane_acc += out[0];
                }
            }
        } else {
            for (int e = 0; e < jb.epw; ++e) {
                const int ev = (blockIdx.x * jb.epw + e) * 4 + wave;
                if (ev < f.E) {
                    double px[1] = {hyp[3 * ev]}, py[1] = {hyp[3 * ev + 1]}, pz[1] = {hyp[3 * ev + 2]};
                    if (ev == ov_evt) {
                        if (ov_cmp == 0) px[0] = ov_val; else if (ov_cmp == 1) py[0] = ov_val; else pz[0] = ov_val;
                    }
                    double out[1];
                    event_misfit_generic<1>(f, ev, lane, s_sx, s_sy, s_sz, tc, ac, ov_kind, ov_idx, ov_val,
                                            px, py, pz, beta, q, out);
                    lane_acc += out[0];
                }
            }
        }
        const double tot = wave_sum1(lane_acc);
        if (lane == 0) s_red[wave] = tot;
        __syncthreads();
        if (threadIdx.x == 0)
            jb.partial[(size_t)m * jb.n_wg + blockIdx.x] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
        __syncthreads();
    }
}

// L[m] = -(sum of partials) - const_sum, fixed summation order (lane-strided, then the DPP tree)
__global__ __launch_bounds__(64) void k_sum_partials(const double *partial, int n_wg, double const_sum,
                                                     double *L)
{
    const int m = blockIdx.x, lane = threadIdx.x;
    double acc = 0.0;
    for (int k = lane; k < n_wg; k += 64) acc += partial[(size_t)m * n_wg + k];
    const double tot = wave_sum1(acc);
    if (lane == 0) L[m] = -tot - const_sum;
}

// ---------------------------------------------------------------------------------------------------
// calc_travel_time / calc_amp (+ _single): writes the demeaned synthetics.  wave <-> event.
// which: 0 = travel time, 1 = amplitude.  ev_only >= 0 restricts to one event and writes out[0..S).
__global__ __launch_bounds__(256) void k_syn(FwdDev f, const double *hypo, const double *corr, double beta,
                                             double q, int which, int ev_only, double *out)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int ev = blockIdx.x * 4 + wave;
    if (ev_only >= 0) { if (ev != 0) return; ev = ev_only; }
    if (ev >= f.E) return;
    const double x = hypo[3 * ev], y = hypo[3 * ev + 1], z = hypo[3 * ev + 2];
    const double *obs = which == 0 ? f.t_obs : f.a_obs;
    const double *prec = which == 0 ? f.t_prec : f.a_prec;
    const double psum = which == 0 ? f.psum_t[ev] : f.psum_a[ev];
    const size_t base = (size_t)ev * f.S;
    const double qbeta = q * beta;
    double acc = 0.0;
    for (int j = lane; j < f.S; j += 64) {
        const double dx = x - f.sx[j], dy = y - f.sy[j], dz = z - f.sz[j];
        const double d = sqrt(dx * dx + dy * dy + dz * dz);
        const double s = which == 0 ? d / beta - corr[j] : -(d * kPi * kFreq / qbeta) - log(d) - corr[j];
        acc += prec[base + j] * (s - obs[base + j]);
    }
    const double mean = wave_sum1(acc) / psum;
    double *o = ev_only >= 0 ? out : out + base;
    for (int j = lane; j < f.S; j += 64) {
        const double dx = x - f.sx[j], dy = y - f.sy[j], dz = z - f.sz[j];
        const double d = sqrt(dx * dx + dy * dy + dz * dz);
        const double s = which == 0 ? d / beta - corr[j] : -(d * kPi * kFreq / qbeta) - log(d) - corr[j];
        o[j] = s - mean;
    }
}

// partially_update_log_likelihood for one event, one wave (host-pointer API)
__global__ __launch_bounds__(64) void k_partial_one(FwdDev f, int ev, const double *xyz_old,
                                                    const double *xyz_new, const double *tc,
                                                    const double *ac, double beta, double q, double L_old,
                                                    double *L_out)
{
    const int lane = threadIdx.x;
    double px[2] = {xyz_old[0], xyz_new[0]}, py[2] = {xyz_old[1], xyz_new[1]}, pz[2] = {xyz_old[2], xyz_new[2]};
    double out[2];
    event_misfit_generic<2>(f, ev, lane, f.sx, f.sy, f.sz, tc, ac, 0, -1, 0.0, px, py, pz, beta, q, out);
    const double tot = wave_sum1(out[0] - out[1]);
    if (lane == 0) *L_out = L_old + tot;
}

// ---------------------------------------------------------------------------------------------------
// k_step
// ---------------------------------------------------------------------------------------------------
struct StepShared {           // fixed-size part of k_step's LDS (the station table follows it)
    uint32_t raw[4 + kMaxWindow + 4];
    double U[kMaxWindow], LOGU[kMaxWindow], G[kMaxWindow];
    Proposal prop[kMaxChains];
    double temp[kMaxChains], L[kMaxChains];
    int pos[kMaxChains + 1];
    int slot_l[kMaxChains], slot_s[kMaxChains];
    uint32_t gen[4];          // generator state after raw[.. wg]
    int wg;                   // transforms valid for positions < wg; raws valid for <= wg
    int anyfail;
    Ctrl c;
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

// thread-0 only: make position p (and the raw after it) available
__device__ inline bool win_ensure(StepShared &sh, int p)
{
    while (p >= sh.wg) {
        if (sh.wg + 1 >= kMaxWindow) return false;
        uint32_t x = sh.gen[0], y = sh.gen[1], z = sh.gen[2], w = sh.gen[3];
        const uint32_t r = xs128_next(x, y, z, w);
        sh.gen[0] = x; sh.gen[1] = y; sh.gen[2] = z; sh.gen[3] = w;
        const int k = sh.wg;              // raw[k] exists, raw[k+1] := r
        sh.raw[4 + k + 1] = r;
        const uint32_t r0 = sh.raw[4 + k];
        const double u = u_of(r0);
        sh.U[k] = u; sh.LOGU[k] = log(u); sh.G[k] = g_of(r0, r);
        sh.wg = k + 1;
    }
    return true;
}

// cls_mcmc.f90:134-165 + cls_model.f90:162-190 for chain c starting at stream position pos.
// Returns the number of draws this chain step consumes (judge draw included iff prior_ok).
__device__ inline int propose_chain(const ChainsDev &cs, StepShared &sh, int c, int pos, bool may_extend)
{
    Proposal pr;
    if (may_extend && !win_ensure(sh, pos + 6)) { sh.c.err = -4; }
    const double a = sh.U[pos];
    int gpos;
    pr.evt = -999;
    if (a < cs.th1) { pr.type = 1; pr.idx = 0; gpos = pos + 1; }
    else if (a < cs.th2) { pr.type = 2; pr.idx = (int)(sh.U[pos + 1] * cs.S); gpos = pos + 2; }
    else if (a < cs.th3) { pr.type = 3; pr.idx = 0; gpos = pos + 1; }
    else if (a < cs.th4) { pr.type = 4; pr.idx = (int)(sh.U[pos + 1] * cs.S); gpos = pos + 2; }
    else {
        const int id = (int)(sh.U[pos + 1] * cs.E) + 1;
        const int icmp = (int)(sh.U[pos + 2] * 3);
        pr.idx = 3 * id - icmp - 1; pr.type = 5 + icmp; pr.evt = id; gpos = pos + 3;
    }
    const ModelDev M = pick_model(cs, pr.type);
    const size_t o = (size_t)c * M.nx + pr.idx;
    const double x_old = M.x[o], mu = M.mu[o], sigma = M.sigma[o], step = M.step[o];
    const double x_new = x_old + sh.G[gpos] * step;
    const double da = x_new - mu, db = x_old - mu;
    double lpr = -(da * da - db * db) / (2.0 * sigma * sigma);
    int ok = 1;
    if (M.ptype[o] == 1) {
        if (x_new <= mu) { lpr = (double)-1.0e+30f; ok = 0; }
        else lpr = lpr + log(x_new - mu) - log(x_old - mu);
    }
    pr.x_new = x_new; pr.lpr = lpr; pr.prior_ok = ok; pr.need_full = 0; pr.accepted = 0; pr.L_new = 0.0;
    const int jpos = gpos + 2;
    pr.r_judge = ok ? sh.U[jpos] : 0.0;
    pr.logr_judge = ok ? sh.LOGU[jpos] : 0.0;
    sh.prop[c] = pr;
    return (jpos - pos) + ok;
}

__device__ inline int draws_if_ok(const ChainsDev &cs, double a)
{
    if (a < cs.th1) return 4;
    if (a < cs.th2) return 5;
    if (a < cs.th3) return 4;
    if (a < cs.th4) return 5;
    return 6;
}

template <int NCH>
__global__ __launch_bounds__(512) void k_step(FwdDev f, ChainsDev cs, int mode, int target_arg,
                                               const double *gathered)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    StepShared &sh = *reinterpret_cast<StepShared *>(smem);
    double *s_sx = reinterpret_cast<double *>(smem + ((sizeof(StepShared) + 15) & ~size_t(15)));
    double *s_sy = s_sx + f.S;
    double *s_sz = s_sy + f.S;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NW = blockDim.x >> 6;
    const int nc = cs.n_chains;
    const bool lockstep = (mode != MODE_RUN);
    const int n_all = cs.n_procs * nc;
    const int RW = 4 + 2 * nc;     // swap-record words

    if (tid == 0) {
        sh.c = *cs.ctrl;
        if (target_arg >= 0) sh.c.iter_target = target_arg;
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

    bool resume_full = (sh.c.stage == ST_WAIT_FULL);
    if (mode == MODE_FINISH && !resume_full) return;
    if (sh.c.stage == ST_WAIT_SWAP) return;          // nothing to do until the swap is applied
    if (resume_full)
        for (int c = tid; c < nc; c += blockDim.x) sh.prop[c] = cs.prop[c];
    __syncthreads();

    for (;;) {
        int iter = sh.c.iter_done + 1;
        if (!resume_full) {
            // ---------------- S0: anything left to do? ------------------------------------------------
            if (sh.c.iter_done >= sh.c.iter_target || sh.c.stop || sh.c.err) break;
            if (sh.c.n_lik + nc > cs.cap_lik || sh.c.n_smp + nc > cs.cap_smp) {
                __syncthreads();
                if (tid == 0) { if (lockstep) sh.c.err = -5; else sh.c.stop = 1; }
                __syncthreads();
                break;
            }
            // ---------------- S0a: raw window (thread 0), mod_random.f90:63-71 -------------------------
            int W = ((6 * nc + 8 + 63) / 64) * 64;
            if (W > kMaxWindow - 8) W = kMaxWindow - 8;
            if (tid == 0) {
                uint32_t x = sh.c.rng[0], y = sh.c.rng[1], z = sh.c.rng[2], w = sh.c.rng[3];
                sh.raw[0] = x; sh.raw[1] = y; sh.raw[2] = z; sh.raw[3] = w;
                for (int p = 0; p <= W; ++p) sh.raw[4 + p] = xs128_next(x, y, z, w);
                sh.gen[0] = x; sh.gen[1] = y; sh.gen[2] = z; sh.gen[3] = w;
                sh.wg = W;
                sh.anyfail = 0;
            }
            __syncthreads();
            // ---------------- S0b: U / log U / Box-Muller for every window position, in parallel --------
            for (int p = tid; p < W; p += blockDim.x) {
                const uint32_t r0 = sh.raw[4 + p], r1 = sh.raw[4 + p + 1];
                const double u = u_of(r0);
                sh.U[p] = u; sh.LOGU[p] = log(u); sh.G[p] = g_of(r0, r1);
            }
            __syncthreads();
            // ---------------- S0c: optimistic scan of stream positions (assumes prior_ok) ---------------
            if (tid == 0) {
                int pos = 0;
                for (int c = 0; c < nc; ++c) {
                    sh.pos[c] = pos;
                    if (!win_ensure(sh, pos + 6)) sh.c.err = -4;
                    pos += draws_if_ok(cs, sh.U[pos]);
                }
                sh.pos[nc] = pos;
            }
            __syncthreads();
            for (int c = wave; c < nc; c += NW)
                if (lane == 0) {
                    propose_chain(cs, sh, c, sh.pos[c], false);
                    if (!sh.prop[c].prior_ok) atomicOr(&sh.anyfail, 1);
                }
            __syncthreads();
            if (sh.anyfail) {   // rare: a Rayleigh prior rejected => later chains start one draw earlier
                if (tid == 0) {
                    int pos = 0;
                    for (int c = 0; c < nc; ++c) { sh.pos[c] = pos; pos += propose_chain(cs, sh, c, pos, true); }
                    sh.pos[nc] = pos;
                }
                __syncthreads();
            }
            // ---------------- S0d: swap plan + RNG commit (cls_parallel.f90:226-230,:294) ---------------
            if (tid == 0) {
                int pos = sh.pos[nc];
                sh.c.swap_i1 = -1; sh.c.swap_i2 = -1; sh.c.swap_r = 0.0; sh.c.swap_logr = 0.0;
                if (n_all > 1) {
                    if (cs.rank == 0) {
                        if (!win_ensure(sh, pos + 2)) sh.c.err = -4;
                        const int i1 = (int)(sh.U[pos++] * cs.n_procs * nc);
                        int i2;
                        for (;;) {
                            if (!win_ensure(sh, pos + 2)) { sh.c.err = -4; i2 = (i1 + 1) % n_all; break; }
                            i2 = (int)(sh.U[pos++] * cs.n_procs * nc);
                            if (i1 != i2) break;
                        }
                        sh.c.swap_i1 = i1; sh.c.swap_i2 = i2;
                    }
                    if (!win_ensure(sh, pos + 2)) sh.c.err = -4;
                    sh.c.swap_r = sh.U[pos]; sh.c.swap_logr = sh.LOGU[pos];
                    for (int k = 0; k < 4; ++k) sh.c.rng_plus1[k] = sh.raw[pos + 1 + k];
                    if (!lockstep) pos++;        // single rank: this rank is always rank1
                }
                for (int k = 0; k < 4; ++k) sh.c.rng[k] = sh.raw[pos + k];
            }
            __syncthreads();

            // ---------------- S1: single-event partial update, wave <-> chain ---------------------------
            for (int c = wave; c < nc; c += NW) {
                const Proposal pr = sh.prop[c];
                if (!pr.prior_ok) continue;
                if (pr.evt > 0 && iter > 1) {
                    const int ev = pr.evt - 1, cmp = pr.idx - 3 * ev;
                    const double *hyp = cs.hypo.x + (size_t)c * cs.hypo.nx + 3 * ev;
                    double px[2], py[2], pz[2];
                    px[0] = px[1] = hyp[0]; py[0] = py[1] = hyp[1]; pz[0] = pz[1] = hyp[2];
                    if (cmp == 0) px[1] = pr.x_new; else if (cmp == 1) py[1] = pr.x_new; else pz[1] = pr.x_new;
                    const double beta = cs.vs.x[c], q = cs.qs.x[c];
                    const double *tc = cs.tc.x + (size_t)c * cs.S, *ac = cs.ac.x + (size_t)c * cs.S;
                    double out[2];
                    if constexpr (NCH > 0) {
                        StaRegs<NCH> st;
                        load_sta_regs<NCH>(st, f.S, lane, s_sx, s_sy, s_sz, tc, ac, 0, -1, 0.0);
                        event_misfit<NCH, 2>(f, ev, lane, st, px, py, pz, beta, q, out);
                    } else {
                        event_misfit_generic<2>(f, ev, lane, s_sx, s_sy, s_sz, tc, ac, 0, -1, 0.0, px, py, pz,
                                                beta, q, out);
                    }
                    const double tot = wave_sum1(out[0] - out[1]);
                    if (lane == 0) sh.prop[c].L_new = sh.L[c] + tot;
                } else if (lane == 0) {
                    sh.prop[c].need_full = 1;
                }
            }
            __syncthreads();
            if (tid == 0) {
                int n = 0, np = 0;
                for (int c = 0; c < nc; ++c) {
                    if (sh.prop[c].need_full) cs.full_list[n++] = c;
                    else if (sh.prop[c].prior_ok) np++;
                }
                sh.c.n_full = n;
                sh.c.n_full_evals += n;
                sh.c.n_partial_evals += np;
            }
            __syncthreads();
            if (sh.c.n_full > 0) {        // hand over to k_full; resume at S2 in the next k_step launch
                for (int c = tid; c < nc; c += blockDim.x) cs.prop[c] = sh.prop[c];
                if (tid == 0) sh.c.stage = ST_WAIT_FULL;
                __syncthreads();
                break;
            }
        }
        resume_full = false;

        // ---------------- S2: collect k_full results, judge (cls_mcmc.f90:186-222) -----------------------
        for (int c = wave; c < nc; c += NW) {
            Proposal pr = sh.prop[c];
            if (pr.need_full) {
                double acc = 0.0;
                for (int k = lane; k < cs.n_wg; k += 64) acc += cs.partial[(size_t)c * cs.n_wg + k];
                const double tot = wave_sum1(acc);
                pr.L_new = -tot - f.const_sum;
            }
            if (lane == 0) {
                const double T = sh.temp[c];
                const bool cool = T < 1.0 + kEps;
                if (cool) cs.n_propose[c * 7 + pr.type - 1] += 1;
                bool acc = false;
                if (pr.prior_ok) {
                    double ratio = (pr.L_new - sh.L[c]) / T;
                    ratio = ratio + pr.lpr;
                    if (pr.r_judge >= kEps) { if (pr.logr_judge <= ratio) acc = true; }
                }
                if (acc) {
                    const ModelDev M = pick_model(cs, pr.type);
                    M.x[(size_t)c * M.nx + pr.idx] = pr.x_new;
                    sh.L[c] = pr.L_new;
                    cs.L[c] = pr.L_new;
                    if (cool) cs.n_accept[c * 7 + pr.type - 1] += 1;
                }
                const int row = sh.c.slog_n + c;
                if (row < sh.c.slog_cap) {
                    int32_t *ir = cs.slog_i + 8 * (size_t)row;
                    double *dr = cs.slog_d + 4 * (size_t)row;
                    ir[0] = iter; ir[1] = c; ir[2] = pr.type; ir[3] = pr.idx + 1; ir[4] = pr.prior_ok;
                    ir[5] = acc ? 1 : 0; ir[6] = pr.need_full; ir[7] = 0;
                    dr[0] = pr.x_new; dr[1] = pr.L_new; dr[2] = sh.L[c]; dr[3] = T;
                }
            }
        }
        __syncthreads();
        // ---------------- recording (hypo_tremor_mcmc.f90:270-280), file order = chain order --------------
        if (tid == 0) {
            if (sh.c.slog_n < sh.c.slog_cap) sh.c.slog_n += nc;
            for (int c = 0; c < nc; ++c) {
                sh.slot_l[c] = -1; sh.slot_s[c] = -1;
                if (sh.temp[c] < 1.0 + kEps && (iter % cs.n_interval) == 1) {
                    if (iter > cs.n_burn) sh.slot_s[c] = sh.c.n_smp++;
                    sh.slot_l[c] = sh.c.n_lik++;
                }
            }
        }
        __syncthreads();
        for (int c = wave; c < nc; c += NW) {
            const int sl = sh.slot_l[c], ss = sh.slot_s[c];
            if (sl >= 0 && lane == 0) { cs.lik_iter[sl] = iter; cs.lik_chain[sl] = c; cs.lik_val[sl] = sh.L[c]; }
            if (ss >= 0) {
                const int nh = cs.hypo.nx, rec = nh + 2 * cs.S + 2;
                double *dst = cs.smp_data + (size_t)ss * rec;
                const double *hx = cs.hypo.x + (size_t)c * nh;
                for (int k = lane; k < nh; k += 64) dst[k] = hx[k];
                for (int k = lane; k < cs.S; k += 64) {
                    dst[nh + k] = cs.tc.x[(size_t)c * cs.S + k];
                    dst[nh + cs.S + k] = cs.ac.x[(size_t)c * cs.S + k];
                }
                if (lane == 0) {
                    dst[nh + 2 * cs.S] = cs.vs.x[c]; dst[nh + 2 * cs.S + 1] = cs.qs.x[c];
                    cs.smp_iter[ss] = iter; cs.smp_chain[ss] = c;
                }
            }
        }
        // ---------------- swap_temperature ------------------------------------------------------------
        if (lockstep) {
            if (tid == 0) {      // export this rank's record; the swap itself happens in MODE_APPLY
                double *rec = cs.swap_rec;
                rec[0] = (double)sh.c.swap_i1; rec[1] = (double)sh.c.swap_i2;
                rec[2] = sh.c.swap_r; rec[3] = (double)iter;
                for (int c = 0; c < nc; ++c) { rec[4 + 2 * c] = sh.temp[c]; rec[5 + 2 * c] = sh.L[c]; }
                sh.c.stage = ST_WAIT_SWAP;
            }
            __syncthreads();
            break;
        }
        if (tid == 0) {
            if (n_all > 1) {     // both chains live on this rank: cls_parallel.f90:121-136 + :285-302
                const int c1 = sh.c.swap_i1, c2 = sh.c.swap_i2;
                const double T1 = sh.temp[c1], T2 = sh.temp[c2];
                const double del_s = (sh.L[c2] - sh.L[c1]) * (1.0 / T1 - 1.0 / T2);
                if (sh.c.swap_r >= kEps && sh.c.swap_logr <= del_s) {
                    sh.temp[c1] = T2; sh.temp[c2] = T1;
                    cs.temp[c1] = T2; cs.temp[c2] = T1;
                }
            }
            sh.c.iter_done = iter;
            sh.c.stage = ST_IDLE;
        }
        __syncthreads();
    }
    if (tid == 0) *cs.ctrl = sh.c;
}

// ---------------------------------------------------------------------------------------------------
// self-test: DPP wave_sum against a serial loop of the same tree order; device RNG against host values
// ---------------------------------------------------------------------------------------------------
__global__ void k_selftest(const double *in, double *out_dpp, double *out_ref, uint32_t *rng_out,
                           double *rng_d)
{
    const int lane = threadIdx.x;
    double v[2] = {in[lane], in[64 + lane]};
    wave_sum<2>(v);
    if (lane == 0) {
        out_dpp[0] = v[0]; out_dpp[1] = v[1];
        for (int s = 0; s < 2; ++s) {   // same association as the DPP tree
            double t[64];
            for (int i = 0; i < 64; ++i) t[i] = in[64 * s + i];
            double q[16];
            for (int i = 0; i < 16; ++i) q[i] = (t[4 * i] + t[4 * i + 1]) + (t[4 * i + 2] + t[4 * i + 3]);
            double r[4];
            for (int i = 0; i < 4; ++i) r[i] = (q[4 * i] + q[4 * i + 1]) + (q[4 * i + 2] + q[4 * i + 3]);
            out_ref[s] = (r[3] + r[2]) + (r[1] + r[0]);
        }
        uint32_t x = 0x4b88a366u, y = 0x1b11733cu, z = 0x097044b6u, w = 0x00676ea2u;  // rank-0 seed state
        for (int i = 0; i < 8; ++i) { rng_out[i] = xs128_next(x, y, z, w); rng_d[i] = u_of(rng_out[i]); }
        rng_d[8] = g_of(rng_out[0], rng_out[1]);
    }
}

}  // namespace htm
