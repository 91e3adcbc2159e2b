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
    const double *vs, *qs;          // [n_models]   (batch mode)
    const FullDesc *desc;           // chain mode: work order written by k_step; blockIdx.y = entry
    int n_models;                   // batch mode: models m = blockIdx.y, blockIdx.y + gridDim.y, ...
    double *partial;                // [n_models][n_wg]
    int n_wg, epw;                  // event tiles; events per wave
};

// Full log-likelihood (reference forward_calc_log_likelihood, src/cls_forward.f90:268-303) of one or many
// models.  Grid: x = event tile, y = model (chain mode: entry of the work order).  wave <-> event,
// lane <-> station; each lane keeps its station's coordinates and corrections in registers; the four
// observation streams are read with coalesced 512-B wave loads.  Output: one partial sum per workgroup,
// reduced in a fixed order by the consumer (deterministic).
// BATCH only gives the two launch shapes distinct kernel names (profilers list them separately):
// false = work order from k_step inside the MCMC loop, true = n_models stacked models.
template <int NCH, bool BATCH, bool F32 = false>
__global__ __launch_bounds__(256) void k_full(FwdDev f, FullJob jb)
{
    __shared__ double s_red[4];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    int k0 = blockIdx.y, kstep = gridDim.y, nm = jb.n_models;
    FullEntry en;
    en.chain = 0; en.type = 0; en.idx = -1; en.pad = 0; en.x_new = 0.0; en.beta = 0.0; en.q = 0.0;
    if (jb.desc) {                                  // one round trip: header + this block's entry
        const int n = jb.desc->n;
        en = jb.desc->e[blockIdx.y];
        if ((int)blockIdx.y >= n) return;
        kstep = 1 << 30; nm = blockIdx.y + 1;       // exactly one model per block
    }
    // one event per wave (E <= 4 * event tiles): the wave's observation rows do not depend on the model -- load
    // them once and keep them in registers over all models of this block
    constexpr int NR = NCH > 0 ? NCH : 1;
    ObsRegs<NR, F32> ob_keep;
    const bool keep_obs = NCH > 0 && jb.epw == 1 && (int)(blockIdx.x * 4 + wave) < f.E;
    if constexpr (NCH > 0) {
        if (keep_obs) load_obs_regs<NCH, F32>(ob_keep, f, blockIdx.x * 4 + wave, lane);
    }
    // Stacked models, one event per wave, its rows resident (the shape of the batched call): rounds of up to four PAIRS of
    // models.  A pair is evaluated side by side (event_misfit_models: with one station per lane a single model's evaluation is
    // one dependent chain after the other).  What does not depend on the event is done once per round, not once per wave and
    // model: the two divisions of a model (1 / vs, pi f / (qs vs): cls_forward.f90:118, :204) by one lane each, and the sums
    // over lanes and waves of the round's eight models by the whole block in one pass over LDS -- in the association of
    // wave_sum and of the single-model trip below (a balanced tree over the lanes, then (w0 + w1) + (w2 + w3)), so a stacked
    // model's value is bit for bit what the one-by-one call gives (tests/test_gpu_forward.py).
    __shared__ double s_rb[8], s_ka[8];
    __shared__ __attribute__((aligned(16))) double s_out[8][4][64];
    if constexpr (NCH > 0) {
        if (!jb.desc && jb.epw == 1) {           // (block-uniform: a wave of the last tile without an event adds zeros and keeps the barriers)
            StaRegs<NCH> geo;
            load_sta_regs<NCH>(geo, f.S, lane, f.sx, f.sy, f.sz, jb.tc, jb.ac, 0, -1, 0.0);      // (coordinates; the corrections come per model)
            const int ev = blockIdx.x * 4 + wave;
            const int evc = keep_obs ? ev : 0;
            int k = k0;
            while (k + kstep < nm) {
                const int left = (nm - k + kstep - 1) / kstep;          // models of this block from k on
                const int np = left / 2 < 4 ? left / 2 : 4;             // pairs of this round
                if ((int)threadIdx.x < 2 * np) {
                    const int m = k + (int)threadIdx.x * kstep;
                    const double beta = jb.vs[m], q = jb.qs[m];
                    s_rb[threadIdx.x] = 1.0 / beta; s_ka[threadIdx.x] = (kPi * kFreq) / (q * beta);
                }
                __syncthreads();      // (also: the round before has been summed before its s_out is overwritten)
                for (int j = 0; j < np; ++j) {
                    double tcm[2][NCH], acm[2][NCH], px[2], py[2], pz[2], rbeta[2], katt[2], out[2] = {0.0, 0.0};
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
                        const int m = k + (2 * j + p) * kstep;
                        const double *hyp = jb.hypo + (size_t)m * jb.hypo_stride;
                        const double *tc = jb.tc + (size_t)m * jb.tc_stride, *ac = jb.ac + (size_t)m * jb.ac_stride;
                        rbeta[p] = s_rb[2 * j + p]; katt[p] = s_ka[2 * j + p];
                        px[p] = hyp[3 * evc]; py[p] = hyp[3 * evc + 1]; pz[p] = hyp[3 * evc + 2];
#pragma unroll
                        for (int c = 0; c < NCH; ++c) {
                            const int jj = lane + 64 * c;
                            tcm[p][c] = jj < f.S ? tc[jj] : 0.0; acm[p][c] = jj < f.S ? ac[jj] : 0.0;
                        }
                    }
                    if (keep_obs) event_misfit_models<NCH, 2, F32, true>(f, ob_keep, lane, geo, tcm, acm, px, py, pz, rbeta, katt, out);
                    s_out[2 * j][wave][lane] = out[0]; s_out[2 * j + 1][wave][lane] = out[1];
                }
                __syncthreads();
                {   // thread t: model t >> 5 of the round, wave (t >> 3) & 3, lanes 8 (t & 7) .. + 7
                    const int t = threadIdx.x, i = t >> 5;
                    const double *src = &s_out[0][0][0] + 8 * t;
                    double v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = src[u];
                    double x = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
                    x += dpp_mov_f64<0xB1, 0xF>(x);      // the eight octets of a wave's 64 lanes: quad_perm [1,0,3,2],
                    x += dpp_mov_f64<0x4E, 0xF>(x);      // [2,3,0,1],
                    x += dpp_mov_f64<0x141, 0xF>(x);     // row_half_mirror
                    x += dpp_mov_f64<0x128, 0xF>(x);     // row_ror:8 -- the other wave of the pair (w0 + w1, w2 + w3)
                    x += dpp_mov_f64<0x142, 0xA>(x);     // row_bcast:15 -> rows 1, 3: (w2 + w3) + (w0 + w1)
                    if ((t & 31) == 31 && i < 2 * np) jb.partial[(size_t)(k + i * kstep) * jb.n_wg + blockIdx.x] = x;
                }
                k += 2 * np * kstep;
            }
            __syncthreads();   // (the single-model trips below use s_red behind their own barriers; keeps the block together)
            k0 = k;            // (an odd model out takes the single-model trip below)
        }
    }
    for (int k = k0; k < nm; k += kstep) {
        const int m = jb.desc ? en.chain : k;
        double beta, q;
        int ov_kind = 0, ov_idx = -1, ov_evt = -1, ov_cmp = 0;
        const double ov_val = en.x_new;
        if (jb.desc) {
            beta = en.beta; q = en.q;
            if (en.type == 2 || en.type == 4) { ov_kind = en.type; ov_idx = en.idx; }
            else if (en.type >= 5) { ov_evt = en.idx / 3; ov_cmp = en.idx - 3 * ov_evt; }
        } else {
            beta = jb.vs[m]; q = jb.qs[m];
        }
        const double *hyp = jb.hypo + (size_t)m * jb.hypo_stride;
        const double *tc = jb.tc + (size_t)m * jb.tc_stride;
        const double *ac = jb.ac + (size_t)m * jb.ac_stride;

        double lane_acc = 0.0;
        if constexpr (NCH > 0) {
            const double rbeta_m = 1.0 / beta, katt_m = (kPi * kFreq) / (q * beta);      // (once per model, not once per event)
            StaRegs<NCH> st;
            load_sta_regs<NCH>(st, f.S, lane, f.sx, f.sy, f.sz, tc, ac, ov_kind, ov_idx, ov_val);
            for (int e = 0; e < jb.epw; ++e) {
                const int ev = (blockIdx.x * jb.epw + e) * 4 + wave;
                if (ev < f.E) {
                    // plain selects: an if/else-if/else chain of stores into these arrays was miscompiled
                    // by hipcc 7.2 at -O3 (the final else-store was dropped), see DESIGN.md §7
                    const bool ov = ev == ov_evt;
                    const double px[1] = {(ov && ov_cmp == 0) ? ov_val : hyp[3 * ev]};
                    const double py[1] = {(ov && ov_cmp == 1) ? ov_val : hyp[3 * ev + 1]};
                    const double pz[1] = {(ov && ov_cmp == 2) ? ov_val : hyp[3 * ev + 2]};
                    double out[1];
                    if (keep_obs) {
                        event_misfit<NCH, 1, F32, true>(f, ob_keep, lane, st, px, py, pz, rbeta_m, katt_m, out);
                    } else {
                        ObsRegs<NCH, F32> ob;
                        load_obs_regs<NCH, F32>(ob, f, ev, lane);
                        event_misfit<NCH, 1, F32, true>(f, ob, lane, st, px, py, pz, rbeta_m, katt_m, out);
                    }
                    lane_acc += out[0];
                }
            }
        } else {
            for (int e = 0; e < jb.epw; ++e) {
                const int ev = (blockIdx.x * jb.epw + e) * 4 + wave;
                if (ev < f.E) {
                    const bool ov = ev == ov_evt;
                    const double px[1] = {(ov && ov_cmp == 0) ? ov_val : hyp[3 * ev]};
                    const double py[1] = {(ov && ov_cmp == 1) ? ov_val : hyp[3 * ev + 1]};
                    const double pz[1] = {(ov && ov_cmp == 2) ? ov_val : hyp[3 * ev + 2]};
                    double out[1];
                    event_misfit_generic<1>(f, ev, lane, f.sx, f.sy, f.sz, tc, ac, ov_kind, ov_idx, ov_val,
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
        const double s = which == 0 ? d / beta - corr[j] : -(d * kPi * kFreq / qbeta) - htm_log(d) - corr[j];
        acc += prec[base + j] * (s - obs[base + j]);
    }
    const double mean = wave_sum1(acc) / psum;
    double *o = ev_only >= 0 ? out : out + base;
    for (int j = lane; j < f.S; j += 64) {
        const double dx = x - f.sx[j], dy = y - f.sy[j], dz = z - f.sz[j];
        const double d = sqrt(dx * dx + dy * dy + dz * dz);
        const double s = which == 0 ? d / beta - corr[j] : -(d * kPi * kFreq / qbeta) - htm_log(d) - corr[j];
        o[j] = s - mean;
    }
}

// partially_update_log_likelihood for one event, one wave (host-pointer API)
template <int NCH, bool F32 = false>
__global__ __launch_bounds__(64) void k_partial_one(FwdDev f, int ev, const double *xyz_old,
                                                    const double *xyz_new, const double *tc,
                                                    const double *ac, double beta, double q, double L_old,
                                                    double *L_out)
{
    const int lane = threadIdx.x;
    double px[2] = {xyz_old[0], xyz_new[0]}, py[2] = {xyz_old[1], xyz_new[1]}, pz[2] = {xyz_old[2], xyz_new[2]};
    double out[2];
    if constexpr (NCH > 0) {        // the register path of the chain kernels (honours the fp32-forward mode)
        StaRegs<NCH> st;
        ObsRegs<NCH, F32> ob;
        load_sta_regs<NCH>(st, f.S, lane, f.sx, f.sy, f.sz, tc, ac, 0, -1, 0.0);
        load_obs_regs<NCH, F32>(ob, f, ev, lane);
        event_misfit<NCH, 2, F32>(f, ob, lane, st, px, py, pz, beta, q, out);
    } else {
        event_misfit_generic<2>(f, ev, lane, f.sx, f.sy, f.sz, tc, ac, 0, -1, 0.0, px, py, pz, beta, q, out);
    }
    const double tot = wave_sum1(out[0] - out[1]);
    if (lane == 0) *L_out = L_old + tot;
}

}  // namespace htm
#include "htm_step.hpp"
namespace htm {

// ---------------------------------------------------------------------------------------------------
// self-test: DPP wave_sum against a serial loop of the same tree order; device RNG against host values
// ---------------------------------------------------------------------------------------------------
// htm_selftest_math: the forward model's own logarithm / square root (htm_device.hpp) on n arbitrary arguments
__global__ void k_mathtest(int which, const double *x, double *y, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (which == 4) {            // the matrix-pipe wave sum: every lane's result for its wave's 64 values (n a multiple of 64)
        if (i < n) y[i] = wave_sum_mfma(x[i]);
        return;
    }
    if (which == 5 || which == 6) {     // four sums per wave at once (5, wave_sum<4>) and one by one (6): x = four blocks of n / 4 values
        const int q = n / 4;
        if (i < q) {
            double v[4] = {x[i], x[q + i], x[2 * q + i], x[3 * q + i]};
            if (which == 5) wave_sum<4>(v);
            else { for (int k = 0; k < 4; ++k) v[k] = wave_sum1(v[k]); }
            for (int k = 0; k < 4; ++k) y[k * q + i] = v[k];
        }
        return;
    }
    if (i < n) y[i] = which == 0 ? htm_log(x[i]) : which == 1 ? htm_sqrt(x[i]) : which == 2 ? sqrt(x[i]) : log(x[i]);   // 3: the device library's log (Rayleigh prior ratio, htm_step.hpp)
}

__global__ void k_selftest(const double *in, double *out_dpp, double *out_ref, uint32_t *rng_out,
                           double *rng_d)
{
    const int lane = threadIdx.x;
    {   // DPP inclusive scan vs a serial prefix sum (draw counts are 3..6, use a wider spread)
        const int v = 3 + ((lane * 7 + 1) % 5);
        const int sc = wave_incl_scan(v);
        int ref = 0;
        for (int i = 0; i <= lane; ++i) ref += 3 + ((i * 7 + 1) % 5);
        const unsigned long long bad = __ballot(sc != ref);
        if (lane == 0) rng_d[9] = bad ? 1.0 : 0.0;
    }
    double v[2] = {in[lane], in[64 + lane]};
    wave_sum<2>(v);
    if (lane == 0) {
        out_dpp[0] = v[0]; out_dpp[1] = v[1];
        for (int s = 0; s < 2; ++s) {   // same association as wave_sum's
            double t[64];
            for (int i = 0; i < 64; ++i) t[i] = in[64 * s + i];
            if (HTM_MFMA_SUM != 0) {     // the matrix instruction adds its four products in order of k, from zero
                double S[16], G[4];
                for (int i = 0; i < 16; ++i) S[i] = ((t[i] + t[i + 16]) + t[i + 32]) + t[i + 48];
                for (int g = 0; g < 4; ++g) G[g] = (S[g] + S[g + 4]) + (S[g + 8] + S[g + 12]);
                out_ref[s] = ((G[0] + G[1]) + G[2]) + G[3];
                continue;
            }
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
