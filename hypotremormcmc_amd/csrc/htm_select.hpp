// htm_select.hpp -- order statistics of recorded samples (step 6 of the reference pipeline:
// src/cls_statistics.f90:216-264, :345-431 sort every parameter's n_mod samples with quick_sort and print the
// elements il, im, iu of the sorted column).  Sorting is not needed for that: the r-th smallest of a column is
// found exactly by a most-significant-digit radix SELECT on the order-preserving integer image of the doubles.
//
// Layout: samples are [n_mod][ld] row-major (one recorded model per row -- the record order of the sample files),
// parameter p in column p.  lane <-> column, so every load is a coalesced 512-B row segment; the RG waves of a
// workgroup split the rows.  16 passes of 4 bits; per pass each thread counts, for each requested rank, the
// digits of the rows that still match that rank's prefix (private LDS counters, no atomics), then every thread
// of a column walks the 16 totals to fix the next digit.  HBM/L2-bound: 16 x n_mod x n_par x 8 B per call.
#pragma once
#include <cstdint>
#include <hip/hip_runtime.h>

namespace htm {

constexpr int kSelRanks = 3;      // il, im, iu
constexpr int kSelRG = 4;         // row groups (waves) per workgroup
constexpr int kSelUnroll = 8;     // rows in flight per thread

__device__ __forceinline__ unsigned long long sel_key(double v)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    return b ^ ((b >> 63) ? ~0ull : 0x8000000000000000ull);       // a < b  <=>  key(a) < key(b)
}
__device__ __forceinline__ double sel_val(unsigned long long k)
{
    const unsigned long long b = (k >> 63) ? (k ^ 0x8000000000000000ull) : ~k;
    return __longlong_as_double((long long)b);
}

// ranks[r] = 0-based rank (0 = smallest) of the wanted element, r < kSelRanks
__global__ __launch_bounds__(64 * kSelRG) void k_select(const double *x, long n_mod, long n_par, long ld, int r0,
                                                        int r1, int r2, double *out)
{
    __shared__ int cnt[kSelRanks][16][kSelRG][64];
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const long p = (long)blockIdx.x * 64 + lane;
    const bool live = p < n_par;
    unsigned long long prefix[kSelRanks] = {0ull, 0ull, 0ull};
    long remaining[kSelRanks] = {r0, r1, r2};
    for (int shift = 60; shift >= 0; shift -= 4) {
#pragma unroll
        for (int r = 0; r < kSelRanks; ++r)
#pragma unroll
            for (int d = 0; d < 16; ++d) cnt[r][d][g][lane] = 0;
        if (live) {
            // kSelUnroll rows in flight per thread: the loads are independent, only the LDS counters serialise
            long k = g;
            for (; k + (kSelUnroll - 1) * kSelRG < n_mod; k += kSelUnroll * kSelRG) {
                unsigned long long key[kSelUnroll];
#pragma unroll
                for (int u = 0; u < kSelUnroll; ++u) key[u] = sel_key(x[(k + u * kSelRG) * ld + p]);
#pragma unroll
                for (int u = 0; u < kSelUnroll; ++u) {
                    const int dig = (int)((key[u] >> shift) & 15ull);
                    const unsigned long long hi = shift == 60 ? 0ull : key[u] >> (shift + 4);
#pragma unroll
                    for (int r = 0; r < kSelRanks; ++r) {
                        const unsigned long long want = shift == 60 ? 0ull : prefix[r] >> (shift + 4);
                        if (hi == want) cnt[r][dig][g][lane] += 1;
                    }
                }
            }
            for (; k < n_mod; k += kSelRG) {
                const unsigned long long key = sel_key(x[k * ld + p]);
                const int dig = (int)((key >> shift) & 15ull);
                const unsigned long long hi = shift == 60 ? 0ull : key >> (shift + 4);
#pragma unroll
                for (int r = 0; r < kSelRanks; ++r) {
                    const unsigned long long want = shift == 60 ? 0ull : prefix[r] >> (shift + 4);
                    if (hi == want) cnt[r][dig][g][lane] += 1;
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < kSelRanks; ++r) {
            long c = 0;
            int pick = 15;
            bool found = false;
            for (int d = 0; d < 16; ++d) {
                long tot = 0;
#pragma unroll
                for (int gg = 0; gg < kSelRG; ++gg) tot += cnt[r][d][gg][lane];
                if (!found && remaining[r] < c + tot) { pick = d; found = true; }
                if (!found) c += tot;
            }
            prefix[r] |= (unsigned long long)pick << shift;
            remaining[r] -= c;
        }
        __syncthreads();
    }
    if (live && g == 0) {
#pragma unroll
        for (int r = 0; r < kSelRanks; ++r) out[p * kSelRanks + r] = sel_val(prefix[r]);
    }
}

// ---- the same select for large sample sets: row slabs over many workgroups, one launch per digit --------------------
// k_select keeps a column group on one workgroup (49 workgroups for 3 130 parameters: a fifth of the CUs, four waves
// each).  Here the rows are cut into slabs as well, grid = (column groups, slabs); a pass counts its slab into private
// LDS counters as above and adds the 48 totals per column to a global histogram (integer atomics: exact, order-free);
// the NEXT launch starts by resolving the previous digit from that histogram -- every workgroup of a column group does
// the same small walk, slab 0 writes the new prefix/remaining state and clears the histogram after next.  Three
// histogram buffers and two state buffers go round; a last, count-free launch writes the values.
struct SelWork {
    int *hist[3];                       // [column group][rank][digit][lane]
    unsigned long long *prefix[2];      // [n_par][rank]
    long *remaining[2];
};
constexpr size_t kSelHistPerGroup = (size_t)kSelRanks * 16 * 64;

__global__ __launch_bounds__(64 * kSelRG) void k_select_pass(const double *x, long n_mod, long n_par, long ld, int r0, int r1,
                                                             int r2, int shift, int pass, long slab_rows, SelWork w,
                                                             double *out)
{
    __shared__ int cnt[kSelRanks][16][kSelRG][64];
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const long p = (long)blockIdx.x * 64 + lane;
    const bool live = p < n_par;
    const int *h_prev = w.hist[(pass + 2) % 3] + (size_t)blockIdx.x * kSelHistPerGroup;
    int *h_cur = w.hist[pass % 3] + (size_t)blockIdx.x * kSelHistPerGroup;
    int *h_next = w.hist[(pass + 1) % 3] + (size_t)blockIdx.x * kSelHistPerGroup;
    const unsigned long long *pre_old = w.prefix[(pass + 1) & 1];
    const long *rem_old = w.remaining[(pass + 1) & 1];
    unsigned long long prefix[kSelRanks] = {0ull, 0ull, 0ull};
    long remaining[kSelRanks] = {r0, r1, r2};
    if (pass > 0 && live) {             // resolve the digit the previous launch counted (at shift + 4)
#pragma unroll
        for (int r = 0; r < kSelRanks; ++r) {
            const long rem = rem_old[p * kSelRanks + r];
            long c = 0;
            int pick = 15;
            bool found = false;
            for (int d = 0; d < 16; ++d) {
                const long tot = h_prev[((size_t)r * 16 + d) * 64 + lane];
                if (!found && rem < c + tot) { pick = d; found = true; }
                if (!found) c += tot;
            }
            prefix[r] = pre_old[p * kSelRanks + r] | ((unsigned long long)pick << (shift + 4));
            remaining[r] = rem - c;
        }
    }
    if (blockIdx.y == 0 && g == 0 && live) {
#pragma unroll
        for (int r = 0; r < kSelRanks; ++r) {
            w.prefix[pass & 1][p * kSelRanks + r] = prefix[r];
            w.remaining[pass & 1][p * kSelRanks + r] = remaining[r];
            if (out) out[p * kSelRanks + r] = sel_val(prefix[r]);
        }
    }
    if (out) return;                    // the last launch only resolves
    if (blockIdx.y == 0)
        for (int k = threadIdx.x; k < (int)kSelHistPerGroup; k += blockDim.x) h_next[k] = 0;
#pragma unroll
    for (int r = 0; r < kSelRanks; ++r)
#pragma unroll
        for (int d = 0; d < 16; ++d) cnt[r][d][g][lane] = 0;
    const long row0 = (long)blockIdx.y * slab_rows, row1 = min(n_mod, row0 + slab_rows);
    if (live) {
        long k = row0 + g;
        for (; k + (kSelUnroll - 1) * kSelRG < row1; k += kSelUnroll * kSelRG) {
            unsigned long long key[kSelUnroll];
#pragma unroll
            for (int u = 0; u < kSelUnroll; ++u) key[u] = sel_key(x[(k + u * kSelRG) * ld + p]);
#pragma unroll
            for (int u = 0; u < kSelUnroll; ++u) {
                const int dig = (int)((key[u] >> shift) & 15ull);
                const unsigned long long hi = shift == 60 ? 0ull : key[u] >> (shift + 4);
#pragma unroll
                for (int r = 0; r < kSelRanks; ++r) {
                    const unsigned long long want = shift == 60 ? 0ull : prefix[r] >> (shift + 4);
                    if (hi == want) cnt[r][dig][g][lane] += 1;
                }
            }
        }
        for (; k < row1; k += kSelRG) {
            const unsigned long long key = sel_key(x[k * ld + p]);
            const int dig = (int)((key >> shift) & 15ull);
            const unsigned long long hi = shift == 60 ? 0ull : key >> (shift + 4);
#pragma unroll
            for (int r = 0; r < kSelRanks; ++r) {
                const unsigned long long want = shift == 60 ? 0ull : prefix[r] >> (shift + 4);
                if (hi == want) cnt[r][dig][g][lane] += 1;
            }
        }
    }
    __syncthreads();
    if (live) {
        // wave g adds the totals of digits 4g .. 4g+3 (all ranks) to the column group's histogram
#pragma unroll
        for (int r = 0; r < kSelRanks; ++r)
#pragma unroll
            for (int dd = 0; dd < 16 / kSelRG; ++dd) {
                const int d = g * (16 / kSelRG) + dd;
                int tot = 0;
#pragma unroll
                for (int gg = 0; gg < kSelRG; ++gg) tot += cnt[r][d][gg][lane];
                if (tot) atomicAdd(&h_cur[((size_t)r * 16 + d) * 64 + lane], tot);
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Step 4 (`hypo_tremor_select`): per detected window the two weighted linear regressions against the
// distance from the station of largest amplitude (src/cls_selector.f90:75-132, src/mod_regress.f90:5-58).
// wave <-> window, lane <-> station (strided beyond 64): one pass for the ten weighted sums of the two regressions,
// one for the six (unweighted, sic: mod_regress.f90:51-53) centred sums of the two correlation coefficients; distances
// and the spreading-corrected amplitude are recomputed per pass (three subtractions, a sqrt, a log) rather than kept.
// out[win] = {vs, b, t0, a0, cc_t, cc_a}, the columns of a regress.dat row after the window id.
__global__ __launch_bounds__(256) void k_regress(int S, int W, const double *sx, const double *sy, const double *sz,
                                                 double z_guess, const double *t, const double *t_err, const double *a,
                                                 const double *a_err, double *out)
{
    const int lane = threadIdx.x & 63;
    const int win = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (win >= W) return;
    const size_t base = (size_t)win * S;
    // maxloc(a): the FIRST maximum (cls_selector.f90:99)
    double best = -1.0e300;
    int near = 0x7fffffff;
    for (int j = lane; j < S; j += 64) {
        const double v = a[base + j];
        if (v > best) { best = v; near = j; }
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
        const double ov = __shfl_xor(best, m);
        const int oi = __shfl_xor(near, m);
        if (ov > best || (ov == best && oi < near)) { best = ov; near = oi; }
    }
    const double nx = sx[near], ny = sy[near];
    double r[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) r[k] = 0.0;
    for (int j = lane; j < S; j += 64) {
        const double dx = sx[j] - nx, dy = sy[j] - ny, dz = sz[j] - z_guess;      // :62-64
        const double d = sqrt(dx * dx + dy * dy + dz * dz);
        const double tj = t[base + j], aj = a[base + j] + log(d);                 // :102-103
        const double wt = 1.0 / (t_err[base + j] * t_err[base + j]), wa = 1.0 / (a_err[base + j] * a_err[base + j]);   // :114, :120
        r[0] += d * wt; r[1] += tj * wt; r[2] += wt; r[3] += d * tj * wt; r[4] += d * d * wt;      // mod_regress.f90:18-24
        r[5] += d * wa; r[6] += aj * wa; r[7] += wa; r[8] += d * aj * wa; r[9] += d * d * wa;
    }
    wave_sum<10>(r);
    const double det_t = r[2] * r[4] - r[0] * r[0], det_a = r[7] * r[9] - r[5] * r[5];            // :26
    const double slope_t = (r[2] * r[3] - r[0] * r[1]) / det_t, icpt_t = (r[4] * r[1] - r[0] * r[3]) / det_t;
    const double slope_a = (r[7] * r[8] - r[5] * r[6]) / det_a, icpt_a = (r[9] * r[6] - r[5] * r[8]) / det_a;
    const double mx_t = r[0] / r[2], my_t = r[1] / r[2], mx_a = r[5] / r[7], my_a = r[6] / r[7];    // :48-49
    double c[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) c[k] = 0.0;
    for (int j = lane; j < S; j += 64) {
        const double dx = sx[j] - nx, dy = sy[j] - ny, dz = sz[j] - z_guess;
        const double d = sqrt(dx * dx + dy * dy + dz * dz);
        const double tj = t[base + j], aj = a[base + j] + log(d);
        c[0] += (d - mx_t) * (d - mx_t); c[1] += (tj - my_t) * (tj - my_t); c[2] += (d - mx_t) * (tj - my_t);      // :51-53
        c[3] += (d - mx_a) * (d - mx_a); c[4] += (aj - my_a) * (aj - my_a); c[5] += (d - mx_a) * (aj - my_a);
    }
    wave_sum<6>(c);
    if (lane == 0) {
        double *o = out + (size_t)win * 6;
        o[0] = 1.0 / slope_t;                 // vs   (cls_selector.f90:117)
        o[1] = -1.0 * slope_a;                // b    (:123)
        o[2] = icpt_t;                        // t0
        o[3] = icpt_a;                        // a0
        o[4] = c[2] / sqrt(c[0] * c[1]);      // cc_t (mod_regress.f90:55)
        o[5] = c[5] / sqrt(c[3] * c[4]);      // cc_a
    }
}

}  // namespace htm
