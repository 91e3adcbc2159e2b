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

}  // namespace htm
