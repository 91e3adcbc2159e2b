// htm_stream.hpp -- producers of the rank's random stream (run on a side stream, ahead of k_step).
//
// mod_random (reference src/mod_random.f90) is one serial xorshift128 stream per rank, consumed in an order
// that depends on chain state only through HOW MANY draws each chain step takes.  The values themselves,
// and everything derived from a value at a given stream position, are independent of the chains.  So the
// stream is produced ahead of time:
//   k_rawgen       one wavefront, the recurrence itself on the scalar ALU (serial by nature), 64 draws per
//                  coalesced store
//   k_stream_tr    per position: rand_u, log(rand_u), the Box-Muller value that starts there
//   k_stream_rec   per position: the chain step that would START there, decoded (proposal type, element,
//                  event, draws consumed) with its Gaussian and its Metropolis draw
//   k_stream_hop   per position: where the 1st..8th following chain step starts (optimistic)
// k_step only copies a window of these rings into LDS and follows them.
#pragma once
#include "htm_device.hpp"

namespace htm {

constexpr int kHops = 8;          // hop tables cover 1..8 chain steps (k_step has at most 8 chain waves)
constexpr int kRecLag = 16;       // a record at p reads transforms up to p+5, a swap plan up to p+13
constexpr int kHopLag = 6 * kHops;

__global__ __launch_bounds__(64) void k_rawgen(StreamDev sd, long long start, int n)
{
    const int lane = threadIdx.x;
    uint32_t x = __builtin_amdgcn_readfirstlane(sd.gen[0]), y = __builtin_amdgcn_readfirstlane(sd.gen[1]);
    uint32_t z = __builtin_amdgcn_readfirstlane(sd.gen[2]), w = __builtin_amdgcn_readfirstlane(sd.gen[3]);
    for (int blk = 0; blk < n; blk += 64) {
        uint32_t mine = 0;
#pragma unroll 16
        for (int k = 0; k < 64; ++k) {
            const uint32_t r = xs128_next(x, y, z, w);
            mine = (lane == k) ? r : mine;
        }
        sd.raw[(start + blk + lane) & sd.mask] = mine;
    }
    if (lane == 0) { sd.gen[0] = x; sd.gen[1] = y; sd.gen[2] = z; sd.gen[3] = w; }
}

__global__ __launch_bounds__(256) void k_stream_tr(StreamDev sd, long long start, long long end)
{
    const long long p = start + (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= end) return;
    const uint32_t r0 = sd.raw[p & sd.mask], r1 = sd.raw[(p + 1) & sd.mask];
    const double u = u_of(r0);
    sd.U[p & sd.mask] = u;
    sd.LOGU[p & sd.mask] = log(u);
    sd.G[p & sd.mask] = g_of(r0, r1);
}

// cls_mcmc.f90:134-165: a_select, then (id,) (icmp,) then the two draws of rand_g, then the judge's rand_u
__global__ __launch_bounds__(256) void k_stream_rec(StreamDev sd, long long start, long long end, double th1,
                                                    double th2, double th3, double th4, int S, int E,
                                                    int n_procs, int n_chains)
{
    const long long p = start + (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= end) return;
    const long long M = sd.mask;
    const double a = sd.U[p & M], u1 = sd.U[(p + 1) & M], u2 = sd.U[(p + 2) & M];
    int type, idx, evt = -999, goff;
    if (a < th1) { type = 1; idx = 0; goff = 1; }
    else if (a < th2) { type = 2; idx = (int)(u1 * S); goff = 2; }
    else if (a < th3) { type = 3; idx = 0; goff = 1; }
    else if (a < th4) { type = 4; idx = (int)(u1 * S); goff = 2; }
    else {
        const int id = (int)(u1 * E) + 1;
        const int icmp = (int)(u2 * 3);
        idx = 3 * id - icmp - 1; type = 5 + icmp; evt = id; goff = 3;
    }
    const long long gpos = p + goff, jpos = gpos + 2;
    sd.dec[p & M] = make_int4(type, idx, evt, goff + 3);     // draws if prior_ok: ..., g(2), r
    sd.pg[p & M] = sd.G[gpos & M];
    sd.pr[p & M] = sd.U[jpos & M];
    sd.plogr[p & M] = sd.LOGU[jpos & M];
    // select_pair (cls_parallel.f90:226-230) if it started at p: i1, then i2 redrawn until it differs
    int i1 = -1, i2 = -1, used = -1;
    if (n_procs * n_chains > 1) {
        i1 = (int)(a * n_procs * n_chains);
        for (int k = 1; k <= 12; ++k) {
            i2 = (int)(sd.U[(p + k) & M] * n_procs * n_chains);
            if (i2 != i1) { used = k + 1; break; }
        }
    }
    sd.sw[p & M] = make_int4(i1, i2, used, 0);
}

__global__ __launch_bounds__(256) void k_stream_hop(StreamDev sd, long long start, long long end)
{
    const long long p = start + (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= end) return;
    long long h = p;
#pragma unroll
    for (int k = 0; k < kHops; ++k) {
        h += sd.dec[h & sd.mask].w;
        sd.hop[(p & sd.mask) * kHops + k] = (int)(h - p);      // stored relative to p
    }
}

__global__ void k_publish(long long *dst, long long v) { *dst = v; }

}  // namespace htm
