// htm_stream.hpp -- producers of the rank's random stream (run on a side stream, ahead of k_step).
//
// mod_random (reference src/mod_random.f90) is one serial xorshift128 stream per rank, consumed in an order
// that depends on chain state only through HOW MANY draws each chain step takes.  The values themselves,
// and everything derived from a value at a given stream position, are independent of the chains.  So the
// stream is produced ahead of time:
//   k_rawgen       the recurrence itself, in parallel: xorshift128 is GF(2)-linear, so every lane jumps to the start
//                  of its own 64-draw segment with precomputed powers of the transition matrix
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

// ---- the recurrence itself, in parallel by jump-ahead ---------------------------------------------------------
// xorshift128 (mod_random.f90:63-71) is linear over GF(2): one step is state' = T * state for a fixed 128 x 128 bit
// matrix T (state = x | y << 32 | z << 64 | w << 96).  The host precomputes J[b] = T^(64 * 2^b) (htm_jump_table),
// stored as 128 columns of 4 words.  Segment g of 64 draws starts from  T^(64 g) * s0 = prod_{bits b of g} J[b] * s0,
// so every LANE can start its own segment: a wave produces 64 segments = 4 096 consecutive draws, transposes them
// through LDS and writes them with coalesced 256-B stores.  Bit-identical to the serial stream by construction.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));      // one matrix column
constexpr int kJumpLevels = 20;       // segments per call < 2^20 (ring capacity <= 2^24 positions)

__device__ __forceinline__ void jump_apply(uint32_t (&s)[4], const u32x4 *J)
{
    uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0;
#pragma unroll
    for (int wd = 0; wd < 4; ++wd) {
        const uint32_t v = s[wd];
#pragma unroll 8
        for (int bit = 0; bit < 32; ++bit) {
            const u32x4 col = ld_const(J + wd * 32 + bit);           // wave-uniform: scalar load
            const uint32_t m = 0u - ((v >> bit) & 1u);
            a0 ^= col.x & m; a1 ^= col.y & m; a2 ^= col.z & m; a3 ^= col.w & m;
        }
    }
    s[0] = a0; s[1] = a1; s[2] = a2; s[3] = a3;
}

// grid = ceil(n / 4096) workgroups of ONE wave; n a multiple of 64.  gen_in: state after the last produced draw
// (read by every wave); gen_out: the state after this call's last draw (a different buffer: no race with the readers).
__global__ __launch_bounds__(64) void k_rawgen(StreamDev sd, long long start, int n, const u32x4 *jump,
                                               const uint32_t *gen_in, uint32_t *gen_out)
{
    __shared__ uint32_t tile[64 * 65];
    const int lane = threadIdx.x;
    const int n_seg = n >> 6;
    const int g = blockIdx.x * 64 + lane;                 // this lane's segment
    uint32_t s[4] = {gen_in[0], gen_in[1], gen_in[2], gen_in[3]};
    // bits 6.. of g are uniform over the wave (scalar branch), bits 0..5 differ by lane (predicated)
    for (int b = 6; b < kJumpLevels; ++b)
        if ((blockIdx.x >> (b - 6)) & 1) jump_apply(s, jump + (size_t)b * 128);
    for (int b = 0; b < 6; ++b) {
        uint32_t t[4] = {s[0], s[1], s[2], s[3]};
        jump_apply(t, jump + (size_t)b * 128);
        if ((lane >> b) & 1) { s[0] = t[0]; s[1] = t[1]; s[2] = t[2]; s[3] = t[3]; }
    }
    uint32_t x = s[0], y = s[1], z = s[2], w = s[3];
#pragma unroll
    for (int k = 0; k < 64; ++k) tile[lane * 65 + k] = xs128_next(x, y, z, w);
    if (g == n_seg - 1) { gen_out[0] = x; gen_out[1] = y; gen_out[2] = z; gen_out[3] = w; }
    __syncthreads();
    const int seg0 = blockIdx.x * 64;
    for (int k = 0; k < 64 && seg0 + k < n_seg; ++k)
        sd.raw[(start + (long long)(seg0 + k) * 64 + lane) & sd.mask] = tile[k * 65 + lane];
}

// the same stream drawn serially by one lane (htm_selftest compares the two)
__global__ void k_rawgen_serial(uint32_t *out, int n, const uint32_t *gen_in, uint32_t *gen_out)
{
    uint32_t x = gen_in[0], y = gen_in[1], z = gen_in[2], w = gen_in[3];
    for (int k = 0; k < n; ++k) out[k] = xs128_next(x, y, z, w);
    gen_out[0] = x; gen_out[1] = y; gen_out[2] = z; gen_out[3] = w;
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
