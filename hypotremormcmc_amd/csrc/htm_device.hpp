// htm_device.hpp -- gfx950 device code for the HypoTremorMCMC likelihood inner loop.
//
// Written for CDNA4 only: 64-wide wavefronts, DPP cross-lane reductions (row_bcast forms of the GFX9 DPP
// encoding), fp64 VALU.  No MFMA: the path is elementwise geometry + reductions.
//
// Mapping: lane <-> station, wave <-> event (full evaluation) or wave <-> chain (partial
// update).  The station table (3 S doubles) is staged in LDS by the chain master's workgroup and copied from there into
// each wave's registers; the full-evaluation kernels (k_full, the worker blocks) read it straight into registers once per
// wave -- one coalesced load per coordinate that L2 serves, a second LDS hop would only add latency (a stated
// deviation from "LDS staging of the station table").  The four observation streams t_obs/t_prec/a_obs/a_prec are read with
// coalesced 512-B wave loads in the reference's own (n_sta, n_events) column-major layout.
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace htm {

constexpr double kPi = 3.141592653589793;        // acos(-1.d0)
constexpr double kPi2 = 6.283185307179586;       // 2.d0 * acos(-1.d0)   (mod_random.f90:33)
constexpr double kFreq = 5.0;                    // cls_forward.f90:190
constexpr double kEps = 2.220446049250313e-16;   // epsilon(1.d0)
constexpr int kMaxChains = 32;                   // chains per rank held by one k_step workgroup

// ---------------------------------------------------------------------------------------------------
// wave-level fp64 sum over 64 lanes, DPP only (no LDS traffic, fixed order => deterministic)
// ---------------------------------------------------------------------------------------------------
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_mov_f64(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    // every lane written (full row mask, in-range source): bound_ctrl tells the compiler the old value is dead,
    // which saves the two zero-initialising moves per step; the row_bcast steps need old = 0 in the rows they skip
    constexpr bool kAll = ROW_MASK == 0xF;
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xF, kAll);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xF, kAll);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double readlane63_f64(double v)
{
    int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}

#ifndef HTM_MFMA_SUM
#define HTM_MFMA_SUM 0
#endif
typedef double f64x4 __attribute__((ext_vector_type(4)));

// The same sum on the matrix pipe: two v_mfma_f64_16x16x4_f64 against a matrix of ones.  Lane l holds A[l & 15][l >> 4], so
// the first product gives S_i = x_i + x_{i+16} + x_{i+32} + x_{i+48} for every row i; a lane of group g = l >> 4 receives the
// rows g, g + 4, g + 8, g + 12 in its four result registers; their sum G_g is that lane's element of the second product's A,
// whose every result element is G_0 + G_1 + G_2 + G_3.  Two matrix instructions and three additions per value in place of
// twelve DPP moves, six additions and two v_readlane: 7 issue slots for 20.  Fixed order, every lane gets the sum.
__device__ __forceinline__ double wave_sum_mfma(double x)
{
    const f64x4 z = {0.0, 0.0, 0.0, 0.0};
    const f64x4 d = __builtin_amdgcn_mfma_f64_16x16x4f64(x, 1.0, z, 0, 0, 0);
    const double g = (d[0] + d[1]) + (d[2] + d[3]);
    const f64x4 e = __builtin_amdgcn_mfma_f64_16x16x4f64(g, 1.0, z, 0, 0, 0);
    return e[0];
}

#ifndef HTM_TSUM
#define HTM_TSUM 1
#endif
// x + x of the lane 16 (32) places away, in every lane: v_permlane16_swap (v_permlane32_swap) on two copies of x leaves the
// even (lower) rows in one and the odd (upper) rows in the other.
__device__ __forceinline__ double swap16_sum(double x)
{
    const unsigned lo = (unsigned)__double2loint(x), hi = (unsigned)__double2hiint(x);
    const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
}
__device__ __forceinline__ double swap32_sum(double x)
{
    const unsigned lo = (unsigned)__double2loint(x), hi = (unsigned)__double2hiint(x);
    const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
}

// Two or four sums at once, TRANSPOSED: the first stage (lane <-> lane ^ 1) leaves each lane with the pair sum of ONE value of
// every two, the second (lane ^ 2, four values) with the quad sum of one of four; from there a single register per lane runs
// through the remaining stages instead of N.  Which value a lane keeps (its "class") is chosen so that the mirror stages pair
// lanes of the same class: class = lane & 3 in the even quads of a row, 3 - (lane & 3) in the odd ones (row_half_mirror and
// row_mirror reverse the lane order within 8 and 16); the stages across rows use the lane-preserving swaps above.  Every
// value's additions are the balanced tree over the lanes in natural order -- the same operands at every node as in the plain
// form below, so the same bits -- in 30 instructions for two values (44) and 45 for four (88).  Lanes 0..N-1 hold classes
// 0..N-1 at the end.
template <int N>
__device__ __forceinline__ void wave_sum_transposed(double (&v)[N])
{
    static_assert(N == 2 || N == 4, "two or four values");
    const unsigned lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const bool b0 = ((lane ^ (lane >> 2)) & 1u) != 0u;
    double x;
    {
        const double keep = b0 ? v[1] : v[0], send = b0 ? v[0] : v[1];
        x = keep + dpp_mov_f64<0xB1, 0xF>(send);                       // quad_perm [1,0,3,2]
    }
    if constexpr (N == 4) {
        const bool b1 = (((lane >> 1) ^ (lane >> 2)) & 1u) != 0u;
        const double keep2 = b0 ? v[3] : v[2], send2 = b0 ? v[2] : v[3];
        const double y = keep2 + dpp_mov_f64<0xB1, 0xF>(send2);
        const double keep = b1 ? y : x, send = b1 ? x : y;
        x = keep + dpp_mov_f64<0x4E, 0xF>(send);                       // quad_perm [2,3,0,1]
    } else {
        x += dpp_mov_f64<0x4E, 0xF>(x);
    }
    x += dpp_mov_f64<0x141, 0xF>(x);                                   // row_half_mirror
    x += dpp_mov_f64<0x140, 0xF>(x);                                   // row_mirror
    x = swap16_sum(x);
    x = swap32_sum(x);
    const int lo = __double2loint(x), hi = __double2hiint(x);
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = __hiloint2double(__builtin_amdgcn_readlane(hi, k), __builtin_amdgcn_readlane(lo, k));
}

// N independent sums, interleaved step by step so that the DPP latencies overlap.
// Must be called with all 64 lanes active; inactive stations contribute 0.
template <int N>
__device__ __forceinline__ void wave_sum(double (&v)[N])
{
    if constexpr (HTM_TSUM != 0 && HTM_MFMA_SUM == 0 && (N == 2 || N == 4)) { wave_sum_transposed<N>(v); return; }
    if constexpr (HTM_MFMA_SUM != 0) {
        const f64x4 z = {0.0, 0.0, 0.0, 0.0};
        double g[N];
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const f64x4 d = __builtin_amdgcn_mfma_f64_16x16x4f64(v[k], 1.0, z, 0, 0, 0);
            g[k] = (d[0] + d[1]) + (d[2] + d[3]);
        }
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const f64x4 e = __builtin_amdgcn_mfma_f64_16x16x4f64(g[k], 1.0, z, 0, 0, 0);
            v[k] = e[0];
        }
        return;
    }
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] += dpp_mov_f64<0xB1, 0xF>(v[k]);   // quad_perm [1,0,3,2]
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] += dpp_mov_f64<0x4E, 0xF>(v[k]);   // quad_perm [2,3,0,1]
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] += dpp_mov_f64<0x141, 0xF>(v[k]);  // row_half_mirror
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] += dpp_mov_f64<0x140, 0xF>(v[k]);  // row_mirror
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] += dpp_mov_f64<0x142, 0xA>(v[k]);  // row_bcast:15 -> rows 1,3
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] += dpp_mov_f64<0x143, 0xC>(v[k]);  // row_bcast:31 -> rows 2,3
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = readlane63_f64(v[k]);
}

__device__ __forceinline__ double wave_sum1(double a)
{
    double v[1] = {a};
    wave_sum<1>(v);
    return v[0];
}

// ---------------------------------------------------------------------------------------------------
// Natural logarithm of the forward model's amplitude term (cls_forward.f90:204, `log(d)`), fp64.
// The device library's `log` keeps its intermediate sums in double-double form (~88 VALU instructions, half of one
// station's evaluation); this one is 31: x = m * 2^e with m in [sqrt(1/2), sqrt(2)), f = m - 1, s = f / (2 + f),
// log(m) = f - (f^2/2 - s * (f^2/2 + R(s^2))), R = z * P(z) with P a degree-6 near-minimax fit of
// (log((1+s)/(1-s)) - 2s) / s^3 on [0, (3 - 2 sqrt 2)^2] (coefficients: tools/log_coefficients.py, |error| 3.1e-16),
// result = e * ln2_hi - ((f^2/2 - (s * (f^2/2 + R) + e * ln2_lo)) - f) with a 32-bit ln2_hi so that e * ln2_hi is
// exact.  Leading term f is exact, s enters only through a term <= 4 % of the result: measured error < 0.75 ulp on the
// GPU (tests/test_gpu_forward.py: 1.3 M arguments against an 80-bit logarithm, htm_selftest_math).  x = 0 -> -inf, NaN -> NaN, subnormals
// are handled by v_frexp; x < 0 and +inf (never produced by a distance) give NaN.  Every operation is an explicit
// fma / mul / add: nothing is left to the compiler's contraction rules, the value is the same in every kernel.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ double htm_log(double x)
{
    const double m0 = __builtin_amdgcn_frexp_mant(x);                 // [0.5, 1)
    const int lo = m0 < 0.70710678118654752 ? 1 : 0;
    const double f = __builtin_fma(m0, lo ? 2.0 : 1.0, -1.0);         // m - 1, exact
    const double k = (double)(__builtin_amdgcn_frexp_exp(x) - lo);
    const double D = f + 2.0;
    double r = __builtin_amdgcn_rcp(D);
    r = __builtin_fma(__builtin_fma(-D, r, 1.0), r, r);
    double s = f * r;
    s = __builtin_fma(__builtin_fma(-s, D, f), r, s);                 // f / D, correctly rounded but for rare cases
    const double z = s * s;
    double p = __builtin_fma(z, 0x1.2b5900de53b32p-3, 0x1.39fe51a7c18f9p-3);
    p = __builtin_fma(z, p, 0x1.7462b51cb66b1p-3);
    p = __builtin_fma(z, p, 0x1.c71c62e3f11e6p-3);
    p = __builtin_fma(z, p, 0x1.2492492df281ap-2);
    p = __builtin_fma(z, p, 0x1.99999999952d7p-2);
    p = __builtin_fma(z, p, 0x1.5555555555558p-1);
    const double hf = 0.5 * f;
    const double a = __builtin_fma(f, hf, z * p);                     // f^2/2 + R
    const double t = __builtin_fma(s, a, k * 0x1.a39ef35793c76p-33);  // + e * ln2_lo
    const double b = __builtin_fma(f, hf, -t);
    const double y = __builtin_fma(k, 0x1.62e42fee00000p-1, -(b - f));
    return x == 0.0 ? -__builtin_inf() : y;
}

// Square root of the squared distance (cls_forward.f90:115-117, :201-203), fp64: the device library's iteration
// (v_rsq_f64 seed, one coupled Goldschmidt step, two residual corrections) without its rescaling of arguments below
// 2^-767 and its class test -- a squared distance in km^2 is never there.  Same value as `sqrt` for every argument in
// [2^-767, inf) (tests/test_gpu_forward.py); 0 -> 0.
__device__ __forceinline__ double htm_sqrt(double x)
{
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    g = __builtin_fma(__builtin_fma(-g, g, x), h, g);
    g = __builtin_fma(__builtin_fma(-g, g, x), h, g);
    return x == 0.0 ? 0.0 : g;
}

// ---------------------------------------------------------------------------------------------------
// mod_random on the device (reference src/mod_random.f90:60-112)
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t xs128_next(uint32_t &x, uint32_t &y, uint32_t &z, uint32_t &w)
{
    uint32_t t = x ^ (x << 11);
    x = y; y = z; z = w;
    w = (w ^ (w >> 19)) ^ (t ^ (t >> 8));
    return w;
}
__device__ __forceinline__ double u_of(uint32_t raw)   // rand_u  :72
{
    return ((double)(int32_t)raw + 2147483648.0) / 4294967296.0;
}
__device__ __forceinline__ double u2_of(uint32_t raw)  // rand_u2 :90
{
    return ((double)(int32_t)raw + 2147483648.0 + 0.5) / 4294967296.0;
}
__device__ __forceinline__ double g_of(uint32_t raw1, uint32_t raw2)  // rand_g :98-100
{
    return sqrt(-2.0 * log(u2_of(raw1))) * cos(kPi2 * u2_of(raw2));
}

// ---------------------------------------------------------------------------------------------------
// forward model (reference src/cls_forward.f90)
// ---------------------------------------------------------------------------------------------------
struct FwdDev {
    int S, E, use_time, use_amp;
    const double *sx, *sy, *sz;                        // [S]
    const double *t_obs, *t_prec, *a_obs, *a_prec;     // [E][S]  (station fastest)
    const double *psum_t, *psum_a;                     // [E]  sum_j prec(j,i), summed in station order
    const double *rpsum_t, *rpsum_a;                   // [E]  their reciprocals (host-computed): the demean is a multiplication
    double const_sum;   // sum over used data types of sum_{j,i} (log_2pi_half + log_stdv(j,i))
    // fp32 forward / fp64 accept (BASELINE configs[4], htm_forward_set_precision): the four observation streams once
    // more as float (half the bytes of a full evaluation), and the flag that makes event_misfit compute the synthetic
    // travel time / amplitude (distance, sqrt, division, log: cls_forward.f90:115-118, :201-204) in single precision.
    // Demean sums, residuals, the misfit sum and the Metropolis decision stay fp64 (:125-132, :281-299).
    const float *t_obs32, *t_prec32, *a_obs32, *a_prec32;
    int fp32;           // host-side dispatch only: the F32 = true instantiations of the kernels are launched
};

// Immutable inputs (priors, step sizes, precision sums) at a wave-uniform address: a load through the constant
// address space is selected as a SCALAR load (K$), which costs no vector-memory slot and no VGPR.  Only for data
// no kernel ever writes.
template <class T>
__device__ __forceinline__ T ld_const(const T *p)
{
    typedef const T __attribute__((address_space(4))) *CP;
    return *(CP)(unsigned long long)p;
}
template <int NCH>
struct StaRegs {
    double sx[NCH], sy[NCH], sz[NCH], tc[NCH], ac[NCH];
};

// Evaluate NPOS candidate hypocentres of ONE event against all stations of this wave's lanes.
// Returns per lane: sum over its stations of [time misfit + amp misfit] for each position, where
// misfit = (obs - (syn - mean))^2 * prec / 2 -- i.e. cls_forward.f90:115-118,:125-132,:283-285 (time) and
// :201-204,:210-217,:294-296 (amplitude).  The per-event weighted means need one wave reduction per data
// type and position; those are issued together.  All 64 lanes must call.
// one event's observations for this lane's stations (the four coalesced HBM streams of the path)
// (fp32 forward: the float streams stay floats here -- half the registers of a worker's look-ahead buffers -- and are promoted,
// exactly, where the evaluation reads them.  Promoting at the load made every pair of loads wait for itself (the conversion
// needs the value): four waits of a memory round trip each in front of a chain step, 3.7 k cycles at 10 000 x 128.)
template <int NCH, bool F32 = false>
struct ObsRegs {
    typedef typename std::conditional<F32, float, double>::type T;
    T tob[NCH], tpr[NCH], aob[NCH], apr[NCH];
    double rpst, rpsa;   // 1 / sum_j precision(j, event), time and amplitude
};

#ifndef HTM_VRPS_WORKERS
#define HTM_VRPS_WORKERS 0   // measured: -6 % at 10 000 x 128 (vector loads of a uniform address cost more than the scalar-cache traffic they avoid)
#endif
#ifndef HTM_NT_WORKERS
#define HTM_NT_WORKERS 0   // 1: the workers read the observation rows with the non-temporal hint -- measured: -3 % (fp64) ... -15 % at 10 000 events (profiles/r03_p_nt_loads.txt): the rows are re-read from the Infinity Cache by every evaluation, the hint keeps them out of it
#endif
// NT: the rows are read once per full evaluation by a worker that streams many events (worker_body): loaded with the
// non-temporal hint they do not push the chain master's working set (chain state, priors, station table) out of its XCD's L2.
template <class T, bool NT>
__device__ __forceinline__ T ld_stream(const T *p)
{
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}
// VRPS: the event's two precision-sum reciprocals come with VECTOR loads (same address in every lane) instead of scalar ones:
// a worker that streams many events would otherwise run 2 x E scalar loads per full evaluation through the scalar cache it
// shares with neighbouring CUs -- the chain master's among them, whose kernel arguments and prior records live there.
__device__ __forceinline__ int lane_zero()
{
    int z;
    asm volatile("v_mov_b32 %0, 0" : "=v"(z));
    return z;
}
template <int NCH, bool F32 = false, bool NT = false, bool VRPS = false, class FW>
__device__ __forceinline__ void load_obs_regs(ObsRegs<NCH, F32> &ob, const FW &f, int ev, int lane)
{
    const size_t base = (size_t)ev * (size_t)f.S;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int j = lane + 64 * c;
        const bool valid = j < f.S;
        ob.tob[c] = ob.tpr[c] = ob.aob[c] = ob.apr[c] = 0;
        if (valid) {
            if constexpr (F32) {   // the float streams
                if (f.use_time) { ob.tob[c] = ld_stream<float, NT>(f.t_obs32 + base + j); ob.tpr[c] = ld_stream<float, NT>(f.t_prec32 + base + j); }
                if (f.use_amp)  { ob.aob[c] = ld_stream<float, NT>(f.a_obs32 + base + j); ob.apr[c] = ld_stream<float, NT>(f.a_prec32 + base + j); }
            } else {
                if (f.use_time) { ob.tob[c] = ld_stream<double, NT>(f.t_obs + base + j); ob.tpr[c] = ld_stream<double, NT>(f.t_prec + base + j); }
                if (f.use_amp)  { ob.aob[c] = ld_stream<double, NT>(f.a_obs + base + j); ob.apr[c] = ld_stream<double, NT>(f.a_prec + base + j); }
            }
        }
    }
    if constexpr (VRPS) {
        const int vz = lane_zero();
        ob.rpst = f.use_time ? f.rpsum_t[ev + vz] : 1.0;
        ob.rpsa = f.use_amp ? f.rpsum_a[ev + vz] : 1.0;
    } else {
        ob.rpst = f.use_time ? ld_const(f.rpsum_t + ev) : 1.0;    // wave-uniform: scalar loads
        ob.rpsa = f.use_amp ? ld_const(f.rpsum_a + ev) : 1.0;
    }
}

// The same rows, requested without a branch: every lane issues all four loads (a lane without a station reads the row's last
// entry and discards it; the arrays exist whatever use_time / use_amp say), so the number of loads per call is fixed and the
// compiler can leave several calls' worth in flight (s_waitcnt vmcnt(N > 0)) -- the worker blocks' event pipeline.
// FULL: every lane has a station in every chunk and both data types are used (the caller has checked: n_sta a multiple of 64,
// use_time and use_amp) -- no selects on the loaded values at all.
template <int NCH, bool F32 = false, bool NT = false, bool VRPS = false, bool FULL = false, class FW>
__device__ __forceinline__ void load_obs_regs_nobranch(ObsRegs<NCH, F32> &ob, const FW &f, int ev, int lane)
{
    const size_t base = (size_t)ev * (size_t)f.S;
    if constexpr (FULL) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const size_t k = base + (size_t)(lane + 64 * c);
            if constexpr (F32) {
                ob.tob[c] = ld_stream<float, NT>(f.t_obs32 + k); ob.tpr[c] = ld_stream<float, NT>(f.t_prec32 + k);
                ob.aob[c] = ld_stream<float, NT>(f.a_obs32 + k); ob.apr[c] = ld_stream<float, NT>(f.a_prec32 + k);
            } else {
                ob.tob[c] = ld_stream<double, NT>(f.t_obs + k); ob.tpr[c] = ld_stream<double, NT>(f.t_prec + k);
                ob.aob[c] = ld_stream<double, NT>(f.a_obs + k); ob.apr[c] = ld_stream<double, NT>(f.a_prec + k);
            }
        }
        ob.rpst = ld_const(f.rpsum_t + ev);
        ob.rpsa = ld_const(f.rpsum_a + ev);
        return;
    }
    const bool ut = f.use_time != 0, ua = f.use_amp != 0;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int j = lane + 64 * c;
        const bool valid = j < f.S;
        const size_t k = base + (size_t)(valid ? j : f.S - 1);
        typename ObsRegs<NCH, F32>::T t, tp, a, ap;
        if constexpr (F32) {
            t = ld_stream<float, NT>(f.t_obs32 + k); tp = ld_stream<float, NT>(f.t_prec32 + k);
            a = ld_stream<float, NT>(f.a_obs32 + k); ap = ld_stream<float, NT>(f.a_prec32 + k);
        } else {
            t = ld_stream<double, NT>(f.t_obs + k); tp = ld_stream<double, NT>(f.t_prec + k);
            a = ld_stream<double, NT>(f.a_obs + k); ap = ld_stream<double, NT>(f.a_prec + k);
        }
        ob.tob[c] = (valid && ut) ? t : 0; ob.tpr[c] = (valid && ut) ? tp : 0;
        ob.aob[c] = (valid && ua) ? a : 0; ob.apr[c] = (valid && ua) ? ap : 0;
    }
    if constexpr (VRPS) {
        const int vz = lane_zero();
        const double t = f.rpsum_t[ev + vz], a = f.rpsum_a[ev + vz];
        ob.rpst = ut ? t : 1.0; ob.rpsa = ua ? a : 1.0;
    } else {
        ob.rpst = ut ? ld_const(f.rpsum_t + ev) : 1.0;
        ob.rpsa = ua ? ld_const(f.rpsum_a + ev) : 1.0;
    }
}

// PRE: `beta` and `q` hold 1 / vs and pi f / (qs vs) already -- the caller formed them once (a worker per order, the chain master
// per commit of vs or qs): two fp64 divisions, ~28 instructions, that the compiler does not move out of a loop over events.
template <int NCH, int NPOS, bool F32 = false, bool PRE = false, class FW>
__device__ __forceinline__ void event_misfit(const FW &f, const ObsRegs<NCH, F32> &ob, int lane,
                                             const StaRegs<NCH> &st, const double (&px)[NPOS],
                                             const double (&py)[NPOS], const double (&pz)[NPOS], double beta,
                                             double q, double (&out)[NPOS])
{
    // Strength reduction (round 2): the reference divides per station -- d / beta, d * pi * freq / (q * beta)
    // (cls_forward.f90:118, :204) -- and per event mean (:131, :217).  beta, q and the precision sums are the same for all
    // stations of a call, so the divisions are done ONCE (two here, the sums' reciprocals on the host) and the per-station
    // work is multiplications: the values differ from the reference's in the last bit at most (the same class as the
    // libm `log` difference and the summation order), a third of the step's fp64 instructions go away.
    double rbeta, katt;
    if constexpr (PRE) { rbeta = beta; katt = q; }
    else { const double qbeta = q * beta; rbeta = 1.0 / beta; katt = (kPi * kFreq) / qbeta; }
    double tob[NCH], tpr[NCH], aob[NCH], apr[NCH];      // (promoted here in the fp32 mode: exact)
#pragma unroll
    for (int c = 0; c < NCH; ++c) { tob[c] = (double)ob.tob[c]; tpr[c] = (double)ob.tpr[c]; aob[c] = (double)ob.aob[c]; apr[c] = (double)ob.apr[c]; }
    double ts[NPOS][NCH], as[NPOS][NCH];
    double red[2 * NPOS];
#pragma unroll
    for (int k = 0; k < 2 * NPOS; ++k) red[k] = 0.0;
    if constexpr (F32) {
        // single-precision forward: the coordinate differences are formed in fp64 (they are what the model state is),
        // everything after them -- distance, travel time, amplitude -- is fp32 arithmetic on the hardware's own
        // sqrt / reciprocal / log2 (v_sqrt_f32, v_rcp_f32, v_log_f32: ~1 ulp), then promoted for the fp64 sums
        const float rbeta32 = (float)rbeta, katt32 = (float)katt;
#pragma unroll
        for (int p = 0; p < NPOS; ++p) {
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const bool valid = (lane + 64 * c) < f.S;
                const float dx = (float)(px[p] - st.sx[c]), dy = (float)(py[p] - st.sy[c]), dz = (float)(pz[p] - st.sz[c]);
                float d = __builtin_amdgcn_sqrtf(__builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx)));
                if (!valid) d = 1.0f;
                ts[p][c] = 0.0; as[p][c] = 0.0;
                if (f.use_time) {
                    ts[p][c] = (double)__builtin_fmaf(d, rbeta32, -(float)st.tc[c]);
                    red[2 * p] = __builtin_fma(tpr[c], ts[p][c] - tob[c], red[2 * p]);
                }
                if (f.use_amp) {
                    const float lg = __builtin_fmaf(__builtin_amdgcn_logf(d), 0.69314718055994531f, (float)st.ac[c]);
                    as[p][c] = (double)(__builtin_fmaf(-d, katt32, -lg));
                    red[2 * p + 1] = __builtin_fma(apr[c], as[p][c] - aob[c], red[2 * p + 1]);
                }
            }
        }
    } else {
#pragma unroll
    for (int p = 0; p < NPOS; ++p) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const bool valid = (lane + 64 * c) < f.S;
            const double dx = px[p] - st.sx[c], dy = py[p] - st.sy[c], dz = pz[p] - st.sz[c];
            // (the library is built with -ffp-contract=off -- proposals and decisions are exact arithmetic; the forward
            // model's products and sums are fused here by hand: one rounding less each, a tenth of the instructions)
            double d = htm_sqrt(__builtin_fma(dz, dz, __builtin_fma(dy, dy, dx * dx)));
            if (!valid) d = 1.0;     // a lane without a station: finite synthetics, and its precisions are 0 (load_obs_regs)
            ts[p][c] = 0.0; as[p][c] = 0.0;
            if (f.use_time) {
                ts[p][c] = __builtin_fma(d, rbeta, -st.tc[c]);
                red[2 * p] = __builtin_fma(tpr[c], ts[p][c] - tob[c], red[2 * p]);
            }
            if (f.use_amp) {
                as[p][c] = __builtin_fma(-d, katt, -htm_log(d)) - st.ac[c];
                red[2 * p + 1] = __builtin_fma(apr[c], as[p][c] - aob[c], red[2 * p + 1]);
            }
        }
    }
    }
    wave_sum<2 * NPOS>(red);
    const double rpst = ob.rpst, rpsa = ob.rpsa;
#pragma unroll
    for (int p = 0; p < NPOS; ++p) {
        const double t_mean = red[2 * p] * rpst;
        const double a_mean = red[2 * p + 1] * rpsa;
        double m = 0.0;                       // twice the misfit: the factor 1/2 (:285, :296) is exact and applied once
#pragma unroll
        for (int c = 0; c < NCH; ++c) {       // (a lane without a station adds r * r * 0)
            if (f.use_time) {
                const double r = tob[c] - (ts[p][c] - t_mean);
                m = __builtin_fma(r * r, tpr[c], m);
            }
            if (f.use_amp) {
                const double r = aob[c] - (as[p][c] - a_mean);
                m = __builtin_fma(r * r, apr[c], m);
            }
        }
        out[p] = 0.5 * m;
    }
}

// The same for NM MODELS of one event at once (k_full on stacked models: one event per wave, its observation rows resident):
// the models differ in position, corrections, vs and qs; their evaluations are independent instruction streams, written
// side by side so that one model's dependent chains (square root, logarithm, the DPP reductions) run under the other's --
// one station per lane has no such parallelism of its own.  Same arithmetic per model as event_misfit<NCH, 1>.
// PRE: `beta` and `q` hold 1 / vs and pi f / (qs vs) already (the caller formed them once for all events).
template <int NCH, int NM, bool F32 = false, bool PRE = false, class FW>
__device__ __forceinline__ void event_misfit_models(const FW &f, const ObsRegs<NCH, F32> &ob, int lane, const StaRegs<NCH> &geo,
                                                    const double (&tcm)[NM][NCH], const double (&acm)[NM][NCH],
                                                    const double (&px)[NM], const double (&py)[NM], const double (&pz)[NM],
                                                    const double (&beta)[NM], const double (&q)[NM], double (&out)[NM])
{
    double rbeta[NM], katt[NM];
#pragma unroll
    for (int p = 0; p < NM; ++p) {
        if constexpr (PRE) { rbeta[p] = beta[p]; katt[p] = q[p]; }
        else { const double qbeta = q[p] * beta[p]; rbeta[p] = 1.0 / beta[p]; katt[p] = (kPi * kFreq) / qbeta; }
    }
    double tob[NCH], tpr[NCH], aob[NCH], apr[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) { tob[c] = (double)ob.tob[c]; tpr[c] = (double)ob.tpr[c]; aob[c] = (double)ob.aob[c]; apr[c] = (double)ob.apr[c]; }
    double ts[NM][NCH], as[NM][NCH];
    double red[2 * NM];
#pragma unroll
    for (int k = 0; k < 2 * NM; ++k) red[k] = 0.0;
    const bool ut = f.use_time != 0, ua = f.use_amp != 0;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const bool valid = (lane + 64 * c) < f.S;
#pragma unroll
        for (int p = 0; p < NM; ++p) {
            if constexpr (F32) {
                const float dx = (float)(px[p] - geo.sx[c]), dy = (float)(py[p] - geo.sy[c]), dz = (float)(pz[p] - geo.sz[c]);
                float d = __builtin_amdgcn_sqrtf(__builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx)));
                if (!valid) d = 1.0f;
                const double t = (double)__builtin_fmaf(d, (float)rbeta[p], -(float)tcm[p][c]);
                const float lg = __builtin_fmaf(__builtin_amdgcn_logf(d), 0.69314718055994531f, (float)acm[p][c]);
                const double a = (double)(__builtin_fmaf(-d, (float)katt[p], -lg));
                ts[p][c] = ut ? t : 0.0; as[p][c] = ua ? a : 0.0;
            } else {
                const double dx = px[p] - geo.sx[c], dy = py[p] - geo.sy[c], dz = pz[p] - geo.sz[c];
                double d = htm_sqrt(__builtin_fma(dz, dz, __builtin_fma(dy, dy, dx * dx)));
                if (!valid) d = 1.0;
                const double t = __builtin_fma(d, rbeta[p], -tcm[p][c]);
                const double a = __builtin_fma(-d, katt[p], -htm_log(d)) - acm[p][c];
                ts[p][c] = ut ? t : 0.0; as[p][c] = ua ? a : 0.0;
            }
            // (a data type that is not used has zero precisions: its terms vanish, as the branches of event_misfit make them)
            red[2 * p] = __builtin_fma(ut ? tpr[c] : 0.0, ts[p][c] - tob[c], red[2 * p]);
            red[2 * p + 1] = __builtin_fma(ua ? apr[c] : 0.0, as[p][c] - aob[c], red[2 * p + 1]);
        }
    }
    wave_sum<2 * NM>(red);
    const double rpst = ob.rpst, rpsa = ob.rpsa;
#pragma unroll
    for (int p = 0; p < NM; ++p) {
        const double t_mean = red[2 * p] * rpst, a_mean = red[2 * p + 1] * rpsa;
        double m = 0.0;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const double rt = tob[c] - (ts[p][c] - t_mean), ra = aob[c] - (as[p][c] - a_mean);
            if (ut) m = __builtin_fma(rt * rt, tpr[c], m);
            if (ua) m = __builtin_fma(ra * ra, apr[c], m);
        }
        out[p] = 0.5 * m;
    }
}

// Generic-S fallback (n_sta > 64*4): strides over stations, recomputing the synthetics in the second
// pass instead of holding them in registers.
template <int NPOS, class FW>
__device__ __forceinline__ void event_misfit_generic(const FW &f, int ev, int lane,
                                                     const double *s_sx, const double *s_sy,
                                                     const double *s_sz, const double *tc,
                                                     const double *ac, int ov_kind, int ov_idx,
                                                     double ov_val, const double (&px)[NPOS],
                                                     const double (&py)[NPOS], const double (&pz)[NPOS],
                                                     double beta, double q, double (&out)[NPOS])
{
    const size_t base = (size_t)ev * (size_t)f.S;
    const double qbeta = q * beta;
    const double rbeta = 1.0 / beta, katt = (kPi * kFreq) / qbeta;      // (as in event_misfit: divisions once per call)
    double red[2 * NPOS];
#pragma unroll
    for (int k = 0; k < 2 * NPOS; ++k) red[k] = 0.0;
    for (int j = lane; j < f.S; j += 64) {
        const double tcj = (ov_kind == 2 && ov_idx == j) ? ov_val : tc[j];
        const double acj = (ov_kind == 4 && ov_idx == j) ? ov_val : ac[j];
#pragma unroll
        for (int p = 0; p < NPOS; ++p) {
            const double dx = px[p] - s_sx[j], dy = py[p] - s_sy[j], dz = pz[p] - s_sz[j];
            const double d = htm_sqrt(dx * dx + dy * dy + dz * dz);
            if (f.use_time) red[2 * p] += f.t_prec[base + j] * ((d * rbeta - tcj) - f.t_obs[base + j]);
            if (f.use_amp)
                red[2 * p + 1] += f.a_prec[base + j] *
                                  ((-(d * katt) - htm_log(d) - acj) - f.a_obs[base + j]);
        }
    }
    wave_sum<2 * NPOS>(red);
    const double rpst = f.use_time ? f.rpsum_t[ev] : 1.0;
    const double rpsa = f.use_amp ? f.rpsum_a[ev] : 1.0;
#pragma unroll
    for (int p = 0; p < NPOS; ++p) out[p] = 0.0;
    for (int j = lane; j < f.S; j += 64) {
        const double tcj = (ov_kind == 2 && ov_idx == j) ? ov_val : tc[j];
        const double acj = (ov_kind == 4 && ov_idx == j) ? ov_val : ac[j];
#pragma unroll
        for (int p = 0; p < NPOS; ++p) {
            const double dx = px[p] - s_sx[j], dy = py[p] - s_sy[j], dz = pz[p] - s_sz[j];
            const double d = htm_sqrt(dx * dx + dy * dy + dz * dz);
            if (f.use_time) {
                const double r = f.t_obs[base + j] - ((d * rbeta - tcj) - red[2 * p] * rpst);
                out[p] += r * r * (0.5 * f.t_prec[base + j]);
            }
            if (f.use_amp) {
                const double r = f.a_obs[base + j] -
                                 ((-(d * katt) - htm_log(d) - acj) - red[2 * p + 1] * rpsa);
                out[p] += r * r * (0.5 * f.a_prec[base + j]);
            }
        }
    }
}

template <int NCH>
__device__ __forceinline__ void load_sta_regs(StaRegs<NCH> &st, int S, int lane, const double *s_sx,
                                              const double *s_sy, const double *s_sz, const double *tc,
                                              const double *ac, int ov_kind, int ov_idx, double ov_val)
{
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int j = lane + 64 * c;
        const bool valid = j < S;
        st.sx[c] = valid ? s_sx[j] : 0.0;
        st.sy[c] = valid ? s_sy[j] : 0.0;
        st.sz[c] = valid ? s_sz[j] : 0.0;
        st.tc[c] = valid ? tc[j] : 0.0;
        st.ac[c] = valid ? ac[j] : 0.0;
        if (valid && ov_idx == j) {
            if (ov_kind == 2) st.tc[c] = ov_val;   // proposed t_corr(id)
            if (ov_kind == 4) st.ac[c] = ov_val;   // proposed a_corr(id)
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// chain-set state in HBM
// ---------------------------------------------------------------------------------------------------
struct ModelDev {            // one reference `type model` group for all chains of the rank
    double *x, *mu, *sigma, *step;   // [n_chains][nx]
    int32_t *ptype;                  // [n_chains][nx]
    int nx;
};

enum Stage : int { ST_IDLE = 0, ST_WAIT_FULL = 1, ST_WAIT_SWAP = 2 };
enum Mode : int { MODE_RUN = 0, MODE_ADVANCE = 1, MODE_FINISH = 2, MODE_APPLY = 3, MODE_LOCKRUN = 4 };

struct Proposal {            // everything random about one chain step, resolved at the start of the iteration
    int    type;             // 1 vs, 2 t_corr, 3 qs, 4 a_corr, 5..7 hypo (5 + icmp), cls_mcmc.f90:139-165
    int    idx;              // 0-based element inside the perturbed model
    int    evt;              // 1-based event id or -999
    int    prior_ok;
    int    need_full;
    int    accepted;
    int    cool;             // temperature < 1 + eps at judge time (counters, cls_mcmc.f90:186,:215)
    int    pad_;
    double x_new, lpr;       // cls_model.f90:172-186
    double x_old, L_old;     // what a speculative commit overwrote (restored if the pass is repeated)
    double r_judge, logr_judge;   // the rand_u() of cls_mcmc.f90:197 and its log (only drawn if prior_ok)
    double L_new;
};

// Work order for k_full, written by k_step in one contiguous block when it hands over, so that every
// workgroup of k_full learns what to do with a single memory round trip.
struct FullEntry {
    int chain, type, idx, pad;    // proposal being evaluated: cls_mcmc.f90 type 1..7, 0-based element
    double x_new, beta, q;        // proposed value; vs and qs of the evaluated model (proposal applied)
};
struct FullDesc {
    int n, pad0, pad1, pad2;      // number of entries; 0 = nothing pending
    FullEntry e[kMaxChains];
};

// Hand-shake words between the chain master (block 0) and the full-evaluation workers (blocks 1..W) of one
// k_mcmc launch.  Every word is polled / updated with agent-scope accesses only and sits on its own 128-B line.
struct PSync {
    unsigned long long quit;   char p3[120];   // launches whose master has finished
};

// Work orders of the persistent kernel as tagged 8-byte granules {tag : 32, payload : 32} (agent-scope
// stores/loads; the data is the flag: a granule is valid for job `tag` iff its upper half equals `tag`).
// Every chain has its own order slot and publishes it by itself, as soon as its proposal is known.  A slot is
// the whole order, so a worker learns everything with the poll that discovers it:
//   g0 {tag, launch (low 32 bits)}   g1 {tag, type | element << 3}   g2 {tag, x_new high}   g3 {tag, x_new low}
//   g4 {tag, element the chain committed in the iteration before, or ~0}   g5, g6 {tag, its value high / low}   g7 {tag, 0}
// (vs and qs of the evaluated model are chain state unless they are the proposal).  g4..g6: an order sent ahead
// by role P may overtake the chain's latest commit store; the worker waits until it reads that value back.  The slots are replicated
// ChainsDev::slot_rep times, slot_stride words apart (different memory channels), and worker w polls replica
// w % slot_rep: every poll is served by the memory side, so the pollers of one line queue up behind each other.
constexpr int kGranPerSlot = 8;
constexpr int kMaxSlotReplicas = 16;

struct Ctrl {
    int iter_done, iter_target, stage, n_full;
    int err, stop, n_lik, n_smp;
    long long spos;          // draws of the rank's mod_random stream consumed (committed) so far
    int swap_i1, swap_i2;    // cls_parallel.f90:226-230 (global chain indices)
    double swap_r, swap_logr;
    int slog_n, slog_cap;
    long long n_full_evals, n_partial_evals;
    unsigned long long jobs_total;   // full-evaluation jobs ever published (persistent kernel): the granule tag
};

// The rank's random stream, produced AHEAD of consumption by k_rawgen / k_stream_* on a side stream.
// Rings over the absolute stream position p (index = p & mask).  Nothing here depends on chain state.
struct StreamDev {
    uint32_t *raw;             // xorshift128 outputs (mod_random.f90:63-71)
    double *U, *LOGU, *G;      // rand_u, log(rand_u), Box-Muller value starting at p
    // for a chain step STARTING at p (cls_mcmc.f90:134-165 decoded ahead of time):
    int4 *dec;                 // {type, 0-based index, event id or -999, draws consumed if prior_ok}
    double *pg, *pr, *plogr;   // its Gaussian, its judge draw r and log r
    int *hop;                  // [p][k-1] = position after k optimistic chain steps, k = 1..8
    int4 *sw;                  // select_pair starting at p (rank 0): {i1, i2, draws used or -1, 0}
    long long mask;            // capacity - 1
    uint32_t *gen;             // [2][4] generator state after the last produced raw (the host alternates the two)
    long long *hop_end;        // every array is complete for positions < *hop_end
};

// What a proposal reads of the element it perturbs, packed: ONE 32-byte scalar load per step where the four per-field arrays
// cost four (four cache lines).  `rs2` = 1 / (2 sigma^2).  When every chain of the rank has the same priors and step sizes --
// the reference's set-up (hypo_tremor_mcmc.f90: one parameter file for all chains) -- only chain 0's records are kept in use
// (ChainsDev::prior_same): 1 / n_chains of the footprint, which decides whether the table stays in L2 (10 000 events x 16 chains:
// 15 MB in four arrays against 1 MB here; the step's front fell from 6 k to 3 k cycles).
struct __attribute__((aligned(32))) PriorRec {
    double mu, rs2, step;
    int32_t ptype, pad;
};

// (through the constant address space: a scalar load, as ld_const; the four doubles travel as one vector)
__device__ __forceinline__ PriorRec ld_prior(const PriorRec *p)
{
    typedef const f64x4 __attribute__((address_space(4))) *CP;
    const f64x4 v = *(CP)(unsigned long long)p;
    PriorRec r;
    r.mu = v[0]; r.rs2 = v[1]; r.step = v[2];
    r.ptype = __double2loint(v[3]); r.pad = 0;
    return r;
}

struct ChainsDev {
    int n_chains, n_procs, rank;
    int S, E;
    ModelDev hypo, tc, vs, ac, qs;
    // The five groups are carved out of ONE allocation per field, in proposal-type order
    // [vs | t_corr | qs | a_corr | hypo], so that the element a step perturbs is base + integer offset
    // (scalar arithmetic) instead of a five-way choice between pointers.
    double *xall, *muall, *sgall, *stall;
    double *rs2all;                  // 1 / (2 sigma^2) of every element (what the log prior ratio multiplies by)
    int32_t *ptall;
    double *temp, *L;                // [n_chains]
    int32_t *n_propose, *n_accept;   // [n_chains][7]
    double th1, th2, th3, th4;       // cumulative proposal thresholds, cls_mcmc.f90:139-153
    int n_burn, n_interval;
    Proposal *prop;                  // [n_chains]
    FullDesc *desc;                  // k_full's work order (chains whose proposal needs a full evaluation)
    double *partial;                 // [n_chains][n_wg]
    int n_wg;
    Ctrl *ctrl;
    // records
    int cap_lik, cap_smp;
    int32_t *lik_iter, *lik_chain; double *lik_val;
    int32_t *smp_iter, *smp_chain; double *smp_data;   // [cap_smp][3E + 2S + 2]
    int32_t *slog_i; double *slog_d;
    double *swap_rec;                // [4 + 2*n_chains] this rank's record (8-byte words)
    PSync *ps;                       // persistent-worker hand-shake (k_mcmc)
    unsigned long long *slots;       // order slots: replica r, chain c, granule g at [r*slot_stride + c*4 + g]
    int slot_rep, slot_stride;       // replicas (<= kMaxSlotReplicas), words between replicas
    unsigned long long *pgran;       // partial sums, tagged granules: chain c, worker k at [(c*n_workers + k)*pgran_stride + {0,1}]
    int pgran_stride;                // words between the granule pairs of two workers (>= 2)
    int npoll;                       // worker polls kept in flight (1..3)
    int mirror_n;                    // doubles of the LDS mirror of xall (vs, t_corr, qs, a_corr of all chains), 0 = none
    int mirror_steps;                // the step sizes of those elements are mirrored too (else role P reads them from memory)
    int rayleigh14;                  // some element of those four groups has a Rayleigh prior (prior_type 1)
    int n_workers;                   // worker blocks of a k_mcmc launch
    // in-kernel exchange of the swap records (MODE_LOCKRUN, htm_step.hpp exchange_post / exchange_finish)
    unsigned long long *inbox;       // [2][n_procs][xg] tagged granules, this rank's own (fine-grained memory: peers write it)
    unsigned long long *const *outbox;   // [n_procs] where rank q's inbox is mapped in this process (q = rank: inbox itself)
    int xg;                          // granules per record = 2 * (4 + 2 n_chains) + 2
    StreamDev stream;
    unsigned long long *stamps;      // diagnostic builds (-DHTM_STAMPS) only, else nullptr
    int dbg;                         // test switches (bit 0: HTM_DEBUG_NO_DROP, chain waves do not take disproved orders back)
    unsigned long long *diag;        // [32] what a wait that gave up was waiting for (written once, on the failure path; the host reports it)
    const PriorRec *prior;           // [element of the rank's parameter vector] (layout of xall)
    int prior_same;                  // all chains share chain 0's records: element (group, chain c, idx) reads (group, 0, idx)
    unsigned long long xwait_ticks;  // how long a rank waits for the peers' swap records (100 MHz ticks; 20 s, HTM_XCHG_TIMEOUT_MS)
    int xown;                        // a rank's own swap record is read from LDS instead of its inbox (default; HTM_XOWN=0: through memory like a peer's)
    int dbg_xfail_iter;              // test switch (HTM_DEBUG_XCHG_FAIL_ITER): from this iteration on the rank posts its swap records to nobody (0 = off)
    struct MbShared *mb;             // what the chains of a rank share when several master workgroups run them (htm_flow.hpp)
    int *prev_mid;                   // [n_chains] type | event << 3 of the chain's last step (free-running master: FlowShared::pv_mid), 0 = none
    unsigned long long *lo_gran;     // [n_chains][16] tagged granules of the pipelined master's orders (htm_pipe.hpp): [0..3] the proposed values of the two
                                     // hypocentre steps before the order's step (master -> workers), [8..15] the left-out events' sums at both positions (back)
};

}  // namespace htm
