// htm_hip.hip -- C ABI (include/htm_hip.h) over the gfx950 kernels in htm_kernels.hpp.
//
// Host code is plain C++17 + the HIP runtime: no torch, no third-party dependency.  There is no CPU
// fallback on purpose: every entry point needs a usable HIP device and fails with HTM_ENODEVICE otherwise.
#include "htm_hip.h"

#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <thread>
#include <string>
#include <vector>

#include "htm_kernels.hpp"
#include "htm_pipe.hpp"
#include "htm_select.hpp"

using namespace htm;

namespace {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIPCHK(call)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (call);                                                                        \
        if (e_ != hipSuccess)                                                                          \
            return fail(HTM_EHIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__,     \
                        __LINE__);                                                                     \
    } while (0)

int use_device(int device)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(HTM_ENODEVICE, "no HIP device available (%s); libhtm_hip has no CPU fallback",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    if (device < 0 || device >= n) return fail(HTM_EINVAL, "device %d out of range (0..%d)", device, n - 1);
    HIPCHK(hipSetDevice(device));
    return HTM_OK;
}

template <typename T>
int dev_alloc(std::vector<void *> &pool, T **p, size_t n)
{
    void *q = nullptr;
    HIPCHK(hipMalloc(&q, std::max<size_t>(n, 1) * sizeof(T)));
    pool.push_back(q);
    *p = static_cast<T *>(q);
    return HTM_OK;
}

template <typename T>
int dev_upload(std::vector<void *> &pool, T **p, const T *src, size_t n)
{
    int rc = dev_alloc(pool, p, n);
    if (rc) return rc;
    if (n) HIPCHK(hipMemcpy(*p, src, n * sizeof(T), hipMemcpyHostToDevice));
    return HTM_OK;
}

int nch_for(int S) { return S <= 64 ? 1 : S <= 128 ? 2 : S <= 256 ? 4 : 0; }

// ---- GF(2) algebra of mod_random's xorshift128 (reference src/mod_random.f90:63-71) ----------------------------
// One step is linear in the 128 state bits (x | y << 32 | z << 64 | w << 96): state' = T * state.  Powers of T let
// the device start any 64-draw segment of the stream directly (htm_stream.hpp, k_rawgen) and the host jump over any
// number of draws (htm_rng_jump).  A matrix is stored as its 128 columns.
struct Bits128 { uint32_t w[4]; };
using Mat128 = std::array<Bits128, 128>;

Bits128 xs_step(Bits128 s)
{
    const uint32_t t = s.w[0] ^ (s.w[0] << 11);
    Bits128 r;
    r.w[0] = s.w[1]; r.w[1] = s.w[2]; r.w[2] = s.w[3];
    r.w[3] = (s.w[3] ^ (s.w[3] >> 19)) ^ (t ^ (t >> 8));
    return r;
}
Bits128 gf2_matvec(const Mat128 &M, const Bits128 &v)
{
    Bits128 a{{0, 0, 0, 0}};
    for (int j = 0; j < 128; ++j)
        if ((v.w[j >> 5] >> (j & 31)) & 1u)
            for (int k = 0; k < 4; ++k) a.w[k] ^= M[j].w[k];
    return a;
}
// P[k] = T^(2^k), k = 0..63
const std::vector<Mat128> &xs_powers()
{
    static const std::vector<Mat128> P = [] {
        std::vector<Mat128> p(64);
        for (int j = 0; j < 128; ++j) {
            Bits128 e{{0, 0, 0, 0}};
            e.w[j >> 5] = 1u << (j & 31);
            p[0][j] = xs_step(e);
        }
        for (int k = 1; k < 64; ++k)
            for (int j = 0; j < 128; ++j) p[k][j] = gf2_matvec(p[k - 1], p[k - 1][j]);
        return p;
    }();
    return P;
}

}  // namespace

struct htm_forward {
    int device = 0, S = 0, E = 0, nch = 1;
    hipStream_t own_stream = nullptr, stream = nullptr;
    FwdDev dev{};
    std::vector<void *> pool;
    int n_wg = 0, epw = 1;
    // scratch for the host-pointer entry points (one model)
    double *d_hypo = nullptr, *d_tc = nullptr, *d_ac = nullptr, *d_scal = nullptr, *d_partial = nullptr;
    double *d_syn = nullptr;
    // scratch for batches
    double *d_bpartial = nullptr; size_t bpartial_cap = 0;
    double *d_bmodels = nullptr;  size_t bmodels_cap = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_wd = nullptr;
};

struct htm_chains {
    htm_forward *fwd = nullptr;
    ChainsDev dev{};
    Ctrl h_ctrl{};
    std::vector<void *> pool;
    hipGraph_t graph = nullptr;
    hipGraphExec_t gexec = nullptr;
    int pairs = 32;
    int nw = 1;                // chain waves of k_step (one more wave is the RNG producer)
    int ring_size = 512, wmax = 64;
    size_t step_smem = 0;
    int h_target = 0;          // host copy of the iteration target
    int rec_len = 0;
    std::vector<int32_t> lik_iter, lik_chain, smp_iter, smp_chain;
    std::vector<double> lik_val, smp_data;
    double last_device_us = 0.0;
    int last_graph_launches = 0;
    long long run_full0 = 0, run_part0 = 0, last_full = 0, last_part = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_wd = nullptr;
    // random-stream service (htm_stream.hpp): produced on a side stream ahead of consumption
    hipStream_t side = nullptr;
    hipEvent_t ev_side = nullptr;
    long long cap = 0;                         // ring capacity (positions)
    long long n_raw = 0, n_tr = 0, n_rec = 0, n_hop = 0;   // positions produced per stage (host view)
    long long spos_lo = 0, spos_hi = 0;        // bounds on the consumed position since the last sync
    const double *pending_gathered = nullptr;  // lock-step: records whose swap the next k_step applies
    double *d_gath_host = nullptr, *h_gath_pinned = nullptr;   // staging buffers of htm_chains_step_end_host
    unsigned long long launch_seq = 0;         // k_mcmc launches of this chain set so far (the kernels' launch index)
    bool persist = true;                       // k_mcmc (master + resident full-evaluation workers) vs k_step + k_full
    bool flow = false;                         // single-rank loop on the free-running master (htm_flow.hpp) instead of step_body
    int worker_cap = 250;                      // most worker blocks a launch takes (HTM_WORKER_CAP)
    long blocks_fit = 0;                       // resident blocks of a k_mcmc launch on this device (htm_chains_share_gpu)
    bool flow_lock = false;                    // lock-step ranks (MODE_LOCKRUN) on the free-running master too
    int mb_blocks = 1;                         // master workgroups of the single-rank loop (> 1: k_mcmc<.., 7>, eight chains each)
    bool pipe = false;                         // single-rank loop on the pipelined master (htm_pipe.hpp)
    bool pipe_lock = false;                    // lock-step ranks (MODE_LOCKRUN) on it too
    size_t pipe_smem = 0; int pipe_ring = 512; // its LDS size and stream window
    bool ctrl_fresh = false;                   // h_ctrl is the device's control block as of an idle stream (no launch since it was read)
    ChainsDev dev_np{};                        // view for the non-persistent kernels (partial sums per k_full tile)
    uint32_t init_state[4] = {0, 0, 0, 0};     // mod_random state at stream position 0
    // in-kernel exchange of the swap records (persistent lock-step): this rank's inbox, the peers' inboxes as mapped here
    unsigned long long *d_inbox = nullptr;
    size_t inbox_bytes = 0;
    std::vector<void *> peer_maps;             // hipIpcOpenMemHandle mappings to close
    unsigned probe_calls = 0;                  // htm_chains_xchg_probe calls so far (part of the probe's tokens)
    unsigned long long **d_outbox = nullptr;
    bool xchg_ready = false;
    u32x4 *d_jump = nullptr;                   // [kJumpLevels][128] columns of T^(64 * 2^b) (k_rawgen)
    int gen_par = 0;                           // which half of StreamDev::gen holds the current generator state
    double th[4] = {0, 0, 0, 0};
};

namespace {

int launch_full(htm_forward *h, const FullJob &jb, int gy)
{
    dim3 grid(h->n_wg, gy), block(256);
    const size_t smem = 0;
    if (h->dev.fp32 && (h->nch == 1 || h->nch == 2 || h->nch == 4)) {        // fp32 forward (htm_forward_set_precision)
        if (jb.desc) {
            switch (h->nch) {
            case 1: hipLaunchKernelGGL((k_full<1, false, true>), grid, block, smem, h->stream, h->dev, jb); break;
            case 2: hipLaunchKernelGGL((k_full<2, false, true>), grid, block, smem, h->stream, h->dev, jb); break;
            default: hipLaunchKernelGGL((k_full<4, false, true>), grid, block, smem, h->stream, h->dev, jb); break;
            }
        } else {
            switch (h->nch) {
            case 1: hipLaunchKernelGGL((k_full<1, true, true>), grid, block, smem, h->stream, h->dev, jb); break;
            case 2: hipLaunchKernelGGL((k_full<2, true, true>), grid, block, smem, h->stream, h->dev, jb); break;
            default: hipLaunchKernelGGL((k_full<4, true, true>), grid, block, smem, h->stream, h->dev, jb); break;
            }
        }
    } else if (jb.desc) {
        switch (h->nch) {
        case 1: hipLaunchKernelGGL((k_full<1, false>), grid, block, smem, h->stream, h->dev, jb); break;
        case 2: hipLaunchKernelGGL((k_full<2, false>), grid, block, smem, h->stream, h->dev, jb); break;
        case 4: hipLaunchKernelGGL((k_full<4, false>), grid, block, smem, h->stream, h->dev, jb); break;
        default: hipLaunchKernelGGL((k_full<0, false>), grid, block, smem, h->stream, h->dev, jb); break;
        }
    } else {
        switch (h->nch) {
        case 1: hipLaunchKernelGGL((k_full<1, true>), grid, block, smem, h->stream, h->dev, jb); break;
        case 2: hipLaunchKernelGGL((k_full<2, true>), grid, block, smem, h->stream, h->dev, jb); break;
        case 4: hipLaunchKernelGGL((k_full<4, true>), grid, block, smem, h->stream, h->dev, jb); break;
        default: hipLaunchKernelGGL((k_full<0, true>), grid, block, smem, h->stream, h->dev, jb); break;
        }
    }
    HIPCHK(hipGetLastError());
    return HTM_OK;
}

int launch_mcmc(htm_chains *hc, int mode, int target, const double *gathered)
{
    htm_forward *h = hc->fwd;
    const bool mb = mode == MODE_RUN && hc->flow && hc->mb_blocks > 1 && !hc->pipe;
    dim3 grid((mb ? hc->mb_blocks : 1) + hc->dev.n_workers), block(512);
    hc->ctrl_fresh = false;
    const unsigned long long seq = ++hc->launch_seq;      // this chain set's k_mcmc launches, counted from 1
#define HTM_LAUNCH_MCMC(N, F, K) hipLaunchKernelGGL((k_mcmc<N, F, K>), grid, block, hc->step_smem, h->stream, h->dev, hc->dev, mode, target, gathered, hc->ring_size, hc->wmax, seq)
#define HTM_LAUNCH_MCMC_K(K)                                                                              \
    do {                                                                                                   \
        if (h->dev.fp32) { if (h->nch == 1) HTM_LAUNCH_MCMC(1, true, K); else HTM_LAUNCH_MCMC(2, true, K); } \
        else if (h->nch == 1) HTM_LAUNCH_MCMC(1, false, K);                                                \
        else if (h->nch == 2) HTM_LAUNCH_MCMC(2, false, K);                                                \
        else HTM_LAUNCH_MCMC(0, false, K);                                                                 \
    } while (0)
    // one instantiation per main loop: the single-rank loop, one lock-step iteration per launch, persistent lock-step
    if (mode == MODE_RUN && hc->pipe) {
        // the pipelined master (htm_pipe.hpp): one or two stations per lane only
#define HTM_LAUNCH_PIPE(N, F) hipLaunchKernelGGL((k_mcmc<N, F, 5>), grid, dim3(mcmc_threads<N, 5>()), hc->pipe_smem, h->stream, h->dev, hc->dev, mode, target, gathered, hc->pipe_ring, hc->wmax, seq)
        if (h->dev.fp32) { if (h->nch == 1) HTM_LAUNCH_PIPE(1, true); else HTM_LAUNCH_PIPE(2, true); }
        else if (h->nch == 1) HTM_LAUNCH_PIPE(1, false);
        else HTM_LAUNCH_PIPE(2, false);
#undef HTM_LAUNCH_PIPE
    }
    else if (mode == MODE_LOCKRUN && hc->pipe_lock) {
#define HTM_LAUNCH_PIPE(N, F) hipLaunchKernelGGL((k_mcmc<N, F, 6>), grid, dim3(mcmc_threads<N, 6>()), hc->pipe_smem, h->stream, h->dev, hc->dev, mode, target, gathered, hc->pipe_ring, hc->wmax, seq)
        if (h->dev.fp32) { if (h->nch == 1) HTM_LAUNCH_PIPE(1, true); else HTM_LAUNCH_PIPE(2, true); }
        else if (h->nch == 1) HTM_LAUNCH_PIPE(1, false);
        else HTM_LAUNCH_PIPE(2, false);
#undef HTM_LAUNCH_PIPE
    }
    else if (mb) {
        // several master workgroups: what their chains share lives in memory (MbShared), set up by a one-wave kernel first
        hipLaunchKernelGGL(k_mb_init, dim3(1), dim3(64), 0, h->stream, hc->dev, target);
        if (h->dev.fp32) { if (h->nch == 1) HTM_LAUNCH_MCMC(1, true, 7); else HTM_LAUNCH_MCMC(2, true, 7); }
        else if (h->nch == 1) HTM_LAUNCH_MCMC(1, false, 7);
        else HTM_LAUNCH_MCMC(2, false, 7);
    }
    else if (mode == MODE_RUN && hc->flow) HTM_LAUNCH_MCMC_K(3);
    else if (mode == MODE_RUN) HTM_LAUNCH_MCMC_K(0);
    else if (mode == MODE_LOCKRUN && hc->flow_lock) HTM_LAUNCH_MCMC_K(4);
    else if (mode == MODE_LOCKRUN) HTM_LAUNCH_MCMC_K(2);
    else HTM_LAUNCH_MCMC_K(1);
#undef HTM_LAUNCH_MCMC_K
#undef HTM_LAUNCH_MCMC
    HIPCHK(hipGetLastError());
    return HTM_OK;
}

int launch_step(htm_chains *hc, int mode, int target, const double *gathered)
{
    htm_forward *h = hc->fwd;
    hc->ctrl_fresh = false;
    dim3 grid(1), block(64 * hc->nw);
    if (h->dev.fp32) {
        if (h->nch == 1) hipLaunchKernelGGL((k_step<1, true>), grid, block, hc->step_smem, h->stream, h->dev, hc->dev_np, mode, target, gathered, hc->ring_size, hc->wmax);
        else hipLaunchKernelGGL((k_step<2, true>), grid, block, hc->step_smem, h->stream, h->dev, hc->dev_np, mode, target, gathered, hc->ring_size, hc->wmax);
        HIPCHK(hipGetLastError());
        return HTM_OK;
    }
    switch (h->nch) {
    case 1: hipLaunchKernelGGL(k_step<1>, grid, block, hc->step_smem, h->stream, h->dev, hc->dev_np, mode, target, gathered, hc->ring_size, hc->wmax); break;
    case 2: hipLaunchKernelGGL(k_step<2>, grid, block, hc->step_smem, h->stream, h->dev, hc->dev_np, mode, target, gathered, hc->ring_size, hc->wmax); break;
    default: hipLaunchKernelGGL(k_step<0>, grid, block, hc->step_smem, h->stream, h->dev, hc->dev_np, mode, target, gathered, hc->ring_size, hc->wmax); break;
    }
    HIPCHK(hipGetLastError());
    return HTM_OK;
}

// Append n (multiple of 64) positions to the rank's random stream on the side stream.  Asynchronous; the
// new coverage is published to the device (StreamDev::hop_end) by the last kernel of the sequence.
int stream_produce(htm_chains *hc, long long n)
{
    if (n <= 0) return HTM_OK;
    n = (n + 63) / 64 * 64;
    const StreamDev &sd = hc->dev.stream;
    // never overwrite positions that may still be needed: everything from (consumed - 4) on
    const long long room = hc->cap - (hc->n_raw - (hc->spos_lo - 4));
    if (n > room) n = room / 64 * 64;
    if (n <= 0) return HTM_OK;
    hipStream_t st = hc->side;
    hipLaunchKernelGGL(k_rawgen, dim3((unsigned)((n + 4095) / 4096)), dim3(64), 0, st, sd, hc->n_raw, (int)n, hc->d_jump,
                       sd.gen + 4 * hc->gen_par, sd.gen + 4 * (hc->gen_par ^ 1));
    hc->gen_par ^= 1;
    hc->n_raw += n;
    auto blocks = [](long long cnt) { return dim3((unsigned)((cnt + 255) / 256)); };
    long long e = hc->n_raw - 1;
    if (e > hc->n_tr) { hipLaunchKernelGGL(k_stream_tr, blocks(e - hc->n_tr), dim3(256), 0, st, sd, hc->n_tr, e); hc->n_tr = e; }
    e = hc->n_tr - kRecLag;
    if (e > hc->n_rec) {
        hipLaunchKernelGGL(k_stream_rec, blocks(e - hc->n_rec), dim3(256), 0, st, sd, hc->n_rec, e, hc->th[0], hc->th[1],
                           hc->th[2], hc->th[3], hc->dev.S, hc->dev.E, hc->dev.n_procs, hc->dev.n_chains);
        hc->n_rec = e;
    }
    e = hc->n_rec - kHopLag;
    if (e > hc->n_hop) { hipLaunchKernelGGL(k_stream_hop, blocks(e - hc->n_hop), dim3(256), 0, st, sd, hc->n_hop, e); hc->n_hop = e; }
    hipLaunchKernelGGL(k_publish, dim3(1), dim3(1), 0, st, sd.hop_end, hc->n_hop);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(hc->ev_side, st));
    return HTM_OK;
}

FullJob chain_full_job(htm_chains *hc)
{
    FullJob jb{};
    const ChainsDev &d = hc->dev_np;
    jb.hypo = d.hypo.x; jb.hypo_stride = d.hypo.nx;
    jb.tc = d.tc.x; jb.tc_stride = d.S;
    jb.ac = d.ac.x; jb.ac_stride = d.S;
    jb.vs = d.vs.x; jb.qs = d.qs.x;
    jb.desc = d.desc;
    jb.n_models = d.n_chains;
    jb.partial = d.partial; jb.n_wg = hc->fwd->n_wg; jb.epw = hc->fwd->epw;
    return jb;
}

int ensure_batch_scratch(htm_forward *h, int n_models)
{
    const size_t need = (size_t)n_models * h->n_wg;
    if (need > h->bpartial_cap) {
        if (h->d_bpartial) HIPCHK(hipFree(h->d_bpartial));
        h->d_bpartial = nullptr;
        HIPCHK(hipMalloc(reinterpret_cast<void **>(&h->d_bpartial), need * sizeof(double)));
        h->bpartial_cap = need;
    }
    return HTM_OK;
}

int full_batch_dev(htm_forward *h, int n_models, const double *d_hypo, const double *d_tc, const double *d_vs,
                   const double *d_ac, const double *d_qs, double *d_L)
{
    int rc = ensure_batch_scratch(h, n_models);
    if (rc) return rc;
    FullJob jb{};
    jb.hypo = d_hypo; jb.hypo_stride = 3L * h->E;
    jb.tc = d_tc; jb.tc_stride = h->S;
    jb.ac = d_ac; jb.ac_stride = h->S;
    jb.vs = d_vs; jb.qs = d_qs;
    jb.n_models = n_models;
    jb.partial = h->d_bpartial; jb.n_wg = h->n_wg; jb.epw = h->epw;
    int gy = std::max(1, std::min(n_models, 2048 / std::max(1, h->n_wg)));
    if (const char *e = getenv("HTM_FULL_BLOCKS")) gy = std::max(1, std::min(n_models, atoi(e) / std::max(1, h->n_wg)));      // (tuning: blocks per launch)
    rc = launch_full(h, jb, gy);
    if (rc) return rc;
    hipLaunchKernelGGL(k_sum_partials, dim3(n_models), dim3(64), 0, h->stream, h->d_bpartial, h->n_wg,
                       h->dev.const_sum, d_L);
    HIPCHK(hipGetLastError());
    return HTM_OK;
}

}  // namespace

// ====================================================================================================
extern "C" {

const char *htm_last_error(void) { return g_err.c_str(); }
int htm_abi_version(void) { return 1; }

int htm_device_count(int *n)
{
    if (!n) return fail(HTM_EINVAL, "n is NULL");
    *n = 0;
    hipError_t e = hipGetDeviceCount(n);
    if (e != hipSuccess) { *n = 0; return fail(HTM_ENODEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
    return HTM_OK;
}

int htm_device_physical_id(int device, int *id)
{
    if (!id) return fail(HTM_EINVAL, "id is NULL");
    *id = -1;
    int dom = 0, bus = 0, dv = 0;
    if (hipDeviceGetAttribute(&dom, hipDeviceAttributePciDomainID, device) != hipSuccess ||
        hipDeviceGetAttribute(&bus, hipDeviceAttributePciBusId, device) != hipSuccess ||
        hipDeviceGetAttribute(&dv, hipDeviceAttributePciDeviceId, device) != hipSuccess)
        return fail(HTM_ENODEVICE, "no PCI address for HIP device %d", device);
    *id = ((dom & 0x7fff) << 16) | ((bus & 0xff) << 8) | (dv & 0xff);
    return HTM_OK;
}

// ---------------------------------------------------------------------------------------------------
int htm_forward_create(int n_sta, int n_events, const double *sta_x, const double *sta_y, const double *sta_z,
                       const double *t_obs, const double *t_stdv, const double *a_obs, const double *a_stdv,
                       int use_time, int use_amp, int device, htm_forward **out)
{
    if (!out) return fail(HTM_EINVAL, "out is NULL");
    *out = nullptr;
    if (n_sta <= 0 || n_events <= 0) return fail(HTM_EINVAL, "n_sta and n_events must be positive");
    if (!sta_x || !sta_y || !sta_z || !t_obs || !t_stdv || !a_obs || !a_stdv)
        return fail(HTM_EINVAL, "NULL input array");
    int rc = use_device(device);
    if (rc) return rc;

    htm_forward *h = new htm_forward();
    h->device = device; h->S = n_sta; h->E = n_events; h->nch = nch_for(n_sta);
    const size_t n = (size_t)n_sta * n_events;

    // init_forward, cls_forward.f90:76-92: precision, log-stdv and the missing-data rule (keyed on t_stdv
    // only; log-stdv := 1.0 (sic), stdv := 1, precision := 1 for BOTH data types)
    std::vector<double> tpr(n), apr(n), pst(n_events), psa(n_events);
    const double log_2pi_half = 0.5 * std::log(2.0 * std::acos(-1.0));
    double const_t = 0.0, const_a = 0.0;
    for (int i = 0; i < n_events; ++i) {
        double st = 0.0, sa = 0.0;
        for (int j = 0; j < n_sta; ++j) {
            const size_t k = (size_t)i * n_sta + j;
            double lts, las;
            if (t_stdv[k] > 1.e-16) {
                lts = std::log(t_stdv[k]); tpr[k] = 1.0 / (t_stdv[k] * t_stdv[k]);
                las = std::log(a_stdv[k]); apr[k] = 1.0 / (a_stdv[k] * a_stdv[k]);
            } else {
                lts = 1.0; tpr[k] = 1.0; las = 1.0; apr[k] = 1.0;
            }
            st += tpr[k]; sa += apr[k];
            const_t += log_2pi_half + lts;
            const_a += log_2pi_half + las;
        }
        pst[i] = st; psa[i] = sa;
    }

    auto cleanup = [&](int code) { htm_forward_destroy(h); return code; };
    double *p = nullptr;
#define UP(dst, src, cnt)                                          \
    if ((rc = dev_upload(h->pool, &p, (src), (cnt)))) return cleanup(rc); \
    dst = p;
    UP(h->dev.sx, sta_x, n_sta) UP(h->dev.sy, sta_y, n_sta) UP(h->dev.sz, sta_z, n_sta)
    UP(h->dev.t_obs, t_obs, n) UP(h->dev.t_prec, tpr.data(), n)
    UP(h->dev.a_obs, a_obs, n) UP(h->dev.a_prec, apr.data(), n)
    UP(h->dev.psum_t, pst.data(), n_events) UP(h->dev.psum_a, psa.data(), n_events)
    {
        std::vector<double> rt(n_events), ra(n_events);
        for (int i = 0; i < n_events; ++i) { rt[i] = 1.0 / pst[i]; ra[i] = 1.0 / psa[i]; }
        UP(h->dev.rpsum_t, rt.data(), n_events) UP(h->dev.rpsum_a, ra.data(), n_events)
    }
#undef UP
    h->dev.S = n_sta; h->dev.E = n_events; h->dev.use_time = use_time ? 1 : 0; h->dev.use_amp = use_amp ? 1 : 0;
    h->dev.const_sum = (use_time ? const_t : 0.0) + (use_amp ? const_a : 0.0);

    h->epw = std::max(1, (n_events + 4 * 1024 - 1) / (4 * 1024));
    h->n_wg = (n_events + 4 * h->epw - 1) / (4 * h->epw);

    if ((rc = dev_alloc(h->pool, &h->d_hypo, 3 * (size_t)n_events))) return cleanup(rc);
    if ((rc = dev_alloc(h->pool, &h->d_tc, n_sta))) return cleanup(rc);
    if ((rc = dev_alloc(h->pool, &h->d_ac, n_sta))) return cleanup(rc);
    if ((rc = dev_alloc(h->pool, &h->d_scal, 16))) return cleanup(rc);
    if ((rc = dev_alloc(h->pool, &h->d_partial, h->n_wg))) return cleanup(rc);
    if ((rc = dev_alloc(h->pool, &h->d_syn, n))) return cleanup(rc);
    hipError_t e = hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking);
    if (e != hipSuccess) return cleanup(fail(HTM_EHIP, "hipStreamCreate: %s", hipGetErrorString(e)));
    h->stream = h->own_stream;
    if (hipEventCreate(&h->ev0) != hipSuccess || hipEventCreate(&h->ev1) != hipSuccess)
        return cleanup(fail(HTM_EHIP, "hipEventCreate failed"));
    *out = h;
    return HTM_OK;
}

int htm_forward_destroy(htm_forward *h)
{
    if (!h) return HTM_OK;
    (void)hipSetDevice(h->device);
    if (h->own_stream) (void)hipStreamSynchronize(h->own_stream);
    for (void *p : h->pool) (void)hipFree(p);
    if (h->d_bpartial) (void)hipFree(h->d_bpartial);
    if (h->d_bmodels) (void)hipFree(h->d_bmodels);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    delete h;
    return HTM_OK;
}

int htm_forward_set_precision(htm_forward *h, int forward_fp32)
{
    if (!h) return fail(HTM_EINVAL, "NULL handle");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (!forward_fp32) { h->dev.fp32 = 0; return HTM_OK; }
    if (h->nch != 1 && h->nch != 2) return fail(HTM_EINVAL, "the fp32 forward covers n_sta <= 128 (this handle has %d stations)", h->S);
    if (!h->dev.t_obs32) {
        // the four observation streams once more as float: the bytes a full evaluation reads are halved
        const size_t n = (size_t)h->S * h->E;
        std::vector<double> tmp(n);
        std::vector<float> f32(n);
        const double *src[4] = {h->dev.t_obs, h->dev.t_prec, h->dev.a_obs, h->dev.a_prec};
        const float **dst[4] = {&h->dev.t_obs32, &h->dev.t_prec32, &h->dev.a_obs32, &h->dev.a_prec32};
        for (int k = 0; k < 4; ++k) {
            HIPCHK(hipMemcpy(tmp.data(), src[k], n * sizeof(double), hipMemcpyDeviceToHost));
            for (size_t i = 0; i < n; ++i) f32[i] = (float)tmp[i];
            float *p = nullptr;
            int rc = dev_upload(h->pool, &p, f32.data(), n);
            if (rc) return rc;
            *dst[k] = p;
        }
    }
    h->dev.fp32 = 1;
    return HTM_OK;
}

int htm_forward_set_stream(htm_forward *h, void *hip_stream)
{
    if (!h) return fail(HTM_EINVAL, "NULL handle");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->stream = static_cast<hipStream_t>(hip_stream);
    return HTM_OK;
}

int htm_forward_reset_stream(htm_forward *h)
{
    if (!h) return fail(HTM_EINVAL, "NULL handle");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->stream = h->own_stream;
    return HTM_OK;
}

int htm_forward_sync(htm_forward *h)
{
    if (!h) return fail(HTM_EINVAL, "NULL handle");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    return HTM_OK;
}

int htm_forward_loglik_full(htm_forward *h, const double *hypo, const double *t_corr, double vs,
                            const double *a_corr, double qs, double *log_likelihood)
{
    if (!h || !hypo || !t_corr || !a_corr || !log_likelihood) return fail(HTM_EINVAL, "NULL argument");
    HIPCHK(hipSetDevice(h->device));
    const double sc[2] = {vs, qs};
    HIPCHK(hipMemcpyAsync(h->d_hypo, hypo, 3 * (size_t)h->E * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->d_tc, t_corr, h->S * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->d_ac, a_corr, h->S * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->d_scal, sc, sizeof(sc), hipMemcpyHostToDevice, h->stream));
    int rc = full_batch_dev(h, 1, h->d_hypo, h->d_tc, h->d_scal, h->d_ac, h->d_scal + 1, h->d_scal + 2);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(log_likelihood, h->d_scal + 2, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return HTM_OK;
}

int htm_forward_loglik_partial(htm_forward *h, int evt_id, const double hypo_old_xyz[3],
                               double log_likelihood_old, const double hypo_xyz[3], const double *t_corr,
                               double vs, const double *a_corr, double qs, double *log_likelihood)
{
    if (!h || !hypo_old_xyz || !hypo_xyz || !t_corr || !a_corr || !log_likelihood)
        return fail(HTM_EINVAL, "NULL argument");
    if (evt_id < 1 || evt_id > h->E) return fail(HTM_EINVAL, "evt_id %d out of range 1..%d", evt_id, h->E);
    HIPCHK(hipSetDevice(h->device));
    const double sc[6] = {hypo_old_xyz[0], hypo_old_xyz[1], hypo_old_xyz[2], hypo_xyz[0], hypo_xyz[1], hypo_xyz[2]};
    HIPCHK(hipMemcpyAsync(h->d_tc, t_corr, h->S * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->d_ac, a_corr, h->S * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->d_scal + 4, sc, sizeof(sc), hipMemcpyHostToDevice, h->stream));
    if (h->dev.fp32 && h->nch <= 2 && h->nch >= 1) {
        if (h->nch == 1) hipLaunchKernelGGL((k_partial_one<1, true>), dim3(1), dim3(64), 0, h->stream, h->dev, evt_id - 1, h->d_scal + 4, h->d_scal + 7, h->d_tc, h->d_ac, vs, qs, log_likelihood_old, h->d_scal + 2);
        else hipLaunchKernelGGL((k_partial_one<2, true>), dim3(1), dim3(64), 0, h->stream, h->dev, evt_id - 1, h->d_scal + 4, h->d_scal + 7, h->d_tc, h->d_ac, vs, qs, log_likelihood_old, h->d_scal + 2);
    } else
    switch (h->nch) {
    case 1: hipLaunchKernelGGL(k_partial_one<1>, dim3(1), dim3(64), 0, h->stream, h->dev, evt_id - 1, h->d_scal + 4, h->d_scal + 7, h->d_tc, h->d_ac, vs, qs, log_likelihood_old, h->d_scal + 2); break;
    case 2: hipLaunchKernelGGL(k_partial_one<2>, dim3(1), dim3(64), 0, h->stream, h->dev, evt_id - 1, h->d_scal + 4, h->d_scal + 7, h->d_tc, h->d_ac, vs, qs, log_likelihood_old, h->d_scal + 2); break;
    case 4: hipLaunchKernelGGL(k_partial_one<4>, dim3(1), dim3(64), 0, h->stream, h->dev, evt_id - 1, h->d_scal + 4, h->d_scal + 7, h->d_tc, h->d_ac, vs, qs, log_likelihood_old, h->d_scal + 2); break;
    default: hipLaunchKernelGGL(k_partial_one<0>, dim3(1), dim3(64), 0, h->stream, h->dev, evt_id - 1, h->d_scal + 4, h->d_scal + 7, h->d_tc, h->d_ac, vs, qs, log_likelihood_old, h->d_scal + 2); break;
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(log_likelihood, h->d_scal + 2, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return HTM_OK;
}

static int syn_common(htm_forward *h, const double *hypo, const double *corr, double beta, double q, int which,
                      int evt_id, double *out)
{
    if (!h || !hypo || !corr || !out) return fail(HTM_EINVAL, "NULL argument");
    if (evt_id != 0 && (evt_id < 1 || evt_id > h->E))
        return fail(HTM_EINVAL, "evt_id %d out of range 1..%d", evt_id, h->E);
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipMemcpyAsync(h->d_hypo, hypo, 3 * (size_t)h->E * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->d_tc, corr, h->S * sizeof(double), hipMemcpyHostToDevice, h->stream));
    const int nblk = evt_id ? 1 : (h->E + 3) / 4;
    hipLaunchKernelGGL(k_syn, dim3(nblk), dim3(256), 0, h->stream, h->dev, h->d_hypo, h->d_tc, beta, q, which,
                       evt_id ? evt_id - 1 : -1, h->d_syn);
    HIPCHK(hipGetLastError());
    const size_t cnt = evt_id ? (size_t)h->S : (size_t)h->S * h->E;
    HIPCHK(hipMemcpyAsync(out, h->d_syn, cnt * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return HTM_OK;
}

int htm_forward_travel_time(htm_forward *h, const double *hypo, const double *t_corr, double vs, double *t_syn)
{ return syn_common(h, hypo, t_corr, vs, 1.0, 0, 0, t_syn); }
int htm_forward_amp(htm_forward *h, const double *hypo, const double *a_corr, double qs, double vs, double *a_syn)
{ return syn_common(h, hypo, a_corr, vs, qs, 1, 0, a_syn); }
int htm_forward_travel_time_single(htm_forward *h, int evt_id, const double *hypo, const double *t_corr,
                                   double vs, double *t_syn)
{
    if (h && (evt_id < 1 || evt_id > h->E)) return fail(HTM_EINVAL, "evt_id %d out of range", evt_id);
    return syn_common(h, hypo, t_corr, vs, 1.0, 0, evt_id, t_syn);
}
int htm_forward_amp_single(htm_forward *h, int evt_id, const double *hypo, const double *a_corr, double qs,
                           double vs, double *a_syn)
{
    if (h && (evt_id < 1 || evt_id > h->E)) return fail(HTM_EINVAL, "evt_id %d out of range", evt_id);
    return syn_common(h, hypo, a_corr, vs, qs, 1, evt_id, a_syn);
}

int htm_forward_loglik_full_batch_dev(htm_forward *h, int n_models, const double *d_hypo, const double *d_t_corr,
                                      const double *d_vs, const double *d_a_corr, const double *d_qs,
                                      double *d_log_likelihood)
{
    if (!h || !d_hypo || !d_t_corr || !d_vs || !d_a_corr || !d_qs || !d_log_likelihood)
        return fail(HTM_EINVAL, "NULL argument");
    if (n_models <= 0) return fail(HTM_EINVAL, "n_models must be positive");
    HIPCHK(hipSetDevice(h->device));
    return full_batch_dev(h, n_models, d_hypo, d_t_corr, d_vs, d_a_corr, d_qs, d_log_likelihood);
}

int htm_forward_loglik_full_batch(htm_forward *h, int n_models, const double *hypo, const double *t_corr,
                                  const double *vs, const double *a_corr, const double *qs, double *log_likelihood)
{
    if (!h || !hypo || !t_corr || !vs || !a_corr || !qs || !log_likelihood) return fail(HTM_EINVAL, "NULL argument");
    if (n_models <= 0) return fail(HTM_EINVAL, "n_models must be positive");
    HIPCHK(hipSetDevice(h->device));
    const size_t nh = 3 * (size_t)h->E, ns = h->S;
    const size_t per = nh + 2 * ns + 3, need = per * n_models;
    if (need > h->bmodels_cap) {
        if (h->d_bmodels) HIPCHK(hipFree(h->d_bmodels));
        h->d_bmodels = nullptr;
        HIPCHK(hipMalloc(reinterpret_cast<void **>(&h->d_bmodels), need * sizeof(double)));
        h->bmodels_cap = need;
    }
    double *dh = h->d_bmodels, *dt = dh + nh * n_models, *da = dt + ns * n_models, *dv = da + ns * n_models,
           *dq = dv + n_models, *dL = dq + n_models;
    HIPCHK(hipMemcpyAsync(dh, hypo, nh * n_models * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(dt, t_corr, ns * n_models * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(da, a_corr, ns * n_models * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(dv, vs, n_models * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(dq, qs, n_models * sizeof(double), hipMemcpyHostToDevice, h->stream));
    int rc = full_batch_dev(h, n_models, dh, dt, dv, da, dq, dL);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(log_likelihood, dL, n_models * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return HTM_OK;
}

int htm_forward_time_full_batch_dev(htm_forward *h, int n_models, const double *d_hypo, const double *d_t_corr,
                                    const double *d_vs, const double *d_a_corr, const double *d_qs,
                                    double *d_log_likelihood, int reps, double *avg_us)
{
    if (!h || !avg_us || reps <= 0) return fail(HTM_EINVAL, "bad argument");
    HIPCHK(hipSetDevice(h->device));
    int rc = htm_forward_loglik_full_batch_dev(h, n_models, d_hypo, d_t_corr, d_vs, d_a_corr, d_qs, d_log_likelihood);
    if (rc) return rc;   // warm-up, also sizes the scratch
    HIPCHK(hipEventRecord(h->ev0, h->stream));
    for (int r = 0; r < reps; ++r) {
        rc = full_batch_dev(h, n_models, d_hypo, d_t_corr, d_vs, d_a_corr, d_qs, d_log_likelihood);
        if (rc) return rc;
    }
    HIPCHK(hipEventRecord(h->ev1, h->stream));
    HIPCHK(hipEventSynchronize(h->ev1));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, h->ev0, h->ev1));
    *avg_us = 1000.0 * ms / reps;
    return HTM_OK;
}

// ---------------------------------------------------------------------------------------------------
// chains
// ---------------------------------------------------------------------------------------------------
// Worker blocks of a persistent launch: as many as fit (one per CU next to the master's), at most one per `waves_per_block` events.
// A full evaluation takes as long as its busiest wave, which evaluates ceil(E / (waves per block x blocks)) events: at 10 000
// events and 8 waves per block 250 blocks give five rounds where 240 gave six (+8 % at 10 000 x 128 x 16 fp32, +6 % at 10 000 x 64 x 8).
// (Taking the SMALLEST block count that reaches the same number of rounds was measured too: -1 % -- the waves with a round less
// leave the memory system to the others sooner.)
static int worker_blocks(int E, int waves_per_block, long cap)
{
    return (int)std::max<long>(1, std::min<long>(std::max<long>(1, cap), (E + waves_per_block - 1) / waves_per_block));
}

// one group of the rank's parameter vector: a window [off, off + nx*nc) of the five per-field allocations
static int upload_model(htm_chains *hc, ModelDev &m, const htm_model_init &in, int nx, int nc, size_t off, const char *name)
{
    if (!in.x) return fail(HTM_EINVAL, "%s.x is NULL", name);
    const size_t n = (size_t)nx * nc;
    std::vector<double> zeros(n, 0.0);
    std::vector<double> ones(n, 1.0);
    std::vector<int32_t> izeros(n, 0);
    const ChainsDev &d = hc->dev;
    m.nx = nx;
    m.x = d.xall + off; m.mu = d.muall + off; m.sigma = d.sgall + off; m.step = d.stall + off; m.ptype = d.ptall + off;
    HIPCHK(hipMemcpy(m.x, in.x, n * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(m.mu, in.mu ? in.mu : zeros.data(), n * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(m.sigma, in.sigma ? in.sigma : ones.data(), n * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(m.step, in.step_size ? in.step_size : zeros.data(), n * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(m.ptype, in.prior_type ? in.prior_type : izeros.data(), n * sizeof(int32_t), hipMemcpyHostToDevice));
    return HTM_OK;
}

int htm_chains_create(htm_forward *h, const htm_chains_init *init, htm_chains **out)
{
    if (!out) return fail(HTM_EINVAL, "out is NULL");
    *out = nullptr;
    if (!h || !init) return fail(HTM_EINVAL, "NULL argument");
    const int nc = init->n_chains;
    if (nc < 1 || nc > kMaxChains) return fail(HTM_EINVAL, "n_chains must be in 1..%d", kMaxChains);
    if (init->n_procs < 1 || init->rank < 0 || init->rank >= init->n_procs) return fail(HTM_EINVAL, "bad rank/n_procs");
    if (init->n_interval < 1) return fail(HTM_EINVAL, "n_interval must be >= 1");
    if (!init->temp) return fail(HTM_EINVAL, "temp is NULL");
    if (h->nch == 4) {
        // k_step holds two candidate positions in registers; 256 stations exceed the budget -> generic path
    }
    HIPCHK(hipSetDevice(h->device));
    htm_chains *hc = new htm_chains();
    hc->fwd = h;
    auto cleanup = [&](int code) { htm_chains_destroy(hc); return code; };
    int rc;
    ChainsDev &d = hc->dev;
    d.n_chains = nc; d.n_procs = init->n_procs; d.rank = init->rank; d.S = h->S; d.E = h->E;
    {
        // [vs | t_corr | qs | a_corr | hypo], the order of the proposal types (chain_pass computes these offsets too)
        const size_t S = (size_t)h->S, E = (size_t)h->E, n = (size_t)nc;
        const size_t total = 2 * n + 2 * n * S + 3 * E * n;
        if ((rc = dev_alloc(hc->pool, &d.xall, total)) || (rc = dev_alloc(hc->pool, &d.muall, total)) ||
            (rc = dev_alloc(hc->pool, &d.sgall, total)) || (rc = dev_alloc(hc->pool, &d.stall, total)) ||
            (rc = dev_alloc(hc->pool, &d.ptall, total))) return cleanup(rc);
        if ((rc = upload_model(hc, d.vs, init->vs, 1, nc, 0, "vs"))) return cleanup(rc);
        if ((rc = upload_model(hc, d.tc, init->t_corr, h->S, nc, n, "t_corr"))) return cleanup(rc);
        if ((rc = upload_model(hc, d.qs, init->qs, 1, nc, n + n * S, "qs"))) return cleanup(rc);
        if ((rc = upload_model(hc, d.ac, init->a_corr, h->S, nc, 2 * n + n * S, "a_corr"))) return cleanup(rc);
        if ((rc = upload_model(hc, d.hypo, init->hypo, 3 * h->E, nc, 2 * n + 2 * n * S, "hypo"))) return cleanup(rc);
        {
            std::vector<double> sg(total), r2(total);
            HIPCHK(hipMemcpy(sg.data(), d.sgall, total * sizeof(double), hipMemcpyDeviceToHost));
            for (size_t k = 0; k < total; ++k) r2[k] = 1.0 / (2.0 * sg[k] * sg[k]);
            if ((rc = dev_upload(hc->pool, &d.rs2all, r2.data(), total))) return cleanup(rc);
            // the packed records the chain steps read (PriorRec), and whether every chain has chain 0's
            std::vector<double> mu(total), stp(total);
            std::vector<int32_t> pt(total);
            HIPCHK(hipMemcpy(mu.data(), d.muall, total * sizeof(double), hipMemcpyDeviceToHost));
            HIPCHK(hipMemcpy(stp.data(), d.stall, total * sizeof(double), hipMemcpyDeviceToHost));
            HIPCHK(hipMemcpy(pt.data(), d.ptall, total * sizeof(int32_t), hipMemcpyDeviceToHost));
            std::vector<PriorRec> pr(total);
            for (size_t k = 0; k < total; ++k) { pr[k].mu = mu[k]; pr[k].rs2 = r2[k]; pr[k].step = stp[k]; pr[k].ptype = pt[k]; pr[k].pad = 0; }
            bool same = true;
            const size_t goffs[5] = {0, n, n + n * S, 2 * n + n * S, 2 * n + 2 * n * S}, gnx[5] = {1, S, 1, S, 3 * E};
            for (int gi = 0; gi < 5 && same; ++gi)
                for (size_t c = 1; c < n && same; ++c)
                    same = memcmp(&pr[goffs[gi] + c * gnx[gi]], &pr[goffs[gi]], gnx[gi] * sizeof(PriorRec)) == 0;
            if (const char *e = getenv("HTM_PRIOR_SAME")) { if (e[0] == '0') same = false; }      // (diagnostics: every chain reads its own records)
            PriorRec *dp = nullptr;
            if ((rc = dev_upload(hc->pool, &dp, pr.data(), total))) return cleanup(rc);
            d.prior = dp; d.prior_same = same ? 1 : 0;
        }
        d.rayleigh14 = 0;
        auto any_rayleigh = [](const htm_model_init &m, size_t cnt) {
            if (!m.prior_type) return false;
            for (size_t k = 0; k < cnt; ++k) if (m.prior_type[k] == 1) return true;
            return false;
        };
        if (any_rayleigh(init->vs, n) || any_rayleigh(init->qs, n) || any_rayleigh(init->t_corr, n * S) ||
            any_rayleigh(init->a_corr, n * S)) d.rayleigh14 = 1;
    }
    if ((rc = dev_upload(hc->pool, &d.temp, init->temp, nc))) return cleanup(rc);
    std::vector<double> L0(nc, -9.e+300);   // cls_mcmc.f90:88
    if ((rc = dev_upload(hc->pool, &d.L, L0.data(), nc))) return cleanup(rc);
    std::vector<int32_t> z7(7 * (size_t)nc, 0);
    if ((rc = dev_upload(hc->pool, &d.n_propose, z7.data(), z7.size()))) return cleanup(rc);
    if ((rc = dev_upload(hc->pool, &d.n_accept, z7.data(), z7.size()))) return cleanup(rc);
    // proposal probabilities, cls_mcmc.f90:91-106, and the cumulative sums of :139-153 in the same order
    const double p_vs = init->solve_vs ? 0.025 : 0.0, p_tc = init->solve_t_corr ? 0.025 : 0.0;
    const double p_qs = init->solve_qs ? 0.025 : 0.0, p_ac = init->solve_a_corr ? 0.025 : 0.0;
    d.th1 = p_vs; d.th2 = p_vs + p_tc; d.th3 = p_vs + p_tc + p_qs; d.th4 = p_vs + p_tc + p_qs + p_ac;
    d.n_burn = init->n_burn; d.n_interval = init->n_interval;
    d.n_wg = h->n_wg;
    if ((rc = dev_alloc(hc->pool, &d.prop, nc))) return cleanup(rc);
    if ((rc = dev_alloc(hc->pool, &d.desc, 1))) return cleanup(rc);
    HIPCHK(hipMemset(d.desc, 0, sizeof(FullDesc)));
    int worker_cap = 250;          // of 256 CUs: the master's, and a few to spare
    if (const char *e = getenv("HTM_WORKER_CAP")) worker_cap = std::max(1, std::min(255, atoi(e)));      // (tuning)
    hc->worker_cap = worker_cap;
    d.n_workers = worker_blocks(h->E, 8, worker_cap);
    if (const char *e = getenv("HTM_MAX_WORKERS")) d.n_workers = std::max(1, std::min(d.n_workers, atoi(e)));   // GPUs shared between ranks
    if ((rc = dev_alloc(hc->pool, &d.partial, (size_t)nc * std::max(h->n_wg, 256)))) return cleanup(rc);      // (<= 256 workers whatever the launch shape)
    if ((rc = dev_alloc(hc->pool, &d.ps, 1))) return cleanup(rc);
    HIPCHK(hipMemset(d.ps, 0, sizeof(PSync)));
    {
        // hand-off geometry (tuning knobs; defaults measured on MI355X)
        auto env_int = [](const char *name, int dflt, int lo, int hi) {
            const char *e = getenv(name);
            const int v = e ? atoi(e) : dflt;
            return std::max(lo, std::min(hi, v));
        };
        d.slot_rep = env_int("HTM_SLOT_REPLICAS", 1, 1, kMaxSlotReplicas);
        d.slot_stride = env_int("HTM_SLOT_STRIDE", 4096, kMaxChains * kGranPerSlot * 8, 1 << 22) / 8;     // bytes -> words
        d.pgran_stride = env_int("HTM_PGRAN_STRIDE", 16, 16, 4096) / 8;
        d.npoll = env_int("HTM_NPOLL", 1, 1, 3);
    }
    if ((rc = dev_alloc(hc->pool, &d.slots, (size_t)d.slot_rep * d.slot_stride))) return cleanup(rc);
    HIPCHK(hipMemset(d.slots, 0, (size_t)d.slot_rep * d.slot_stride * sizeof(unsigned long long)));
    {
        unsigned long long *mbw = nullptr;
        if ((rc = dev_alloc(hc->pool, &mbw, sizeof(MbShared) / sizeof(unsigned long long)))) return cleanup(rc);
        d.mb = reinterpret_cast<MbShared *>(mbw);
    }
    if ((rc = dev_alloc(hc->pool, &d.prev_mid, (size_t)nc))) return cleanup(rc);
    HIPCHK(hipMemset(d.prev_mid, 0, (size_t)nc * sizeof(int)));
    if ((rc = dev_alloc(hc->pool, &d.lo_gran, (size_t)nc * 16))) return cleanup(rc);
    HIPCHK(hipMemset(d.lo_gran, 0, (size_t)nc * 16 * sizeof(unsigned long long)));
    if ((rc = dev_alloc(hc->pool, &d.pgran, (size_t)nc * 256 * d.pgran_stride))) return cleanup(rc);
    HIPCHK(hipMemset(d.pgran, 0, (size_t)nc * 256 * d.pgran_stride * sizeof(unsigned long long)));
    {
        const char *env = getenv("HTM_PERSIST");
        hc->persist = !(env && env[0] == '0');
    }
    if ((rc = dev_alloc(hc->pool, &d.swap_rec, 4 + 2 * (size_t)nc))) return cleanup(rc);
    HIPCHK(hipMemset(d.prop, 0, nc * sizeof(Proposal)));
    HIPCHK(hipMemset(d.swap_rec, 0, (4 + 2 * (size_t)nc) * sizeof(double)));

    hc->rec_len = 3 * h->E + 2 * h->S + 2;
    d.cap_lik = init->lik_capacity > 0 ? init->lik_capacity : 1 << 16;
    if (init->sample_capacity > 0) d.cap_smp = init->sample_capacity;
    else d.cap_smp = (int)std::max<size_t>(4 * (size_t)nc, std::min<size_t>(4096, (size_t(128) << 20) / (hc->rec_len * sizeof(double))));
    if (d.cap_lik < 2 * nc || d.cap_smp < 2 * nc) return cleanup(fail(HTM_EINVAL, "record capacities must be >= 2*n_chains"));
    if ((rc = dev_alloc(hc->pool, &d.lik_iter, d.cap_lik))) return cleanup(rc);
    if ((rc = dev_alloc(hc->pool, &d.lik_chain, d.cap_lik))) return cleanup(rc);
    if ((rc = dev_alloc(hc->pool, &d.lik_val, d.cap_lik))) return cleanup(rc);
    if ((rc = dev_alloc(hc->pool, &d.smp_iter, d.cap_smp))) return cleanup(rc);
    if ((rc = dev_alloc(hc->pool, &d.smp_chain, d.cap_smp))) return cleanup(rc);
    if ((rc = dev_alloc(hc->pool, &d.smp_data, (size_t)d.cap_smp * hc->rec_len))) return cleanup(rc);
    d.slog_i = nullptr; d.slog_d = nullptr;
    { const char *e = getenv("HTM_DEBUG_NO_DROP"); d.dbg = (e && atoi(e) != 0) ? 1 : 0; }
    { const char *e = getenv("HTM_XCHG_TIMEOUT_MS"); const double ms = e ? atof(e) : 20000.0; d.xwait_ticks = (unsigned long long)(std::max(1.0, ms) * 1.0e5); }
    { const char *e = getenv("HTM_DEBUG_XCHG_FAIL_ITER"); d.dbg_xfail_iter = e ? atoi(e) : 0; }
    { const char *e = getenv("HTM_XOWN"); d.xown = (e && e[0] == '0') ? 0 : 1; }
    if ((rc = dev_alloc(hc->pool, &d.diag, 32))) return cleanup(rc);
    HIPCHK(hipMemset(d.diag, 0, 32 * sizeof(unsigned long long)));
    d.stamps = nullptr;
#ifdef HTM_STAMPS
    if ((rc = dev_alloc(hc->pool, &d.stamps, 128 + 2 * 8192))) return cleanup(rc);      // (+ the event trace of htm_pipe.hpp)
    HIPCHK(hipMemset(d.stamps, 0, (128 + 2 * 8192) * sizeof(unsigned long long)));
#endif

    // random-stream service
    hc->th[0] = d.th1; hc->th[1] = d.th2; hc->th[2] = d.th3; hc->th[3] = d.th4;
    for (int k = 0; k < 4; ++k) hc->init_state[k] = init->rng_state[k];
    hc->cap = 1 << 20;     // stream positions kept in HBM (~100 B each)
    if (const char *e = getenv("HTM_STREAM_CAP")) {          // power of two >= 2^17 (tests: ring wrap-around in short runs)
        long long v = atoll(e), c2 = 1 << 17;
        while (c2 < v && c2 < (1ll << 24)) c2 <<= 1;
        hc->cap = c2;
    }
    {
        StreamDev &sd = d.stream;
        const size_t n = (size_t)hc->cap;
        sd.mask = hc->cap - 1;
        if ((rc = dev_alloc(hc->pool, &sd.raw, n))) return cleanup(rc);
        if ((rc = dev_alloc(hc->pool, &sd.U, n))) return cleanup(rc);
        if ((rc = dev_alloc(hc->pool, &sd.LOGU, n))) return cleanup(rc);
        if ((rc = dev_alloc(hc->pool, &sd.G, n))) return cleanup(rc);
        if ((rc = dev_alloc(hc->pool, &sd.pg, n))) return cleanup(rc);
        if ((rc = dev_alloc(hc->pool, &sd.pr, n))) return cleanup(rc);
        if ((rc = dev_alloc(hc->pool, &sd.plogr, n))) return cleanup(rc);
        if ((rc = dev_alloc(hc->pool, &sd.dec, n))) return cleanup(rc);
        if ((rc = dev_alloc(hc->pool, &sd.hop, n * kHops))) return cleanup(rc);
        if ((rc = dev_alloc(hc->pool, &sd.sw, n))) return cleanup(rc);
        {
            uint32_t g8[8] = {hc->init_state[0], hc->init_state[1], hc->init_state[2], hc->init_state[3], 0, 0, 0, 0};
            if ((rc = dev_upload(hc->pool, &sd.gen, g8, 8))) return cleanup(rc);
            hc->gen_par = 0;
            const std::vector<Mat128> &P = xs_powers();
            std::vector<u32x4> jt((size_t)kJumpLevels * 128);
            for (int b = 0; b < kJumpLevels; ++b)
                for (int j = 0; j < 128; ++j) {
                    const Bits128 &c = P[6 + b][j];
                    jt[(size_t)b * 128 + j] = u32x4{c.w[0], c.w[1], c.w[2], c.w[3]};
                }
            if ((rc = dev_upload(hc->pool, &hc->d_jump, jt.data(), jt.size()))) return cleanup(rc);
        }
        const long long zero = 0;
        if ((rc = dev_upload(hc->pool, &sd.hop_end, &zero, 1))) return cleanup(rc);
    }
    {
        hipError_t e1 = hipStreamCreateWithFlags(&hc->side, hipStreamNonBlocking);
        hipError_t e2 = hipEventCreateWithFlags(&hc->ev_side, hipEventDisableTiming);
        if (e1 != hipSuccess || e2 != hipSuccess) return cleanup(fail(HTM_EHIP, "side stream/event creation failed"));
    }

    Ctrl c{};
    c.spos = 0;
    c.stage = ST_IDLE;
    hc->h_ctrl = c;
    if ((rc = dev_upload(hc->pool, &d.ctrl, &c, 1))) return cleanup(rc);

    hc->nw = std::min(nc, 8);
    // stream window: a chain step draws <= 6 numbers, select_pair/judge_swap a few more (cls_parallel.f90:226-230)
    hc->wmax = ((6 * nc + 16 + 63) / 64) * 64;
    // LDS of the master: the stream window -- two iterations + their swaps ahead where it fits (role P sends orders two
    // iterations ahead), else one -- and the mirror of (vs, t_corr, qs, a_corr) x all chains + their step sizes that
    // role P reads (without it no orders are sent ahead).  Per ring position: U, LOGU, pg, pr, plogr (5 doubles), dec,
    // sw (int4), hop (kHops ints).
    const size_t lds_fixed = ((sizeof(FlowShared) + 15) & ~size_t(15)) + 3 * (size_t)h->S * sizeof(double) + kGathStage * sizeof(double);
    const size_t lds_pos = 5 * sizeof(double) + 2 * sizeof(int4) + kHops * sizeof(int);
    const size_t mir = 2 * (size_t)nc + 2 * (size_t)nc * h->S;
    const size_t lds_cap = 156 * 1024;
    // (sized by what an iteration can really draw, 6 per chain step + the swap's, not by wmax's rounding to 64: at 16 chains
    // the two-iteration window then fits a 512-position ring instead of 1024 -- half the LDS, so the mirror fits too)
    const int wdraw = 6 * nc + 16;
    // Several master workgroups (9..16 chains, k_mcmc<.., 7>): a workgroup's window is kept by the wave of its FIRST chain, and
    // that chain can be two iterations ahead of a chain of its own workgroup that sat in a full evaluation (its turn asks the
    // later chains for their checks of the iteration before only).  The late chain then adopts an anchor that lies up to three
    // iterations behind the keeper's position: the ring must hold the look-ahead (3 wd + 24) AND three iterations + the spread of
    // eight chains behind it, or the anchor's table entries have been overwritten by positions one ring further on (the
    // mismatch of profiles/r04_z_mb_open_issue.txt: a base computed from evicted entries).  With two chains per wave -- one
    // workgroup -- the keeper cannot get further than one iteration ahead, and up to eight chains leave the ring mostly empty.
    const int mb_need = (3 * wdraw + 24) + (3 * wdraw + 16) + 6 * 8 + 16;
    bool mb_want = nc > 8 && nc <= 16 && d.n_procs == 1 && (h->nch == 1 || h->nch == 2) && !(getenv("HTM_MB") && getenv("HTM_MB")[0] == '0');
    // The same rule for the free-running loop of ONE workgroup with a wave per chain (up to eight chains; single rank and lock-step
    // ranks): the keeper's wave runs chain 0 only and gets as far ahead of a late chain.  (More than eight chains on one
    // workgroup: the keeper's wave has two chains and meets every other chain's check within an iteration.)  Rings of 256
    // positions -- 4 and 5 chains -- were too short for it by this count; HTM_RING_SLACK=0 keeps the old sizes (A/B only).
    const bool slack = !(getenv("HTM_RING_SLACK") && getenv("HTM_RING_SLACK")[0] == '0');
    const int fr_need = (nc <= 8 && slack) ? (3 * wdraw + 24) + (3 * wdraw + 16) : 0;
    auto ring_for = [&](int look) { int r = 256; while (r < look * wdraw + 64 || r < fr_need || (mb_want && r < mb_need)) r *= 2; return r; };
    hc->dev.mirror_n = 0; hc->dev.mirror_steps = 0;
    for (int pass = 0; pass < 2; ++pass) {
        hc->ring_size = ring_for(2);
        // preference: long window + values + step sizes, long window + values, short window + both, short + values, nothing
        const int looks[4] = {4, 4, 2, 2}, steps[4] = {1, 0, 1, 0};
        for (int k = 0; k < 4; ++k)
            if (lds_fixed + (size_t)ring_for(looks[k]) * lds_pos + (1 + steps[k]) * mir * sizeof(double) <= lds_cap) {
                hc->ring_size = ring_for(looks[k]); hc->dev.mirror_n = (int)mir; hc->dev.mirror_steps = steps[k];
                break;
            }
        if (hc->dev.mirror_n > 0 || !mb_want) break;
        mb_want = false;      // (the longer ring does not fit beside the mirror: one workgroup, the usual ring)
    }
    hc->step_smem = lds_fixed + (size_t)hc->ring_size * lds_pos + (1 + hc->dev.mirror_steps) * (size_t)hc->dev.mirror_n * sizeof(double);
    if (hc->step_smem > lds_cap) return cleanup(fail(HTM_EINVAL, "n_chains / n_sta too large for k_step's LDS budget"));
    if (hc->step_smem > 48 * 1024) {
        const void *fn = h->dev.fp32 ? (h->nch == 1 ? (const void *)k_step<1, true> : (const void *)k_step<2, true>)
                         : h->nch == 1 ? (const void *)k_step<1> : h->nch == 2 ? (const void *)k_step<2> : (const void *)k_step<0>;
        HIPCHK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)hc->step_smem));
    }
    if (hipEventCreateWithFlags(&hc->ev_wd, hipEventDisableTiming) != hipSuccess) return cleanup(fail(HTM_EHIP, "hipEventCreate failed"));
    if (hipEventCreate(&hc->ev0) != hipSuccess || hipEventCreate(&hc->ev1) != hipSuccess)
        return cleanup(fail(HTM_EHIP, "hipEventCreate failed"));
    hc->dev_np = hc->dev;                       // same state; partial sums laid out per k_full tile
    if (hc->step_smem < 1024) hc->step_smem = 1024;
    if (hc->persist) {
        // Master and workers of a k_mcmc launch wait for each other, so every block must be RESIDENT: never ask for more
        // worker blocks than the device can hold next to the master (a partitioned or CU-masked GPU has fewer CUs; every
        // block carries the master's LDS size).  Workers take events round-robin, so fewer of them only take longer.
#define HTM_MCMC_FN(K) (h->dev.fp32 ? (h->nch == 1 ? (const void *)k_mcmc<1, true, K> : (const void *)k_mcmc<2, true, K>)                  \
                        : h->nch == 1 ? (const void *)k_mcmc<1, false, K> : h->nch == 2 ? (const void *)k_mcmc<2, false, K> : (const void *)k_mcmc<0, false, K>)
        std::vector<const void *> fns = {HTM_MCMC_FN(0), HTM_MCMC_FN(1), HTM_MCMC_FN(2), HTM_MCMC_FN(3), HTM_MCMC_FN(4)};
#undef HTM_MCMC_FN
        // (several master workgroups, k_mcmc<.., 7>: one or two stations per lane)
        if (h->nch == 1) fns.push_back(h->dev.fp32 ? (const void *)k_mcmc<1, true, 7> : (const void *)k_mcmc<1, false, 7>);
        else if (h->nch == 2) fns.push_back(h->dev.fp32 ? (const void *)k_mcmc<2, true, 7> : (const void *)k_mcmc<2, false, 7>);
        if (hc->step_smem > 48 * 1024)
            for (const void *g : fns) HIPCHK(hipFuncSetAttribute(g, hipFuncAttributeMaxDynamicSharedMemorySize, (int)hc->step_smem));
        // the residency bound holds for whichever main loop gets launched: the smallest of the instantiations' occupancies
        int per_cu = 1 << 20, n_cu = 0;
        for (const void *g : fns) {
            int pc = 0;
            HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&pc, g, 512, hc->step_smem));
            per_cu = std::min(per_cu, pc);
        }
        HIPCHK(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, h->device));
        hc->blocks_fit = (long)per_cu * n_cu;
        long room = (long)per_cu * n_cu - 1;
        // Several ranks on one GPU (more masters instead of more rounds per master: 4 ranks x 8 chains run 2.7 M steps/s where one
        // rank x 32 chains runs 1.7 M): every rank's blocks must be resident at once, so each takes its share of the CUs
        if (const char *e = getenv("HTM_RANKS_PER_GPU")) {
            const int k = atoi(e);
            const long per_xcd = (long)per_cu * n_cu / 8;      // (blocks are dealt to the 8 XCDs in turn: htm_chains_share_gpu)
            if (k > 1) room = per_xcd >= k ? 8 * (per_xcd / k) - 1 : (long)per_cu * n_cu / k - 1;
        }
        if (room < 1) hc->persist = false;      // not even one worker fits next to the master: two-kernel path
        else if (hc->dev.n_workers > room) hc->dev.n_workers = worker_blocks(h->E, 8, room);
    }
    hc->dev.n_wg = hc->dev.n_workers;           // persistent kernel: one partial per worker block
    {
        // The free-running master (htm_flow.hpp) runs the single-rank loop when its stream window fits: three iterations of
        // look-ahead + one behind + the spread of chain 0's wave over its chains (flow_step's window extension), and the LDS
        // mirror the orders are computed from.  HTM_FLOW=0 keeps the loop with barriers (step_body); so do the hand-off tests'
        // debug switches.
        const char *e = getenv("HTM_FLOW");
        const int wd = 6 * nc + 16, c_max = 8 * ((nc - 1) / 8);
        const bool window_ok = hc->persist && !(e && e[0] == '0') && d.dbg == 0 && hc->dev.mirror_n > 0 &&
                               hc->ring_size >= 4 * wd + 32 + 2 * c_max + 16;
        hc->flow = window_ok && d.n_procs == 1;
        // More than eight chains on a rank: a master workgroup for every eight (k_mcmc<.., 7>, htm_flow.hpp MbShared) instead of
        // rounds on the same eight waves.
        {
            // (HTM_MB=0: one workgroup.  The ring was sized for several above: mb_want, mb_need)
            const bool on = mb_want && hc->ring_size >= mb_need;
            // (9..16 chains: two workgroups.  More would need more of the stream window per step than the one wave of a workgroup that
            // keeps it can load -- 128 positions, an iteration of 16 chains takes ~90)
            if (hc->flow && on && nc > 8 && nc <= 16 && (h->nch == 1 || h->nch == 2)) {
                const int nb = (nc + 7) / 8;
                if (hc->blocks_fit - nb >= 1) {
                    hc->mb_blocks = nb;
                    const long room = hc->blocks_fit - nb;
                    if (hc->dev.n_workers > room) { hc->dev.n_workers = worker_blocks(h->E, 8, room); hc->dev.n_wg = hc->dev.n_workers; }
                }
            }
        }
        hc->flow_lock = window_ok && d.n_procs <= 60 && !(getenv("HTM_FLOW_LOCK") && getenv("HTM_FLOW_LOCK")[0] == '0');      // (MODE_LOCKRUN: up to 60 ranks -- a lane per rank reads its stop word, flow_xload)
        // The pipelined master (htm_pipe.hpp; HTM_PIPE=0 keeps the free-running one): front / evaluators / decider over an LDS ring
        // of iteration slots.  Needs the LDS mirror of the non-hypocentre parameters (what its evaluators read) and one or two
        // stations per lane.
        {
            const char *ep = getenv("HTM_PIPE");
            const int wdp = 6 * nc + 16;
            int ring = 512;
            while (ring < 2 * wdp + 160) ring *= 2;
            const size_t mirp = 2 * (size_t)nc + 2 * (size_t)nc * h->S;
            const size_t smem = ((sizeof(PipeShared) + 15) & ~size_t(15)) + (size_t)ring * lds_pos + (3 * (size_t)h->S + kGathStage) * sizeof(double) +
                                mirp * sizeof(double) + pipe_ring_bytes(nc);
            // (opt-in, HTM_PIPE=1 / HTM_PIPE_LOCK=1: on one CU it matches the free-running master -- 4.8 us per iteration at 1000 x 64 x 8,
            // +8 % at 16 chains, profiles/r04_pipe_*.txt -- and does not beat it; DESIGN.md 3.6 says what it is for)
            const char *epl = getenv("HTM_PIPE_LOCK");
            const bool usable = hc->persist && d.dbg == 0 && (h->nch == 1 || h->nch == 2) && smem <= lds_cap &&
                                hc->dev.mirror_n == (int)mirp;      // (the mirror is part of every loop's LDS layout: one size for all)
            hc->pipe = usable && d.n_procs == 1 && ep && ep[0] == '1';
            hc->pipe_lock = usable && epl && epl[0] == '1' &&
                            (size_t)d.n_procs * (4 + 2 * (size_t)nc) <= (size_t)kGathStage;      // (MODE_LOCKRUN: any number of ranks)
            if (hc->pipe || hc->pipe_lock) {
                hc->pipe_smem = smem; hc->pipe_ring = ring;
                int n_cu2 = 0;
                HIPCHK(hipDeviceGetAttribute(&n_cu2, hipDeviceAttributeMultiprocessorCount, h->device));
                const int pthreads = h->nch == 1 ? mcmc_threads<1, 5>() : 512;
                if (pthreads == 768) {
                    // (12-wave blocks: a worker block takes 12 events; every loop of this chain set then runs with this many blocks)
                    int nw = worker_blocks(h->E, 12, hc->worker_cap);
                    if (const char *e3 = getenv("HTM_MAX_WORKERS")) nw = std::max(1, std::min(nw, atoi(e3)));
                    if (const char *e4 = getenv("HTM_RANKS_PER_GPU")) { const int k = atoi(e4); if (k > 1) nw = std::min<long>(nw, std::max<long>(1, hc->blocks_fit / k - 1)); }
                    hc->dev.n_workers = std::min(hc->dev.n_workers, nw); hc->dev.n_wg = hc->dev.n_workers;
                }
                for (int lk = 5; lk <= 6; ++lk) {
                    const void *pfn = lk == 5 ? (h->dev.fp32 ? (h->nch == 1 ? (const void *)k_mcmc<1, true, 5> : (const void *)k_mcmc<2, true, 5>)
                                                            : (h->nch == 1 ? (const void *)k_mcmc<1, false, 5> : (const void *)k_mcmc<2, false, 5>))
                                              : (h->dev.fp32 ? (h->nch == 1 ? (const void *)k_mcmc<1, true, 6> : (const void *)k_mcmc<2, true, 6>)
                                                            : (h->nch == 1 ? (const void *)k_mcmc<1, false, 6> : (const void *)k_mcmc<2, false, 6>));
                    int pc = 0;
                    if (smem > 48 * 1024) HIPCHK(hipFuncSetAttribute(pfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
                    HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&pc, pfn, pthreads, smem));
                    hc->blocks_fit = std::min<long>(hc->blocks_fit, (long)pc * n_cu2);
                    if ((long)pc * n_cu2 - 1 < hc->dev.n_workers) { if (lk == 5) hc->pipe = false; else hc->pipe_lock = false; }      // (the launch shape was sized for the other loops: keep it)
                }
            }
        }
    }
    // first stretch of the random stream (synchronous)
    if ((rc = stream_produce(hc, 1 << 16))) return cleanup(rc);
    HIPCHK(hipStreamSynchronize(hc->side));
    *out = hc;
    return HTM_OK;
}

int htm_chains_destroy(htm_chains *hc)
{
    if (!hc) return HTM_OK;
    (void)hipSetDevice(hc->fwd->device);
    if (hc->side) (void)hipStreamSynchronize(hc->side);
    (void)hipStreamSynchronize(hc->fwd->stream);
    if (hc->ev_side) (void)hipEventDestroy(hc->ev_side);
    if (hc->side) (void)hipStreamDestroy(hc->side);
    if (hc->gexec) (void)hipGraphExecDestroy(hc->gexec);
    if (hc->graph) (void)hipGraphDestroy(hc->graph);
    for (void *p : hc->peer_maps) (void)hipIpcCloseMemHandle(p);
    if (hc->d_inbox) (void)hipFree(hc->d_inbox);
    for (void *p : hc->pool) (void)hipFree(p);
    if (hc->h_gath_pinned) (void)hipHostFree(hc->h_gath_pinned);
    if (hc->ev_wd) (void)hipEventDestroy(hc->ev_wd);
    if (hc->ev0) (void)hipEventDestroy(hc->ev0);
    if (hc->ev1) (void)hipEventDestroy(hc->ev1);
    delete hc;
    return HTM_OK;
}

static int flush_pending(htm_chains *hc)
{
    if (!hc->pending_gathered) return HTM_OK;
    const double *g = hc->pending_gathered;
    hc->pending_gathered = nullptr;
    return launch_step(hc, MODE_APPLY, hc->h_target, g);
}

// Wait for the chain kernels' stream, but never forever: every device-side wait is bounded (workers 30 s, master
// 5 s per hand-over), so a stream that has not drained after kWatchdogSeconds means the device is wedged -- report
// it instead of hanging the caller (HTM_WATCHDOG_S overrides the limit; 0 = wait without limit).
static int bounded_stream_sync(htm_chains *hc, const char *what)
{
    static const double limit = [] { const char *e = getenv("HTM_WATCHDOG_S"); return e ? atof(e) : 300.0; }();
    hipStream_t st = hc->fwd->stream;
    if (limit <= 0.0) { HIPCHK(hipStreamSynchronize(st)); return HTM_OK; }
    HIPCHK(hipEventRecord(hc->ev_wd, st));
    const auto t0 = std::chrono::steady_clock::now();
    for (long spin = 0;; ++spin) {
        const hipError_t e = hipEventQuery(hc->ev_wd);
        if (e == hipSuccess) return HTM_OK;
        if (e != hipErrorNotReady) return fail(HTM_EHIP, "%s: %s", what, hipGetErrorString(e));
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (dt > limit)
            return fail(HTM_ESTATE, "%s: the device made no progress for %.0f s (iteration %d + in-flight work, launch %llu)",
                        what, dt, hc->h_ctrl.iter_done, hc->launch_seq);
        if (spin > 2000) std::this_thread::sleep_for(std::chrono::microseconds(dt > 0.01 ? 200 : 5));
    }
}

static int read_ctrl(htm_chains *hc)
{
    // (nothing launched since the last read: the copy is current -- a driver that runs in short slices and asks for the state
    // in between would pay the copy and the stream synchronisation three times per slice)
    if (hc->ctrl_fresh && !hc->pending_gathered) return HTM_OK;
    int rc_ = flush_pending(hc);
    if (rc_) return rc_;
    HIPCHK(hipMemcpyAsync(&hc->h_ctrl, hc->dev.ctrl, sizeof(Ctrl), hipMemcpyDeviceToHost, hc->fwd->stream));
    if ((rc_ = bounded_stream_sync(hc, "waiting for the chain kernels"))) return rc_;
    hc->spos_lo = hc->spos_hi = hc->h_ctrl.spos;
    hc->ctrl_fresh = true;
    return HTM_OK;
}

static int ctrl_error(const htm_chains *hc)
{
    switch (hc->h_ctrl.err) {
    case 0: return HTM_OK;
    case -4: return fail(HTM_ESTATE, "device RNG window exhausted (iteration %d)", hc->h_ctrl.iter_done + 1);
    case -5: return fail(HTM_EOVERFLOW, "record buffer overflow in lock-step mode: call htm_chains_drain more often");
    case -6: return fail(HTM_EDESYNC, "swap records of the ranks carry different iteration numbers");
    case -8: {
        // what the chain wave that gave up was waiting for (ChainsDev::diag), the chain's order slot and how many workers answered
        const ChainsDev &d = hc->dev;
        unsigned long long dg[32] = {0}, slot[kGranPerSlot] = {0};
        char extra[900] = "";
        if (hipMemcpy(dg, d.diag, sizeof(dg), hipMemcpyDeviceToHost) == hipSuccess && dg[0] == 1 && (int)dg[1] < d.n_chains) {
            const int c = (int)dg[1];
            int answered = 0;
            std::vector<unsigned long long> pg((size_t)d.n_workers * d.pgran_stride);
            if (hipMemcpy(slot, d.slots + (size_t)c * kGranPerSlot, sizeof(slot), hipMemcpyDeviceToHost) == hipSuccess &&
                hipMemcpy(pg.data(), d.pgran + (size_t)c * d.n_workers * d.pgran_stride, pg.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess) {
                for (int k = 0; k < d.n_workers; ++k)
                    if ((unsigned)(pg[(size_t)k * d.pgran_stride] >> 32) == (unsigned)dg[2]) ++answered;
                snprintf(extra, sizeof(extra),
                         "; chain %d waited for the sums of order %08llx (%s, step type %llu index %llu at stream position %llu, %s pass); "
                         "%d of %d workers answered; the chain's order slot holds tags %08llx %08llx %08llx %08llx %08llx %08llx %08llx %08llx, launch word %llu",
                         c, dg[2], dg[3] ? (dg[4] == 2 ? "sent two iterations ahead" : "sent one iteration ahead") : "sent by the chain's own wave",
                         dg[7], dg[8], dg[6], dg[11] ? "first" : "repeated", answered, d.n_workers,
                         slot[0] >> 32, slot[1] >> 32, slot[2] >> 32, slot[3] >> 32, slot[4] >> 32, slot[5] >> 32, slot[6] >> 32, slot[7] >> 32,
                         slot[0] & 0xffffffffull);
            }
            if (dg[24]) {
                const size_t n = strlen(extra);
                snprintf(extra + n, sizeof(extra) - n, "; worker 0 put %llu orders aside (named commit not visible within 20 us)", dg[24]);
            }
            if (dg[16] == 1) {
                const size_t n = strlen(extra);
                snprintf(extra + n, sizeof(extra) - n,
                         "; worker 0 has been waiting for more than a second for the commit named by order %08llx of chain %llu: element %llu "
                         "should read %016llx, reads %016llx (order word %llx, element left out %llu)",
                         dg[18], dg[17], dg[19], dg[20], dg[21], dg[22], dg[23]);
            }
        }
        return fail(HTM_ESTATE, "persistent workers did not answer within 5 s (iteration %d)%s", hc->h_ctrl.iter_done + 1, extra);
    }
    case -7: return fail(HTM_ESTATE, "random stream underrun in lock-step mode (iteration %d)", hc->h_ctrl.iter_done + 1);
    case -10: return fail(HTM_ESTATE, "swap records of another rank did not arrive within %.1f s (after iteration %d; HTM_XCHG_TIMEOUT_MS)",
                          (double)hc->dev.xwait_ticks * 1.0e-8, hc->h_ctrl.iter_done);
    case -11: return fail(HTM_ESTATE, "another rank reported a failure (iteration %d)", hc->h_ctrl.iter_done + 1);
    case -15: return fail(HTM_EDESYNC, "a swap record in the inbox was overwritten before it was read (after iteration %d): the ranks are more than the ring's depth apart", hc->h_ctrl.iter_done);
    case -12: return fail(HTM_ESTATE, "a chain wave of the master waited 5 s for another chain's step (its check, its commit before a swap, or the "
                          "rank's own header) after iteration %d: a wave has stopped making progress", hc->h_ctrl.iter_done);
    case -13: return fail(HTM_ESTATE, "the master's window of the random stream did not reach a step's position within 5 s (after iteration %d)",
                          hc->h_ctrl.iter_done);
    case -14: return fail(HTM_ESTATE, "an evaluator of the pipelined master waited 5 s for a proposal record (after iteration %d)", hc->h_ctrl.iter_done);
    case -16: {
        unsigned long long dg[8] = {0};
        (void)hipMemcpy(dg, hc->dev.diag, sizeof(dg), hipMemcpyDeviceToHost);
        return fail(HTM_ESTATE, "several master workgroups: chain %llu waited 5 s in its turn of iteration %llu for the chains %#llx (its epoch %llu, "
                    "the epoch in memory %llu, first awaited word %#llx)", dg[1], dg[2], dg[3], dg[4], dg[5], dg[6]);
    }
    case -17: case -18: case -19: case -20: case -21:
        return fail(HTM_ESTATE, "several master workgroups: a wait for %s gave up after 5 s (after iteration %d)",
                    hc->h_ctrl.err == -17 ? "the swap record of the iteration before" : hc->h_ctrl.err == -18 ? "the swap partner's record"
                    : hc->h_ctrl.err == -19 ? "the anchor the wave had just written" : "the last iteration's records at the end of the launch", hc->h_ctrl.iter_done);
    default: return fail(HTM_ESTATE, "device error flag %d", hc->h_ctrl.err);
    }
}

// move device record buffers to the host vectors; requires h_ctrl to be current and the stream idle
static int drain_records(htm_chains *hc)
{
    const ChainsDev &d = hc->dev;
    const int nl = hc->h_ctrl.n_lik, ns = hc->h_ctrl.n_smp;
    if (nl > 0) {
        const size_t o = hc->lik_iter.size();
        hc->lik_iter.resize(o + nl); hc->lik_chain.resize(o + nl); hc->lik_val.resize(o + nl);
        HIPCHK(hipMemcpy(hc->lik_iter.data() + o, d.lik_iter, nl * sizeof(int32_t), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(hc->lik_chain.data() + o, d.lik_chain, nl * sizeof(int32_t), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(hc->lik_val.data() + o, d.lik_val, nl * sizeof(double), hipMemcpyDeviceToHost));
    }
    if (ns > 0) {
        const size_t o = hc->smp_iter.size();
        hc->smp_iter.resize(o + ns); hc->smp_chain.resize(o + ns); hc->smp_data.resize((o + ns) * hc->rec_len);
        HIPCHK(hipMemcpy(hc->smp_iter.data() + o, d.smp_iter, ns * sizeof(int32_t), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(hc->smp_chain.data() + o, d.smp_chain, ns * sizeof(int32_t), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(hc->smp_data.data() + o * hc->rec_len, d.smp_data, (size_t)ns * hc->rec_len * sizeof(double),
                         hipMemcpyDeviceToHost));
    }
    if (hc->flow || hc->flow_lock) {
        // the free-running master hands out record slots as its waves get to them: put the drained records into the
        // reference's order, iteration by iteration and chain by chain (hypo_tremor_mcmc.f90:270-280)
        auto order = [](size_t n, const int32_t *it, const int32_t *ch) {
            std::vector<size_t> idx(n);
            for (size_t k = 0; k < n; ++k) idx[k] = k;
            std::stable_sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return it[a] != it[b] ? it[a] < it[b] : ch[a] < ch[b]; });
            return idx;
        };
        if (nl > 1) {
            const size_t o = hc->lik_iter.size() - nl;
            const std::vector<size_t> idx = order(nl, hc->lik_iter.data() + o, hc->lik_chain.data() + o);
            std::vector<int32_t> it(nl), ch(nl); std::vector<double> v(nl);
            for (int k = 0; k < nl; ++k) { it[k] = hc->lik_iter[o + idx[k]]; ch[k] = hc->lik_chain[o + idx[k]]; v[k] = hc->lik_val[o + idx[k]]; }
            std::copy(it.begin(), it.end(), hc->lik_iter.begin() + o); std::copy(ch.begin(), ch.end(), hc->lik_chain.begin() + o);
            std::copy(v.begin(), v.end(), hc->lik_val.begin() + o);
        }
        if (ns > 1) {
            const size_t o = hc->smp_iter.size() - ns, rl = hc->rec_len;
            const std::vector<size_t> idx = order(ns, hc->smp_iter.data() + o, hc->smp_chain.data() + o);
            std::vector<int32_t> it(ns), ch(ns); std::vector<double> v((size_t)ns * rl);
            for (int k = 0; k < ns; ++k) {
                it[k] = hc->smp_iter[o + idx[k]]; ch[k] = hc->smp_chain[o + idx[k]];
                std::copy(hc->smp_data.begin() + (o + idx[k]) * rl, hc->smp_data.begin() + (o + idx[k] + 1) * rl, v.begin() + (size_t)k * rl);
            }
            std::copy(it.begin(), it.end(), hc->smp_iter.begin() + o); std::copy(ch.begin(), ch.end(), hc->smp_chain.begin() + o);
            std::copy(v.begin(), v.end(), hc->smp_data.begin() + o * rl);
        }
    }
    if (nl > 0 || ns > 0 || hc->h_ctrl.stop) {   // stop: 1 = record buffers full, 2 = stream underrun
        hc->h_ctrl.n_lik = 0; hc->h_ctrl.n_smp = 0; hc->h_ctrl.stop = 0;
        // stop, n_lik, n_smp are consecutive ints in Ctrl
        HIPCHK(hipMemcpy(&hc->dev.ctrl->stop, &hc->h_ctrl.stop, 3 * sizeof(int), hipMemcpyHostToDevice));
    }
    return HTM_OK;
}

static int build_graph(htm_chains *hc)
{
    if (hc->gexec) return HTM_OK;
    htm_forward *h = hc->fwd;
    hipStream_t cap = nullptr;
    HIPCHK(hipStreamCreateWithFlags(&cap, hipStreamNonBlocking));
    hipStream_t saved = h->stream;
    h->stream = cap;
    hipError_t e = hipStreamBeginCapture(cap, hipStreamCaptureModeRelaxed);
    int rc = HTM_OK;
    if (e != hipSuccess) rc = fail(HTM_EHIP, "hipStreamBeginCapture: %s", hipGetErrorString(e));
    const FullJob jb = chain_full_job(hc);
    for (int k = 0; k < hc->pairs && rc == HTM_OK; ++k) {
        rc = launch_step(hc, MODE_RUN, -1, nullptr);
        if (rc == HTM_OK) rc = launch_full(h, jb, hc->dev.n_chains);
    }
    hipGraph_t g = nullptr;
    e = hipStreamEndCapture(cap, &g);
    h->stream = saved;
    if (rc == HTM_OK && e != hipSuccess) rc = fail(HTM_EHIP, "hipStreamEndCapture: %s", hipGetErrorString(e));
    if (rc == HTM_OK) {
        hc->graph = g;
        e = hipGraphInstantiate(&hc->gexec, g, nullptr, nullptr, 0);
        if (e != hipSuccess) rc = fail(HTM_EHIP, "hipGraphInstantiate: %s", hipGetErrorString(e));
    }
    (void)hipStreamDestroy(cap);
    return rc;
}

// Single-rank main loop on the persistent kernel: one k_mcmc launch runs until the target, a full record
// buffer or the end of the produced random stream; no graph, no per-hand-over launches.
static int run_persistent(htm_chains *hc, int n_iter)
{
    htm_forward *h = hc->fwd;
    int rc;
    hc->h_target = hc->h_ctrl.iter_done + n_iter;
    hc->run_full0 = hc->h_ctrl.n_full_evals; hc->run_part0 = hc->h_ctrl.n_partial_evals;
    hc->last_graph_launches = 0;
    HIPCHK(hipEventRecord(hc->ev0, h->stream));
    while (hc->h_ctrl.iter_done < hc->h_target) {
        // a launch may run as far as the produced stream reaches
        const long long fed = (hc->n_hop - hc->h_ctrl.spos) / (6 * hc->dev.n_chains + 4) - 2;
        const int target = (int)std::min<long long>(hc->h_target, hc->h_ctrl.iter_done + std::max<long long>(1, fed));
        if ((rc = launch_mcmc(hc, MODE_RUN, target, nullptr))) return rc;
        hc->last_graph_launches += 1;
        // keep the random stream about half a ring ahead (asynchronous, side stream), while the launch runs
        {
            const long long ahead = hc->n_hop - hc->h_ctrl.spos;
            if (ahead < hc->cap / 2 && (rc = stream_produce(hc, std::min<long long>(hc->cap / 2 - ahead + 4096, 1 << 18)))) return rc;
        }
        if ((rc = read_ctrl(hc))) return rc;
        if ((rc = ctrl_error(hc))) return rc;
        if (hc->h_ctrl.stop == 2) HIPCHK(hipStreamSynchronize(hc->side));
        if (hc->h_ctrl.stop) { if ((rc = drain_records(hc))) return rc; }
    }
    HIPCHK(hipEventRecord(hc->ev1, h->stream));
    HIPCHK(hipEventSynchronize(hc->ev1));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, hc->ev0, hc->ev1));
    hc->last_device_us = 1000.0 * ms;
    hc->last_full = hc->h_ctrl.n_full_evals - hc->run_full0;
    hc->last_part = hc->h_ctrl.n_partial_evals - hc->run_part0;
    return drain_records(hc);
}

int htm_chains_run(htm_chains *hc, int n_iter)
{
    if (!hc) return fail(HTM_EINVAL, "NULL handle");
    if (n_iter < 0) return fail(HTM_EINVAL, "n_iter must be >= 0");
    if (hc->dev.n_procs != 1)
        return fail(HTM_ESTATE, "htm_chains_run is the single-rank driver; use step_begin/step_end for n_procs > 1");
    htm_forward *h = hc->fwd;
    HIPCHK(hipSetDevice(h->device));
    int rc = read_ctrl(hc);
    if (rc) return rc;
    if (hc->h_ctrl.stage == ST_WAIT_SWAP) return fail(HTM_ESTATE, "a lock-step iteration is in flight");
    if (hc->persist && hc->h_ctrl.stage == ST_IDLE) return run_persistent(hc, n_iter);
    if ((rc = build_graph(hc))) return rc;
    hc->h_target = hc->h_ctrl.iter_done + n_iter;
    HIPCHK(hipMemcpyAsync(&hc->dev.ctrl->iter_target, &hc->h_target, sizeof(int), hipMemcpyHostToDevice, h->stream));
    hc->run_full0 = hc->h_ctrl.n_full_evals; hc->run_part0 = hc->h_ctrl.n_partial_evals;
    hc->last_graph_launches = 0;
    HIPCHK(hipEventRecord(hc->ev0, h->stream));
    double it_per_launch = 1.5 * hc->pairs;     // refined from what the device actually achieved
    while (true) {
        const int remaining = hc->h_target - hc->h_ctrl.iter_done;
        if (remaining <= 0 && hc->h_ctrl.stage == ST_IDLE) break;
        int g = (int)(0.8 * remaining / it_per_launch);
        // no more launches than the produced random stream can feed (an iteration draws < 6*n_chains + 4)
        const long long fed = (hc->n_hop - hc->h_ctrl.spos) / (6 * hc->dev.n_chains + 4);
        g = std::min<long long>(g, (long long)(0.9 * fed / it_per_launch));
        g = std::max(1, std::min(g, 512));
        const int before = hc->h_ctrl.iter_done;
        hc->ctrl_fresh = false;
        for (int k = 0; k < g; ++k) HIPCHK(hipGraphLaunch(hc->gexec, h->stream));
        hc->last_graph_launches += g;
        if ((rc = read_ctrl(hc))) return rc;
        if ((rc = ctrl_error(hc))) return rc;
        if (hc->h_ctrl.iter_done > before && !hc->h_ctrl.stop)
            it_per_launch = std::max(1.0, double(hc->h_ctrl.iter_done - before) / g);
        // keep the random stream about two batches ahead (asynchronous, on the side stream)
        {
            const long long want = hc->cap / 2;
            const long long ahead = hc->n_hop - hc->h_ctrl.spos;
            if (ahead < want && (rc = stream_produce(hc, std::min<long long>(want - ahead + 4096, 1 << 18)))) return rc;
        }
        if (hc->h_ctrl.stop == 2) HIPCHK(hipStreamSynchronize(hc->side));   // underrun: wait for the producer
        if (hc->h_ctrl.stop) { if ((rc = drain_records(hc))) return rc; }
    }
    HIPCHK(hipEventRecord(hc->ev1, h->stream));
    HIPCHK(hipEventSynchronize(hc->ev1));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, hc->ev0, hc->ev1));
    hc->last_device_us = 1000.0 * ms;
    hc->last_full = hc->h_ctrl.n_full_evals - hc->run_full0;
    hc->last_part = hc->h_ctrl.n_partial_evals - hc->run_part0;
    return drain_records(hc);
}

int htm_chains_profile(htm_chains *hc, int n_iter, double *step_us, int *step_launches, double *full_us,
                       int *full_launches, int64_t *full_evals, int64_t *partial_evals)
{
    if (!hc || n_iter < 0) return fail(HTM_EINVAL, "bad argument");
    if (hc->dev.n_procs != 1) return fail(HTM_ESTATE, "htm_chains_profile is single-rank");
    htm_forward *h = hc->fwd;
    HIPCHK(hipSetDevice(h->device));
    int rc = read_ctrl(hc);
    if (rc) return rc;
    hc->h_target = hc->h_ctrl.iter_done + n_iter;
    HIPCHK(hipMemcpyAsync(&hc->dev.ctrl->iter_target, &hc->h_target, sizeof(int), hipMemcpyHostToDevice, h->stream));
    const long long f0 = hc->h_ctrl.n_full_evals, p0 = hc->h_ctrl.n_partial_evals;
    constexpr int B = 64;                       // launch pairs between host checks
    struct Events {             // destroyed on every return path
        std::vector<hipEvent_t> v;
        ~Events() { for (hipEvent_t e : v) if (e) (void)hipEventDestroy(e); }
    } evs;
    evs.v.assign(3 * B, nullptr);
    std::vector<hipEvent_t> &ev = evs.v;
    for (auto &e : ev) HIPCHK(hipEventCreate(&e));
    const FullJob jb = chain_full_job(hc);
    double s_us = 0.0, f_us = 0.0;
    int s_n = 0, f_n = 0;
    while (true) {
        if (hc->h_target - hc->h_ctrl.iter_done <= 0 && hc->h_ctrl.stage == ST_IDLE) break;
        for (int k = 0; k < B; ++k) {
            HIPCHK(hipEventRecord(ev[3 * k], h->stream));
            if ((rc = launch_step(hc, MODE_RUN, -1, nullptr))) return rc;
            HIPCHK(hipEventRecord(ev[3 * k + 1], h->stream));
            if ((rc = launch_full(h, jb, hc->dev.n_chains))) return rc;
            HIPCHK(hipEventRecord(ev[3 * k + 2], h->stream));
        }
        if ((rc = read_ctrl(hc))) return rc;
        if ((rc = ctrl_error(hc))) return rc;
        if (hc->n_hop - hc->h_ctrl.spos < 16384 && (rc = stream_produce(hc, 16384))) return rc;
        if (hc->h_ctrl.stop == 2) HIPCHK(hipStreamSynchronize(hc->side));
        for (int k = 0; k < B; ++k) {
            float a = 0.f, b = 0.f;
            HIPCHK(hipEventElapsedTime(&a, ev[3 * k], ev[3 * k + 1]));
            HIPCHK(hipEventElapsedTime(&b, ev[3 * k + 1], ev[3 * k + 2]));
            s_us += 1000.0 * a; f_us += 1000.0 * b; s_n++; f_n++;
        }
        if (hc->h_ctrl.stop && (rc = drain_records(hc))) return rc;
    }
    if (step_us) *step_us = s_us;
    if (step_launches) *step_launches = s_n;
    if (full_us) *full_us = f_us;
    if (full_launches) *full_launches = f_n;
    if (full_evals) *full_evals = hc->h_ctrl.n_full_evals - f0;
    if (partial_evals) *partial_evals = hc->h_ctrl.n_partial_evals - p0;
    return drain_records(hc);
}

int htm_chains_step_begin(htm_chains *hc)
{
    if (!hc) return fail(HTM_EINVAL, "NULL handle");
    htm_forward *h = hc->fwd;
    HIPCHK(hipSetDevice(h->device));
    hc->h_target += 1;
    int rc;
    // an iteration consumes at most wmax draws; keep the produced stream safely ahead of that bound
    hc->spos_hi += hc->wmax;
    if (hc->n_hop - hc->spos_hi < 4 * (long long)hc->wmax) {
        if ((rc = read_ctrl(hc))) return rc;              // exact position (also frees ring capacity)
        if ((rc = ctrl_error(hc))) return rc;
        hc->spos_hi += hc->wmax;
        if ((rc = stream_produce(hc, 1 << 16))) return rc;
        HIPCHK(hipStreamWaitEvent(h->stream, hc->ev_side, 0));
    }
    if (hc->persist) {       // whole iteration (full evaluations included) in one launch
        rc = launch_mcmc(hc, MODE_ADVANCE, hc->h_target, hc->pending_gathered);
        hc->pending_gathered = nullptr;
        return rc;
    }
    rc = launch_step(hc, MODE_ADVANCE, hc->h_target, hc->pending_gathered);   // applies the previous swap first
    hc->pending_gathered = nullptr;
    if (rc) return rc;
    if ((rc = launch_full(h, chain_full_job(hc), hc->dev.n_chains))) return rc;
    return launch_step(hc, MODE_FINISH, hc->h_target, nullptr);
}

int htm_chains_run_lockstep(htm_chains *hc, int n_iter, htm_allgather_fn allgather, void *comm, void *d_gathered)
{
    if (!hc || !allgather || !d_gathered || n_iter < 0) return fail(HTM_EINVAL, "bad argument");
    htm_forward *h = hc->fwd;
    HIPCHK(hipSetDevice(h->device));
    const size_t words = 4 + 2 * (size_t)hc->dev.n_chains;
    // records a rank may hold on the device between drains: n_chains per iteration at most
    const int drain_every = std::max(1, std::min(hc->dev.cap_lik, hc->dev.cap_smp) / (2 * hc->dev.n_chains) - 2);
    int k = 0, since_drain = 0, rc;
    // stats of this call (htm_chains_last_run_stats): evaluations counted on the device, HIP events on the kernels' stream
    if ((rc = read_ctrl(hc))) return rc;
    hc->run_full0 = hc->h_ctrl.n_full_evals; hc->run_part0 = hc->h_ctrl.n_partial_evals;
    hc->last_graph_launches = 0;
    HIPCHK(hipEventRecord(hc->ev0, h->stream));
    for (; k < n_iter; ++k) {
        rc = htm_chains_step_begin(hc);
        if (rc) return rc;
        const int st = allgather(hc->dev.swap_rec, d_gathered, words, 8 /* ncclFloat64 */, comm, h->stream);
        if (st != 0) return fail(HTM_EHIP, "all-gather of the swap records failed with status %d", st);
        if ((rc = htm_chains_step_end(hc, d_gathered))) return rc;
        if (++since_drain >= drain_every) { if ((rc = htm_chains_drain(hc))) return rc; since_drain = 0; }
        else if ((k & 511) == 511 && (rc = bounded_stream_sync(hc, "lock-step loop"))) return rc;
    }
    hc->last_graph_launches = n_iter;
    HIPCHK(hipEventRecord(hc->ev1, h->stream));
    if ((rc = read_ctrl(hc))) return rc;          // flushes the last swap, waits for the stream
    if ((rc = ctrl_error(hc))) return rc;
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, hc->ev0, hc->ev1));
    hc->last_device_us = 1000.0 * ms;
    hc->last_full = hc->h_ctrl.n_full_evals - hc->run_full0;
    hc->last_part = hc->h_ctrl.n_partial_evals - hc->run_part0;
    return HTM_OK;
}

// ---- persistent lock-step: swap records exchanged inside the launch (exchange_post / exchange_finish in htm_step.hpp) --------
static int xchg_alloc(htm_chains *hc)
{
    if (hc->d_inbox) return HTM_OK;
    ChainsDev &d = hc->dev;
    d.xg = 2 * (4 + 2 * d.n_chains) + 2;
    // (region A [2][n_procs][xg]: the barrier loop's records; region B [kXSlots][n_procs][xg]: the free-running lock-step loop's, htm_flow.hpp)
    hc->inbox_bytes = (size_t)(2 + kXSlots) * d.n_procs * d.xg * sizeof(unsigned long long);
    void *p = nullptr;
    // fine-grained: peers write it over xGMI while this rank's kernel polls it (system-scope accesses, no L2 copy)
    hipError_t e = hipExtMallocWithFlags(&p, hc->inbox_bytes, hipDeviceMallocFinegrained);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        e = hipExtMallocWithFlags(&p, hc->inbox_bytes, hipDeviceMallocUncached);
    }
    if (e != hipSuccess) { (void)hipGetLastError(); HIPCHK(hipMalloc(&p, hc->inbox_bytes)); }
    HIPCHK(hipMemset(p, 0, hc->inbox_bytes));
    hc->d_inbox = static_cast<unsigned long long *>(p);
    d.inbox = hc->d_inbox;
    return HTM_OK;
}

int htm_chains_xchg_handle(htm_chains *hc, void *handle, size_t handle_bytes)
{
    if (!hc || !handle) return fail(HTM_EINVAL, "NULL argument");
    if (handle_bytes < sizeof(hipIpcMemHandle_t)) return fail(HTM_EINVAL, "handle buffer too small (%zu < %zu)", handle_bytes, sizeof(hipIpcMemHandle_t));
    HIPCHK(hipSetDevice(hc->fwd->device));
    int rc = xchg_alloc(hc);
    if (rc) return rc;
    hipIpcMemHandle_t h;
    HIPCHK(hipIpcGetMemHandle(&h, hc->d_inbox));
    std::memset(handle, 0, handle_bytes);
    std::memcpy(handle, &h, sizeof(h));
    return HTM_OK;
}

int htm_chains_xchg_connect(htm_chains *hc, const void *handles, size_t handle_bytes)
{
    if (!hc) return fail(HTM_EINVAL, "NULL handle");
    const ChainsDev &d0 = hc->dev;
    if (d0.n_procs > 1 && (!handles || handle_bytes < sizeof(hipIpcMemHandle_t)))
        return fail(HTM_EINVAL, "handles of all %d ranks are needed", d0.n_procs);
    if (!hc->persist) return fail(HTM_ESTATE, "the in-kernel exchange needs the persistent kernel (HTM_PERSIST=0 is set)");
    if ((size_t)d0.n_procs * (4 + 2 * d0.n_chains) > (size_t)kGathStage)
        return fail(HTM_EINVAL, "%d ranks x %d chains: the gathered records do not fit the kernel's staging area", d0.n_procs, d0.n_chains);
    HIPCHK(hipSetDevice(hc->fwd->device));
    int rc = xchg_alloc(hc);
    if (rc) return rc;
    if (hc->xchg_ready) return HTM_OK;
    std::vector<unsigned long long *> ptr(d0.n_procs, nullptr);
    for (int q = 0; q < d0.n_procs; ++q) {
        if (q == d0.rank) { ptr[q] = hc->d_inbox; continue; }
        hipIpcMemHandle_t h;
        std::memcpy(&h, static_cast<const char *>(handles) + (size_t)q * handle_bytes, sizeof(h));
        void *p = nullptr;
        const hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            for (void *m : hc->peer_maps) (void)hipIpcCloseMemHandle(m);
            hc->peer_maps.clear();
            return fail(HTM_EHIP, "hipIpcOpenMemHandle of rank %d's inbox failed: %s", q, hipGetErrorString(e));
        }
        hc->peer_maps.push_back(p);
        ptr[q] = static_cast<unsigned long long *>(p);
    }
    if ((rc = dev_upload(hc->pool, &hc->d_outbox, ptr.data(), ptr.size()))) return rc;
    hc->dev.outbox = hc->d_outbox;
    hc->xchg_ready = true;
    return HTM_OK;
}

int htm_chains_xchg_probe(htm_chains *hc, unsigned token, double seconds)
{
    if (!hc) return fail(HTM_EINVAL, "NULL handle");
    if (!hc->xchg_ready) return fail(HTM_ESTATE, "htm_chains_xchg_connect has not been called");
    HIPCHK(hipSetDevice(hc->fwd->device));
    int *d_res = nullptr;
    int rc = dev_alloc(hc->pool, &d_res, 1);
    if (rc) return rc;
    HIPCHK(hipMemset(d_res, 0, sizeof(int)));
    const unsigned long long ticks = (unsigned long long)(std::max(0.01, std::min(seconds, 60.0)) * 1e8);
    // (the probe is collective: every rank has made the same number of calls, so the count makes a repeated probe's tokens
    // new ones whatever token the callers pass -- a second probe must not pass on what the first one left in the inboxes)
    const unsigned eff = (token + 0x10000u * ++hc->probe_calls) & 0x7fffffffu;
    hipLaunchKernelGGL(k_xchg_probe, dim3(1), dim3(64), 0, hc->fwd->stream, hc->dev, eff, ticks, d_res);
    HIPCHK(hipGetLastError());
    int res = 0;
    HIPCHK(hipMemcpyAsync(&res, d_res, sizeof(int), hipMemcpyDeviceToHost, hc->fwd->stream));
    if ((rc = bounded_stream_sync(hc, "inbox probe"))) return rc;
    if (!res) return fail(HTM_ESTATE, "inbox probe: the tokens of the other ranks did not arrive within %.1f s", seconds);
    return HTM_OK;
}

// n_iter lock-step iterations with the exchange inside the kernel: one k_mcmc launch runs until the target, or until
// some rank asks everybody to stop (its record buffers or its produced random stream are nearly used up); then every
// rank drains / refills and launches again.  All ranks leave a launch after the same iteration.
int htm_chains_run_lockstep_direct(htm_chains *hc, int n_iter)
{
    if (!hc || n_iter < 0) return fail(HTM_EINVAL, "bad argument");
    if (!hc->xchg_ready) return fail(HTM_ESTATE, "htm_chains_xchg_connect has not been called");
    htm_forward *h = hc->fwd;
    HIPCHK(hipSetDevice(h->device));
    int rc = read_ctrl(hc);
    if (rc) return rc;
    if ((rc = ctrl_error(hc))) return rc;
    if (hc->h_ctrl.stage != ST_IDLE) return fail(HTM_ESTATE, "an iteration is in flight");
    hc->h_target = hc->h_ctrl.iter_done + n_iter;
    hc->run_full0 = hc->h_ctrl.n_full_evals; hc->run_part0 = hc->h_ctrl.n_partial_evals;
    hc->last_graph_launches = 0;
    HIPCHK(hipEventRecord(hc->ev0, h->stream));
    while (hc->h_ctrl.iter_done < hc->h_target) {
        // the launch reads the produced stream's extent once, at its start: have a good stretch ready
        long long ahead = hc->n_hop - hc->h_ctrl.spos;
        if (ahead < hc->cap / 2 && (rc = stream_produce(hc, std::min<long long>(hc->cap / 2 - ahead + 4096, 1 << 18)))) return rc;
        HIPCHK(hipStreamWaitEvent(h->stream, hc->ev_side, 0));      // (the producer takes ~0.1 ms per 2^18 positions)
        const int before = hc->h_ctrl.iter_done;
        if ((rc = launch_mcmc(hc, MODE_LOCKRUN, hc->h_target, nullptr))) return rc;
        hc->last_graph_launches += 1;
        ahead = hc->n_hop - hc->h_ctrl.spos;        // top up while the launch runs (for the next one)
        if (ahead < hc->cap / 2 && (rc = stream_produce(hc, std::min<long long>(hc->cap / 2 - ahead + 4096, 1 << 18)))) return rc;
        if ((rc = read_ctrl(hc))) return rc;
        if ((rc = ctrl_error(hc))) return rc;
        if (hc->h_ctrl.stop) { if ((rc = drain_records(hc))) return rc; }
        else if (hc->h_ctrl.iter_done == before && hc->h_ctrl.iter_done < hc->h_target)
            return fail(HTM_ESTATE, "lock-step launch made no progress at iteration %d", before);
    }
    HIPCHK(hipEventRecord(hc->ev1, h->stream));
    HIPCHK(hipEventSynchronize(hc->ev1));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, hc->ev0, hc->ev1));
    hc->last_device_us = 1000.0 * ms;
    hc->last_full = hc->h_ctrl.n_full_evals - hc->run_full0;
    hc->last_part = hc->h_ctrl.n_partial_evals - hc->run_part0;
    return drain_records(hc);
}

// ---- RCCL reached from the C ABI (so that a Fortran / C host needs no Python to use RCCL for the swap) ----------
// librccl is bound at run time (dlopen): the library itself does not link against it, and inside a PyTorch process the
// copy torch has already loaded is reused.  Only the four entry points the swap needs.
struct htm_comm {
    void *lib = nullptr, *comm = nullptr;
    int rank = 0, n_ranks = 1, device = 0;
    int (*all_gather)(const void *, void *, size_t, int, void *, void *) = nullptr;
    int (*comm_destroy)(void *) = nullptr;
    double *d_gathered = nullptr;
    size_t gathered_cap = 0;
};
namespace {
struct NcclId { char b[HTM_COMM_ID_BYTES]; };
void *rccl_open()
{
    static void *lib = [] {
        const char *names[] = {getenv("HTM_RCCL_LIB"), "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
        void *h = nullptr;
        for (const char *n : names) {
            if (!n || !*n) continue;
            if ((h = dlopen(n, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD))) break;      // already in the process (PyTorch's copy)
        }
        for (const char *n : names) {
            if (h) break;
            if (!n || !*n) continue;
            h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        }
        return h;
    }();
    return lib;
}
}  // namespace

int htm_comm_unique_id(void *id, size_t id_bytes)
{
    if (!id || id_bytes < HTM_COMM_ID_BYTES) return fail(HTM_EINVAL, "id buffer must hold %d bytes", HTM_COMM_ID_BYTES);
    void *lib = rccl_open();
    if (!lib) return fail(HTM_ENODEVICE, "librccl.so not found (%s)", dlerror());
    auto get_id = reinterpret_cast<int (*)(NcclId *)>(dlsym(lib, "ncclGetUniqueId"));
    if (!get_id) return fail(HTM_ENODEVICE, "ncclGetUniqueId not found in librccl");
    NcclId u;
    std::memset(&u, 0, sizeof(u));
    const int st = get_id(&u);
    if (st != 0) return fail(HTM_EHIP, "ncclGetUniqueId failed with status %d", st);
    std::memset(id, 0, id_bytes);
    std::memcpy(id, &u, sizeof(u));
    return HTM_OK;
}

int htm_comm_create(const void *id, size_t id_bytes, int rank, int n_ranks, int device, htm_comm **out)
{
    if (!out) return fail(HTM_EINVAL, "out is NULL");
    *out = nullptr;
    if (!id || id_bytes < HTM_COMM_ID_BYTES || n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(HTM_EINVAL, "bad argument");
    int rc = use_device(device);
    if (rc) return rc;
    void *lib = rccl_open();
    if (!lib) return fail(HTM_ENODEVICE, "librccl.so not found (%s)", dlerror());
    auto init_rank = reinterpret_cast<int (*)(void **, int, NcclId, int)>(dlsym(lib, "ncclCommInitRank"));
    auto all_gather = reinterpret_cast<int (*)(const void *, void *, size_t, int, void *, void *)>(dlsym(lib, "ncclAllGather"));
    auto destroy = reinterpret_cast<int (*)(void *)>(dlsym(lib, "ncclCommDestroy"));
    if (!init_rank || !all_gather || !destroy) return fail(HTM_ENODEVICE, "librccl lacks ncclCommInitRank / ncclAllGather / ncclCommDestroy");
    NcclId u;
    std::memcpy(&u, id, sizeof(u));
    void *comm = nullptr;
    const int st = init_rank(&comm, n_ranks, u, rank);
    if (st != 0 || !comm) return fail(HTM_EHIP, "ncclCommInitRank failed with status %d (RCCL refuses two ranks on one device)", st);
    htm_comm *c = new htm_comm();
    c->lib = lib; c->comm = comm; c->rank = rank; c->n_ranks = n_ranks; c->device = device;
    c->all_gather = all_gather; c->comm_destroy = destroy;
    *out = c;
    return HTM_OK;
}

int htm_comm_destroy(htm_comm *c)
{
    if (!c) return HTM_OK;
    (void)hipSetDevice(c->device);
    if (c->d_gathered) (void)hipFree(c->d_gathered);
    if (c->comm && c->comm_destroy) (void)c->comm_destroy(c->comm);
    delete c;
    return HTM_OK;
}

int htm_comm_allgather(htm_comm *c, const void *d_send, void *d_recv, size_t bytes_per_rank, void *hip_stream)
{
    if (!c || !d_send || !d_recv) return fail(HTM_EINVAL, "NULL argument");
    const int st = c->all_gather(d_send, d_recv, bytes_per_rank, 0 /* ncclInt8 */, c->comm, hip_stream);
    if (st != 0) return fail(HTM_EHIP, "ncclAllGather failed with status %d", st);
    return HTM_OK;
}

int htm_chains_run_lockstep_comm(htm_chains *hc, int n_iter, htm_comm *c)
{
    if (!hc || !c) return fail(HTM_EINVAL, "NULL argument");
    if (c->n_ranks != hc->dev.n_procs || c->rank != hc->dev.rank) return fail(HTM_EINVAL, "communicator rank/size do not match the chain set");
    HIPCHK(hipSetDevice(hc->fwd->device));
    const size_t words = (4 + 2 * (size_t)hc->dev.n_chains) * (size_t)c->n_ranks;
    if (words > c->gathered_cap) {
        if (c->d_gathered) HIPCHK(hipFree(c->d_gathered));
        c->d_gathered = nullptr;
        HIPCHK(hipMalloc(reinterpret_cast<void **>(&c->d_gathered), words * sizeof(double)));
        c->gathered_cap = words;
    }
    return htm_chains_run_lockstep(hc, n_iter, c->all_gather, c->comm, c->d_gathered);
}

int htm_chains_swap_record(htm_chains *hc, void **d_record, size_t *record_bytes)
{
    if (!hc || !d_record || !record_bytes) return fail(HTM_EINVAL, "NULL argument");
    *d_record = hc->dev.swap_rec;
    *record_bytes = (4 + 2 * (size_t)hc->dev.n_chains) * sizeof(double);
    return HTM_OK;
}

int htm_chains_step_end(htm_chains *hc, const void *d_gathered_records)
{
    if (!hc) return fail(HTM_EINVAL, "NULL handle");
    if (!d_gathered_records) return fail(HTM_EINVAL, "gathered records pointer is NULL");
    HIPCHK(hipSetDevice(hc->fwd->device));
    // deferred: the next step_begin's kernel applies the swap before it advances (one launch less per
    // iteration); anything that looks at the state flushes it first (flush_pending)
    hc->pending_gathered = static_cast<const double *>(d_gathered_records);
    return HTM_OK;
}

int htm_chains_swap_record_host(htm_chains *hc, double *record)
{
    if (!hc || !record) return fail(HTM_EINVAL, "NULL argument");
    HIPCHK(hipSetDevice(hc->fwd->device));
    const size_t words = 4 + 2 * (size_t)hc->dev.n_chains;
    HIPCHK(hipMemcpyAsync(record, hc->dev.swap_rec, words * sizeof(double), hipMemcpyDeviceToHost, hc->fwd->stream));
    return bounded_stream_sync(hc, "waiting for the iteration's swap record");
}

int htm_chains_step_end_host(htm_chains *hc, const double *gathered_records)
{
    if (!hc || !gathered_records) return fail(HTM_EINVAL, "NULL argument");
    HIPCHK(hipSetDevice(hc->fwd->device));
    const size_t words = (4 + 2 * (size_t)hc->dev.n_chains) * (size_t)hc->dev.n_procs;
    if (!hc->d_gath_host) {
        int rc = dev_alloc(hc->pool, &hc->d_gath_host, words);
        if (rc) return rc;
        HIPCHK(hipHostMalloc(reinterpret_cast<void **>(&hc->h_gath_pinned), words * sizeof(double), hipHostMallocDefault));
    }
    // the caller may reuse its array at once: stage through pinned memory (the previous copy has completed --
    // swap_record_host of this iteration waited for the stream)
    std::memcpy(hc->h_gath_pinned, gathered_records, words * sizeof(double));
    HIPCHK(hipMemcpyAsync(hc->d_gath_host, hc->h_gath_pinned, words * sizeof(double), hipMemcpyHostToDevice, hc->fwd->stream));
    return htm_chains_step_end(hc, hc->d_gath_host);
}

int htm_chains_sync(htm_chains *hc)
{
    if (!hc) return fail(HTM_EINVAL, "NULL handle");
    HIPCHK(hipSetDevice(hc->fwd->device));
    int rc = read_ctrl(hc);
    if (rc) return rc;
    return ctrl_error(hc);
}

int htm_chains_drain(htm_chains *hc)
{
    int rc = htm_chains_sync(hc);
    if (rc) return rc;
    return drain_records(hc);
}

int htm_chains_iterations_done(htm_chains *hc, int *n)
{
    if (!hc || !n) return fail(HTM_EINVAL, "NULL argument");
    int rc = htm_chains_sync(hc);
    if (rc) return rc;
    *n = hc->h_ctrl.iter_done;
    return HTM_OK;
}

// ---- checkpoint / resume ------------------------------------------------------------------------------
namespace {
struct CkptHeader {
    uint32_t magic, version;
    int32_t n_chains, n_sta, n_events, n_procs, rank, iter_done;
    uint32_t rng_state[4];
    int64_t n_full_evals, n_partial_evals;
    uint64_t jobs_total;
    uint64_t total;           // doubles in the parameter vector
};
constexpr uint32_t kCkptMagic = 0x48544d43u;   // "HTMC"
size_t ckpt_total(const htm_chains *hc)
{
    const size_t n = hc->dev.n_chains, S = hc->dev.S, E = hc->dev.E;
    return 2 * n + 2 * n * S + 3 * E * n;
}
size_t ckpt_bytes(const htm_chains *hc)
{
    const size_t n = hc->dev.n_chains;
    return sizeof(CkptHeader) + (ckpt_total(hc) + 2 * n) * sizeof(double) + (2 * 7 + 1) * n * sizeof(int32_t);      // (+ prev_mid)
}
}  // namespace

int htm_chains_checkpoint_size(htm_chains *hc, size_t *bytes)
{
    if (!hc || !bytes) return fail(HTM_EINVAL, "NULL argument");
    *bytes = ckpt_bytes(hc);
    return HTM_OK;
}

int htm_chains_checkpoint_save(htm_chains *hc, void *blob, size_t bytes)
{
    if (!hc || !blob) return fail(HTM_EINVAL, "NULL argument");
    if (bytes < ckpt_bytes(hc)) return fail(HTM_EINVAL, "checkpoint buffer too small (%zu < %zu)", bytes, ckpt_bytes(hc));
    int rc = htm_chains_sync(hc);            // flushes a pending swap, raises device error flags
    if (rc) return rc;
    if (hc->h_ctrl.stage != ST_IDLE) return fail(HTM_ESTATE, "a lock-step iteration is in flight");
    CkptHeader h{};
    h.magic = kCkptMagic; h.version = 2;
    h.n_chains = hc->dev.n_chains; h.n_sta = hc->dev.S; h.n_events = hc->dev.E; h.n_procs = hc->dev.n_procs; h.rank = hc->dev.rank;
    h.iter_done = hc->h_ctrl.iter_done;
    if ((rc = htm_chains_get_rng(hc, h.rng_state))) return rc;
    h.n_full_evals = hc->h_ctrl.n_full_evals; h.n_partial_evals = hc->h_ctrl.n_partial_evals;
    h.jobs_total = hc->h_ctrl.jobs_total;
    h.total = ckpt_total(hc);
    char *p = static_cast<char *>(blob);
    std::memcpy(p, &h, sizeof(h)); p += sizeof(h);
    const size_t n = hc->dev.n_chains;
    HIPCHK(hipMemcpy(p, hc->dev.xall, h.total * sizeof(double), hipMemcpyDeviceToHost)); p += h.total * sizeof(double);
    HIPCHK(hipMemcpy(p, hc->dev.temp, n * sizeof(double), hipMemcpyDeviceToHost)); p += n * sizeof(double);
    HIPCHK(hipMemcpy(p, hc->dev.L, n * sizeof(double), hipMemcpyDeviceToHost)); p += n * sizeof(double);
    HIPCHK(hipMemcpy(p, hc->dev.n_propose, 7 * n * sizeof(int32_t), hipMemcpyDeviceToHost)); p += 7 * n * sizeof(int32_t);
    HIPCHK(hipMemcpy(p, hc->dev.n_accept, 7 * n * sizeof(int32_t), hipMemcpyDeviceToHost)); p += 7 * n * sizeof(int32_t);
    // (what each chain's last step was: decides which event its next full evaluation leaves to the chain's own wave, so that
    // a continued run sums in the order of the uninterrupted one)
    HIPCHK(hipMemcpy(p, hc->dev.prev_mid, n * sizeof(int32_t), hipMemcpyDeviceToHost));
    return HTM_OK;
}

int htm_chains_checkpoint_load(htm_chains *hc, const void *blob, size_t bytes)
{
    if (!hc || !blob) return fail(HTM_EINVAL, "NULL argument");
    if (bytes < sizeof(CkptHeader)) return fail(HTM_EINVAL, "checkpoint blob truncated");
    CkptHeader h{};
    std::memcpy(&h, blob, sizeof(h));
    if (h.magic != kCkptMagic || h.version != 2) return fail(HTM_EINVAL, "not a checkpoint of this library (magic/version)");
    if (h.n_chains != hc->dev.n_chains || h.n_sta != hc->dev.S || h.n_events != hc->dev.E || h.n_procs != hc->dev.n_procs ||
        h.rank != hc->dev.rank || h.total != ckpt_total(hc))
        return fail(HTM_EINVAL, "checkpoint shape (%d chains, %d x %d, rank %d/%d) does not match this chain set", h.n_chains,
                    h.n_events, h.n_sta, h.rank, h.n_procs);
    if (bytes < ckpt_bytes(hc)) return fail(HTM_EINVAL, "checkpoint blob truncated");
    // A load REPLACES the control block -- iteration counter, stage, error word -- so it must work on a chain set whose last
    // run failed (the roll-back of a failed lock-step run, parallel.py): wait for the stream, do not report what is about to
    // be overwritten.  (An iteration of the per-launch lock-step in flight -- records out, swap not applied -- is dropped too.)
    HIPCHK(hipSetDevice(hc->fwd->device));
    hc->pending_gathered = nullptr;
    int rc = bounded_stream_sync(hc, "waiting for the chain kernels before a checkpoint load");
    if (rc) return rc;
    hc->ctrl_fresh = false;
    HIPCHK(hipStreamSynchronize(hc->side));
    const char *p = static_cast<const char *>(blob) + sizeof(h);
    const size_t n = hc->dev.n_chains;
    HIPCHK(hipMemcpy(hc->dev.xall, p, h.total * sizeof(double), hipMemcpyHostToDevice)); p += h.total * sizeof(double);
    HIPCHK(hipMemcpy(hc->dev.temp, p, n * sizeof(double), hipMemcpyHostToDevice)); p += n * sizeof(double);
    HIPCHK(hipMemcpy(hc->dev.L, p, n * sizeof(double), hipMemcpyHostToDevice)); p += n * sizeof(double);
    HIPCHK(hipMemcpy(hc->dev.n_propose, p, 7 * n * sizeof(int32_t), hipMemcpyHostToDevice)); p += 7 * n * sizeof(int32_t);
    HIPCHK(hipMemcpy(hc->dev.n_accept, p, 7 * n * sizeof(int32_t), hipMemcpyHostToDevice)); p += 7 * n * sizeof(int32_t);
    HIPCHK(hipMemcpy(hc->dev.prev_mid, p, n * sizeof(int32_t), hipMemcpyHostToDevice));
    // the random stream restarts at the saved generator state: position 0 of a fresh stream
    for (int k = 0; k < 4; ++k) hc->init_state[k] = h.rng_state[k];
    HIPCHK(hipMemcpy(hc->dev.stream.gen, hc->init_state, 4 * sizeof(uint32_t), hipMemcpyHostToDevice));
    hc->gen_par = 0;
    const long long zero = 0;
    HIPCHK(hipMemcpy(hc->dev.stream.hop_end, &zero, sizeof(zero), hipMemcpyHostToDevice));
    hc->n_raw = hc->n_tr = hc->n_rec = hc->n_hop = 0;
    hc->spos_lo = hc->spos_hi = 0;
    Ctrl c{};
    c.stage = ST_IDLE; c.spos = 0; c.iter_done = h.iter_done; c.iter_target = h.iter_done;
    c.n_full_evals = h.n_full_evals; c.n_partial_evals = h.n_partial_evals; c.jobs_total = h.jobs_total;
    c.slog_n = 0; c.slog_cap = hc->h_ctrl.slog_cap;
    hc->h_ctrl = c;
    hc->h_target = h.iter_done;
    hc->pending_gathered = nullptr;
    HIPCHK(hipMemcpy(hc->dev.ctrl, &c, sizeof(Ctrl), hipMemcpyHostToDevice));
    // hand-off buffers of the persistent kernel: tags are functions of the job counter and of iteration | chain, which
    // a roll-back repeats -- nothing an earlier run left there may match
    {
        const ChainsDev &d = hc->dev;
        HIPCHK(hipMemset(d.slots, 0, (size_t)d.slot_rep * d.slot_stride * sizeof(unsigned long long)));
        HIPCHK(hipMemset(d.pgran, 0, (size_t)d.n_chains * d.n_workers * d.pgran_stride * sizeof(unsigned long long)));
        HIPCHK(hipMemset(d.lo_gran, 0, (size_t)d.n_chains * 16 * sizeof(unsigned long long)));
        HIPCHK(hipMemset(d.diag, 0, 24 * sizeof(unsigned long long)));      // (what a failed wait left for the error message; [24], [25] are counters)
        // The swap-record inbox too: its tags are bare iteration numbers, and the iterations after the saved one are about
        // to be run again (records and stop / error words of the first time must not be taken for the second).  For a
        // multi-rank job a load is therefore COLLECTIVE: every rank loads, then the ranks meet at a barrier before the next
        // htm_chains_run_lockstep_direct (no peer posts outside a run, so nothing lands in a cleared inbox before that).
        if (hc->d_inbox && hc->inbox_bytes) HIPCHK(hipMemset(hc->d_inbox, 0, hc->inbox_bytes));
    }
    hc->lik_iter.clear(); hc->lik_chain.clear(); hc->lik_val.clear();
    hc->smp_iter.clear(); hc->smp_chain.clear(); hc->smp_data.clear();
    if ((rc = stream_produce(hc, 1 << 16))) return rc;
    HIPCHK(hipStreamSynchronize(hc->side));
    return HTM_OK;
}

int htm_chains_get_state(htm_chains *hc, int chain, double *hypo, double *t_corr, double *vs, double *a_corr,
                         double *qs, double *temp, double *log_likelihood, int32_t n_propose[7], int32_t n_accept[7])
{
    if (!hc) return fail(HTM_EINVAL, "NULL handle");
    const ChainsDev &d = hc->dev;
    if (chain < 0 || chain >= d.n_chains) return fail(HTM_EINVAL, "chain %d out of range", chain);
    int rc = htm_chains_sync(hc);
    if (rc) return rc;
    const size_t c = chain;
    if (hypo) HIPCHK(hipMemcpy(hypo, d.hypo.x + c * d.hypo.nx, d.hypo.nx * sizeof(double), hipMemcpyDeviceToHost));
    if (t_corr) HIPCHK(hipMemcpy(t_corr, d.tc.x + c * d.S, d.S * sizeof(double), hipMemcpyDeviceToHost));
    if (a_corr) HIPCHK(hipMemcpy(a_corr, d.ac.x + c * d.S, d.S * sizeof(double), hipMemcpyDeviceToHost));
    if (vs) HIPCHK(hipMemcpy(vs, d.vs.x + c, sizeof(double), hipMemcpyDeviceToHost));
    if (qs) HIPCHK(hipMemcpy(qs, d.qs.x + c, sizeof(double), hipMemcpyDeviceToHost));
    if (temp) HIPCHK(hipMemcpy(temp, d.temp + c, sizeof(double), hipMemcpyDeviceToHost));
    if (log_likelihood) HIPCHK(hipMemcpy(log_likelihood, d.L + c, sizeof(double), hipMemcpyDeviceToHost));
    if (n_propose) HIPCHK(hipMemcpy(n_propose, d.n_propose + 7 * c, 7 * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (n_accept) HIPCHK(hipMemcpy(n_accept, d.n_accept + 7 * c, 7 * sizeof(int32_t), hipMemcpyDeviceToHost));
    return HTM_OK;
}

int htm_chains_share_gpu(htm_chains *hc, int ranks_on_this_gpu)
{
    if (!hc || ranks_on_this_gpu < 1) return fail(HTM_EINVAL, "bad argument");
    if (getenv("HTM_RANKS_PER_GPU") || !hc->persist || ranks_on_this_gpu == 1) return HTM_OK;      // (an explicit setting stands)
    // (a chain set that has already run keeps the launch shape it ran with: ADVICE r3 -- a world built around it must not fail)
    if (hc->launch_seq > 0) return HTM_OK;
    // Blocks go to the 8 XCDs of the GPU in turn, every launch starting with the first: a rank's blocks must be spread evenly over
    // them (a multiple of 8) and the ranks' shares of one XCD's CUs must add up to no more than it has -- 5 ranks x 51 blocks are
    // 255 of 256 CUs and still do not fit (7 blocks x 5 ranks on XCDs 0..2: the launches wait for each other's CUs forever).
    const long per_xcd = hc->blocks_fit / 8;
    const long room = per_xcd >= ranks_on_this_gpu ? 8 * (per_xcd / ranks_on_this_gpu) - 1 : hc->blocks_fit / ranks_on_this_gpu - 1;
    if (room < 1) return fail(HTM_ESTATE, "%d ranks on one GPU: not even one worker block per rank fits next to the masters", ranks_on_this_gpu);
    if (hc->dev.n_workers > room) hc->dev.n_workers = worker_blocks(hc->fwd->E, 8, room);
    hc->dev.n_wg = hc->dev.n_workers;
    return HTM_OK;
}

int htm_chains_get_loglik(htm_chains *hc, int chain, double *log_likelihood)
{
    if (!hc || !log_likelihood) return fail(HTM_EINVAL, "NULL argument");
    if (chain < 0 || chain >= hc->dev.n_chains) return fail(HTM_EINVAL, "chain %d out of range", chain);
    int rc = htm_chains_sync(hc);
    if (rc) return rc;
    HIPCHK(hipMemcpy(log_likelihood, hc->dev.L + chain, sizeof(double), hipMemcpyDeviceToHost));
    return HTM_OK;
}

int htm_chains_get_rng(htm_chains *hc, uint32_t state[4])
{
    if (!hc || !state) return fail(HTM_EINVAL, "NULL argument");
    int rc = htm_chains_sync(hc);
    if (rc) return rc;
    // mod_random state after spos draws = words spos..spos+3 of [x0, y0, z0, w0, out_0, out_1, ...]
    const long long sp = hc->h_ctrl.spos;
    for (int k = 0; k < 4; ++k) {
        const long long word = sp + k;
        if (word < 4) state[k] = hc->init_state[word];
        else HIPCHK(hipMemcpy(&state[k], hc->dev.stream.raw + ((word - 4) & hc->dev.stream.mask), sizeof(uint32_t),
                              hipMemcpyDeviceToHost));
    }
    return HTM_OK;
}

int htm_chains_lik_count(htm_chains *hc, int *n)
{
    if (!hc || !n) return fail(HTM_EINVAL, "NULL argument");
    *n = (int)hc->lik_iter.size();
    return HTM_OK;
}

int htm_chains_lik_read(htm_chains *hc, int32_t *iter, int32_t *chain, double *log_likelihood)
{
    if (!hc) return fail(HTM_EINVAL, "NULL handle");
    const size_t n = hc->lik_iter.size();
    if (iter) memcpy(iter, hc->lik_iter.data(), n * sizeof(int32_t));
    if (chain) memcpy(chain, hc->lik_chain.data(), n * sizeof(int32_t));
    if (log_likelihood) memcpy(log_likelihood, hc->lik_val.data(), n * sizeof(double));
    return HTM_OK;
}

int htm_chains_sample_count(htm_chains *hc, int *n)
{
    if (!hc || !n) return fail(HTM_EINVAL, "NULL argument");
    *n = (int)hc->smp_iter.size();
    return HTM_OK;
}

int htm_chains_sample_read(htm_chains *hc, int k, int32_t *iter, int32_t *chain, double *vs, double *qs,
                           double *hypo, double *t_corr, double *a_corr)
{
    if (!hc) return fail(HTM_EINVAL, "NULL handle");
    if (k < 0 || k >= (int)hc->smp_iter.size()) return fail(HTM_EINVAL, "sample %d out of range", k);
    const int nh = hc->dev.hypo.nx, S = hc->dev.S;
    const double *r = hc->smp_data.data() + (size_t)k * hc->rec_len;
    if (iter) *iter = hc->smp_iter[k];
    if (chain) *chain = hc->smp_chain[k];
    if (hypo) memcpy(hypo, r, nh * sizeof(double));
    if (t_corr) memcpy(t_corr, r + nh, S * sizeof(double));
    if (a_corr) memcpy(a_corr, r + nh + S, S * sizeof(double));
    if (vs) *vs = r[nh + 2 * S];
    if (qs) *qs = r[nh + 2 * S + 1];
    return HTM_OK;
}

int htm_chains_clear_records(htm_chains *hc)
{
    if (!hc) return fail(HTM_EINVAL, "NULL handle");
    hc->lik_iter.clear(); hc->lik_chain.clear(); hc->lik_val.clear();
    hc->smp_iter.clear(); hc->smp_chain.clear(); hc->smp_data.clear();
    return HTM_OK;
}

int htm_chains_enable_steplog(htm_chains *hc, int capacity)
{
    if (!hc || capacity < 0) return fail(HTM_EINVAL, "bad argument");
    int rc = htm_chains_sync(hc);
    if (rc) return rc;
    if (capacity > 0) {
        if ((rc = dev_alloc(hc->pool, &hc->dev.slog_i, 8 * (size_t)capacity))) return rc;
        if ((rc = dev_alloc(hc->pool, &hc->dev.slog_d, 4 * (size_t)capacity))) return rc;
    }
    // the graph bakes ChainsDev by value: rebuild it with the new pointers
    if (hc->gexec) { (void)hipGraphExecDestroy(hc->gexec); hc->gexec = nullptr; }
    if (hc->graph) { (void)hipGraphDestroy(hc->graph); hc->graph = nullptr; }
    hc->dev_np.slog_i = hc->dev.slog_i; hc->dev_np.slog_d = hc->dev.slog_d;
    hc->h_ctrl.slog_n = 0; hc->h_ctrl.slog_cap = capacity;
    HIPCHK(hipMemcpy(&hc->dev.ctrl->slog_n, &hc->h_ctrl.slog_n, 2 * sizeof(int), hipMemcpyHostToDevice));
    return HTM_OK;
}

int htm_chains_steplog_read(htm_chains *hc, int *n, int32_t *irows, double *drows)
{
    if (!hc || !n) return fail(HTM_EINVAL, "NULL argument");
    int rc = htm_chains_sync(hc);
    if (rc) return rc;
    const int rows = std::min(hc->h_ctrl.slog_n, hc->h_ctrl.slog_cap);
    *n = rows;
    if (rows > 0 && irows) HIPCHK(hipMemcpy(irows, hc->dev.slog_i, 8 * (size_t)rows * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (rows > 0 && drows) HIPCHK(hipMemcpy(drows, hc->dev.slog_d, 4 * (size_t)rows * sizeof(double), hipMemcpyDeviceToHost));
    return HTM_OK;
}

int htm_chains_last_run_stats(htm_chains *hc, double *device_us, int *graph_launches, int64_t *full_evals,
                              int64_t *partial_evals)
{
    if (!hc) return fail(HTM_EINVAL, "NULL handle");
    if (device_us) *device_us = hc->last_device_us;
    if (graph_launches) *graph_launches = hc->last_graph_launches;
    if (full_evals) *full_evals = hc->last_full;
    if (partial_evals) *partial_evals = hc->last_part;
    return HTM_OK;
}

int htm_chains_master_stats(htm_chains *hc, int *single_rank_loop, int *lockstep_loop, int64_t *flushes)
{
    if (!hc) return fail(HTM_EINVAL, "NULL argument");
    if (single_rank_loop) *single_rank_loop = !hc->persist ? -1 : hc->pipe ? 5 : (hc->flow && hc->mb_blocks > 1) ? 7 : hc->flow ? 3 : 0;
    if (lockstep_loop) *lockstep_loop = !hc->persist ? -1 : hc->pipe_lock ? 6 : hc->flow_lock ? 4 : 2;
    if (flushes) {
        HIPCHK(hipStreamSynchronize(hc->fwd->stream));
        unsigned long long v = 0;
        HIPCHK(hipMemcpy(&v, hc->dev.diag + 25, sizeof(v), hipMemcpyDeviceToHost));
        *flushes = (int64_t)v;
    }
    return HTM_OK;
}

int htm_chains_handoff_stats(htm_chains *hc, int64_t *orders_put_aside)
{
    if (!hc || !orders_put_aside) return fail(HTM_EINVAL, "NULL argument");
    HIPCHK(hipStreamSynchronize(hc->fwd->stream));
    unsigned long long v = 0;
    HIPCHK(hipMemcpy(&v, hc->dev.diag + 24, sizeof(v), hipMemcpyDeviceToHost));
    *orders_put_aside = (int64_t)v;
    return HTM_OK;
}

#ifdef HTM_STAMPS
/* diagnostic builds only: the pipelined master's event trace ({time, code << 48 | iteration << 8 | chain} pairs) */
int htm_chains_read_trace(htm_chains *hc, unsigned long long *out, int n_pairs)
{
#ifdef HTM_STAMPS
    if (!hc || !out || n_pairs > 8192) return fail(HTM_EINVAL, "bad argument");
    HIPCHK(hipStreamSynchronize(hc->fwd->stream));
    HIPCHK(hipMemcpy(out, hc->dev.stamps + 128, (size_t)n_pairs * 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return HTM_OK;
#else
    (void)hc; (void)out; (void)n_pairs;
    return fail(HTM_ESTATE, "not a diagnostic build");
#endif
}

int htm_chains_read_stamps(htm_chains *hc, unsigned long long out[128])
{
    HIPCHK(hipStreamSynchronize(hc->fwd->stream));
    HIPCHK(hipMemcpy(out, hc->dev.stamps, 128 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return HTM_OK;
}
/* diagnostics of a wedged hand-off: out[0..7] = the order slot of `chain` (replica 0), out[8..8+2*n) = the first n workers'
 * partial-sum granules of that chain (n <= 28) */
int htm_chains_read_handoff(htm_chains *hc, int chain, unsigned long long out[64])
{
    (void)hipStreamSynchronize(hc->fwd->stream);
    const ChainsDev &d = hc->dev;
    if (chain < 0 || chain >= d.n_chains) return fail(HTM_EINVAL, "chain out of range");
    HIPCHK(hipMemcpy(out, d.slots + (size_t)chain * kGranPerSlot, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    const int n = std::min(28, d.n_workers);
    for (int k = 0; k < n; ++k)
        HIPCHK(hipMemcpy(out + 8 + 2 * k, d.pgran + ((size_t)chain * d.n_workers + k) * d.pgran_stride, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return HTM_OK;
}
#endif

int htm_quantiles_dev(int device, const double *d_samples, long n_mod, long n_par, long ld, const int ranks_1based[3],
                      double *d_out, void *hip_stream)
{
    if (!d_samples || !d_out || !ranks_1based) return fail(HTM_EINVAL, "NULL argument");
    if (n_mod < 1 || n_par < 1 || ld < n_par) return fail(HTM_EINVAL, "bad shape (n_mod %ld, n_par %ld, ld %ld)", n_mod, n_par, ld);
    for (int r = 0; r < 3; ++r)
        if (ranks_1based[r] < 1 || ranks_1based[r] > n_mod)
            return fail(HTM_EINVAL, "rank %d outside 1..%ld (the reference would index outside its sorted column)", ranks_1based[r], n_mod);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(HTM_ENODEVICE, "no HIP device");
    HIPCHK(hipSetDevice(device));
    const dim3 grid((unsigned)((n_par + 63) / 64)), block(64 * kSelRG);
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    const char *force = getenv("HTM_SELECT_SLABS");
    // small sets: one launch, a column group per workgroup; large sets: row slabs over the whole chip, a launch per digit
    long slabs = 1;
    if ((double)n_mod * (double)n_par >= (double)(1 << 22)) {
        slabs = std::max(1L, std::min((n_mod + 255) / 256, (long)(2048 / grid.x)));
        slabs = std::min(slabs, 1024L);
    }
    if (force) slabs = std::max(1L, std::min(atol(force), std::min(n_mod, 65535L)));
    if (slabs <= 1 && !force) {
        hipLaunchKernelGGL(k_select, grid, block, 0, st, d_samples, n_mod, n_par, ld,
                           ranks_1based[0] - 1, ranks_1based[1] - 1, ranks_1based[2] - 1, d_out);
        HIPCHK(hipGetLastError());
        return HTM_OK;
    }
    // workspace: three histograms, two prefix/remaining states (stream-ordered allocation keeps the call asynchronous)
    const size_t hist_b = (size_t)grid.x * kSelHistPerGroup * sizeof(int);
    const size_t st_b = (size_t)n_par * kSelRanks * sizeof(unsigned long long);
    const size_t total = 3 * hist_b + 4 * st_b;
    char *ws = nullptr;
    bool async_alloc = hipMallocAsync(reinterpret_cast<void **>(&ws), total, st) == hipSuccess;
    if (!async_alloc) {
        (void)hipGetLastError();
        if (hipMalloc(reinterpret_cast<void **>(&ws), total) != hipSuccess) return fail(HTM_EHIP, "hipMalloc of %zu bytes failed", total);
    }
    HIPCHK(hipMemsetAsync(ws, 0, total, st));
    SelWork w;
    for (int k = 0; k < 3; ++k) w.hist[k] = reinterpret_cast<int *>(ws + k * hist_b);
    for (int k = 0; k < 2; ++k) {
        w.prefix[k] = reinterpret_cast<unsigned long long *>(ws + 3 * hist_b + (2 * k) * st_b);
        w.remaining[k] = reinterpret_cast<long *>(ws + 3 * hist_b + (2 * k + 1) * st_b);
    }
    const long slab_rows = (n_mod + slabs - 1) / slabs;
    const dim3 grid2(grid.x, (unsigned)slabs);
    int pass = 0;
    for (int shift = 60; shift >= 0; shift -= 4, ++pass)
        hipLaunchKernelGGL(k_select_pass, grid2, block, 0, st, d_samples, n_mod, n_par, ld, ranks_1based[0] - 1,
                           ranks_1based[1] - 1, ranks_1based[2] - 1, shift, pass, slab_rows, w, (double *)nullptr);
    hipLaunchKernelGGL(k_select_pass, grid, block, 0, st, d_samples, n_mod, n_par, ld, ranks_1based[0] - 1,
                       ranks_1based[1] - 1, ranks_1based[2] - 1, -4, pass, slab_rows, w, d_out);
    HIPCHK(hipGetLastError());
    if (async_alloc) {
        HIPCHK(hipFreeAsync(ws, st));
    } else {
        HIPCHK(hipStreamSynchronize(st));
        (void)hipFree(ws);
    }
    return HTM_OK;
}

int htm_quantiles(int device, const double *samples, long n_mod, long n_par, const int ranks_1based[3], double *out)
{
    if (!samples || !out) return fail(HTM_EINVAL, "NULL argument");
    if (n_mod < 1 || n_par < 1) return fail(HTM_EINVAL, "bad shape");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(HTM_ENODEVICE, "no HIP device");
    HIPCHK(hipSetDevice(device));
    double *d_x = nullptr, *d_o = nullptr;
    const size_t nb = (size_t)n_mod * n_par * sizeof(double), ob = (size_t)n_par * 3 * sizeof(double);
    if (hipMalloc(reinterpret_cast<void **>(&d_x), nb) != hipSuccess) return fail(HTM_EHIP, "hipMalloc of %zu bytes failed", nb);
    if (hipMalloc(reinterpret_cast<void **>(&d_o), ob) != hipSuccess) { (void)hipFree(d_x); return fail(HTM_EHIP, "hipMalloc failed"); }
    int rc = HTM_OK;
    if (hipMemcpy(d_x, samples, nb, hipMemcpyHostToDevice) != hipSuccess) rc = fail(HTM_EHIP, "upload failed");
    if (rc == HTM_OK) rc = htm_quantiles_dev(device, d_x, n_mod, n_par, n_par, ranks_1based, d_o, nullptr);
    if (rc == HTM_OK && hipMemcpy(out, d_o, ob, hipMemcpyDeviceToHost) != hipSuccess) rc = fail(HTM_EHIP, "download failed");
    (void)hipFree(d_x); (void)hipFree(d_o);
    return rc;
}

int htm_select_regress(int device, int n_sta, int n_win, const double *sta_x, const double *sta_y, const double *sta_z,
                       double z_guess, const double *t, const double *t_err, const double *a, const double *a_err, double *out)
{
    if (!sta_x || !sta_y || !sta_z || !t || !t_err || !a || !a_err || !out) return fail(HTM_EINVAL, "NULL argument");
    if (n_sta < 3 || n_win < 1) return fail(HTM_EINVAL, "need n_sta >= 3 and n_win >= 1 (got %d, %d)", n_sta, n_win);
    int rc = use_device(device);
    if (rc) return rc;
    std::vector<void *> pool;
    auto done = [&](int code) { for (void *p : pool) (void)hipFree(p); return code; };
    const size_t n = (size_t)n_sta * n_win;
    double *dx = nullptr, *dy = nullptr, *dz = nullptr, *dt = nullptr, *dte = nullptr, *da = nullptr, *dae = nullptr, *dout = nullptr;
    if ((rc = dev_upload(pool, &dx, sta_x, n_sta)) || (rc = dev_upload(pool, &dy, sta_y, n_sta)) || (rc = dev_upload(pool, &dz, sta_z, n_sta)) ||
        (rc = dev_upload(pool, &dt, t, n)) || (rc = dev_upload(pool, &dte, t_err, n)) || (rc = dev_upload(pool, &da, a, n)) ||
        (rc = dev_upload(pool, &dae, a_err, n)) || (rc = dev_alloc(pool, &dout, 6 * (size_t)n_win))) return done(rc);
    hipLaunchKernelGGL(k_regress, dim3((n_win + 3) / 4), dim3(256), 0, 0, n_sta, n_win, dx, dy, dz, z_guess, dt, dte, da, dae, dout);
    if (hipGetLastError() != hipSuccess) return done(fail(HTM_EHIP, "k_regress launch failed"));
    if (hipMemcpy(out, dout, 6 * (size_t)n_win * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return done(fail(HTM_EHIP, "download failed"));
    return done(HTM_OK);
}

int htm_rng_jump(const uint32_t state_in[4], unsigned long long n_draws, uint32_t state_out[4])
{
    if (!state_in || !state_out) return fail(HTM_EINVAL, "NULL argument");
    const std::vector<Mat128> &P = xs_powers();
    Bits128 s{{state_in[0], state_in[1], state_in[2], state_in[3]}};
    for (int k = 0; k < 64; ++k)
        if ((n_draws >> k) & 1ull) s = gf2_matvec(P[k], s);
    for (int k = 0; k < 4; ++k) state_out[k] = s.w[k];
    return HTM_OK;
}

// the parallel generator against a serial loop on the device: n draws from `seed`, ring of `cap` positions starting at `start`
static int selftest_rawgen(const uint32_t seed[4], int n, long long start, long long cap)
{
    std::vector<void *> pool;
    auto done = [&](int code) { for (void *p : pool) (void)hipFree(p); return code; };
    StreamDev sd{};
    sd.mask = cap - 1;
    uint32_t *d_ser = nullptr, *d_gen = nullptr;
    u32x4 *d_jump = nullptr;
    int rc;
    if ((rc = dev_alloc(pool, &sd.raw, (size_t)cap)) || (rc = dev_alloc(pool, &d_ser, (size_t)n))) return done(rc);
    uint32_t g16[16] = {seed[0], seed[1], seed[2], seed[3]};
    if ((rc = dev_upload(pool, &d_gen, g16, 16))) return done(rc);
    const std::vector<Mat128> &P = xs_powers();
    std::vector<u32x4> jt((size_t)kJumpLevels * 128);
    for (int b = 0; b < kJumpLevels; ++b)
        for (int j = 0; j < 128; ++j) jt[(size_t)b * 128 + j] = u32x4{P[6 + b][j].w[0], P[6 + b][j].w[1], P[6 + b][j].w[2], P[6 + b][j].w[3]};
    if ((rc = dev_upload(pool, &d_jump, jt.data(), jt.size()))) return done(rc);
    hipLaunchKernelGGL(k_rawgen, dim3((unsigned)((n + 4095) / 4096)), dim3(64), 0, 0, sd, start, n, d_jump, d_gen, d_gen + 4);
    hipLaunchKernelGGL(k_rawgen_serial, dim3(1), dim3(1), 0, 0, d_ser, n, d_gen, d_gen + 8);
    if (hipDeviceSynchronize() != hipSuccess) return done(fail(HTM_EHIP, "rawgen selftest kernels failed"));
    std::vector<uint32_t> ring((size_t)cap), ser((size_t)n);
    uint32_t g[16];
    if (hipMemcpy(ring.data(), sd.raw, cap * 4, hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(ser.data(), d_ser, (size_t)n * 4, hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(g, d_gen, sizeof(g), hipMemcpyDeviceToHost) != hipSuccess) return done(fail(HTM_EHIP, "download failed"));
    for (int k = 0; k < n; ++k)
        if (ring[(size_t)((start + k) & (cap - 1))] != ser[k])
            return done(fail(HTM_ESTATE, "parallel xorshift128 differs from the serial stream at draw %d of %d", k, n));
    uint32_t hj[4];
    htm_rng_jump(seed, (unsigned long long)n, hj);
    for (int k = 0; k < 4; ++k)
        if (g[4 + k] != g[8 + k] || g[4 + k] != hj[k])
            return done(fail(HTM_ESTATE, "generator state after %d draws: parallel %08x serial %08x host jump %08x", n, g[4 + k], g[8 + k], hj[k]));
    return done(HTM_OK);
}

int htm_selftest_math(int device, int which, const double *x, double *y, int n)
{
    int rc = use_device(device);
    if (rc) return rc;
    if (!x || !y || n < 0 || which < 0 || which > 6 || (which == 4 && n % 64 != 0) || (which >= 5 && n % 256 != 0))
        return fail(HTM_EINVAL, "htm_selftest_math: null pointer, negative count or unknown function");
    if (n == 0) return HTM_OK;
    double *d = nullptr;
    HIPCHK(hipMalloc(reinterpret_cast<void **>(&d), 2 * (size_t)n * sizeof(double)));
    auto done = [&](int code) { (void)hipFree(d); return code; };
    if (hipMemcpy(d, x, (size_t)n * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return done(fail(HTM_EHIP, "htm_selftest_math: copy in"));
    hipLaunchKernelGGL(k_mathtest, dim3((n + 255) / 256), dim3(256), 0, 0, which, d, d + n, n);
    if (hipGetLastError() != hipSuccess || hipMemcpy(y, d + n, (size_t)n * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess)
        return done(fail(HTM_EHIP, "htm_selftest_math: kernel or copy out failed"));
    return done(HTM_OK);
}

int htm_selftest(int device)
{
    int rc = use_device(device);
    if (rc) return rc;
    std::vector<double> in(128);
    uint32_t s = 12345u;
    for (auto &v : in) { s = s * 1664525u + 1013904223u; v = (double)(int32_t)s / 65536.0 / 7.0; }
    double *d_in = nullptr, *d_o = nullptr;
    uint32_t *d_r = nullptr;
    HIPCHK(hipMalloc(reinterpret_cast<void **>(&d_in), 128 * sizeof(double)));
    HIPCHK(hipMalloc(reinterpret_cast<void **>(&d_o), 16 * sizeof(double)));
    HIPCHK(hipMalloc(reinterpret_cast<void **>(&d_r), 8 * sizeof(uint32_t)));
    HIPCHK(hipMemcpy(d_in, in.data(), 128 * sizeof(double), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_selftest, dim3(1), dim3(64), 0, 0, d_in, d_o, d_o + 2, d_r, d_o + 4);
    HIPCHK(hipGetLastError());
    double o[16];
    uint32_t r[8];
    HIPCHK(hipMemcpy(o, d_o, sizeof(o), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(r, d_r, sizeof(r), hipMemcpyDeviceToHost));
    (void)hipFree(d_in); (void)hipFree(d_o); (void)hipFree(d_r);
    if (memcmp(&o[0], &o[2], 2 * sizeof(double)) != 0)
        return fail(HTM_ESTATE, "DPP wave_sum mismatch: %.17g vs %.17g / %.17g vs %.17g", o[0], o[2], o[1], o[3]);
    // SURVEY.md §8a golden vector: first five rand_u() of rank 0
    const double want[5] = {0.55850877496413887, 0.12064291047863662, 0.58295862120576203, 0.68001799611374736,
                            0.45020412676967681};
    for (int i = 0; i < 5; ++i)
        if (o[4 + i] != want[i]) return fail(HTM_ESTATE, "device rand_u[%d] = %.17g, want %.17g", i, o[4 + i], want[i]);
    if (o[13] != 0.0) return fail(HTM_ESTATE, "DPP wave_incl_scan disagrees with the serial prefix sum");
    if (std::fabs(o[12] - 0.78381228502204603) > 1e-15)
        return fail(HTM_ESTATE, "device rand_g = %.17g, want 0.78381228502204603", o[12]);
    // jump-ahead generator == serial generator: one wave, several waves with a ragged tail, a ring wrap-around
    const uint32_t seed0[4] = {0x4b88a366u, 0x1b11733cu, 0x097044b6u, 0x00676ea2u};   // rank-0 state
    const uint32_t seed1[4] = {0x311ce1d7u, 0x6c840a86u, 0x28236c5fu, 0x019ea85du};   // rank 1
    if ((rc = selftest_rawgen(seed0, 64, 0, 1 << 12))) return rc;
    if ((rc = selftest_rawgen(seed0, 4096 * 3 + 64 * 5, 0, 1 << 14))) return rc;
    if ((rc = selftest_rawgen(seed1, 1 << 16, (1 << 16) - 4096 - 192, 1 << 16))) return rc;
    if ((rc = selftest_rawgen(seed1, 1 << 18, 12345 * 64, 1 << 18))) return rc;
    return HTM_OK;
}

}  // extern "C"
