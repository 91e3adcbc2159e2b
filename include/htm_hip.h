/*
 * htm_hip.h -- C ABI of libhtm_hip.so: the MI355X (gfx950) implementation of HypoTremorMCMC's
 * per-proposal likelihood inner loop (step 5, `hypo_tremor_mcmc`).
 *
 * The reference has no FFI for this path: the boundary is the Fortran derived type `forward`
 * (reference src/cls_forward.f90:6-41) plus the chain bookkeeping of `mcmc` / `parallel`
 * (src/cls_mcmc.f90:7-53, src/cls_parallel.f90:7-22) driven by `program main`
 * (src/hypo_tremor_mcmc.f90:236-284).  Each entry point below names the reference interface it
 * replaces.  The Fortran ISO_C_BINDING module that re-creates `type forward` on top of these symbols is
 * hypotremormcmc_amd/fortran/cls_forward_hip.f90 over htm_c_api.f90 (shown in INTEGRATION.md).
 *
 * Conventions
 *   - plain C, no C++/torch types; every function returns 0 on success, a negative HTM_E* code otherwise;
 *     htm_last_error() gives the message of the last failure on the calling thread.
 *   - all floating-point data is IEEE fp64; integers are 32-bit unless stated.
 *   - 2-D observation arrays are (n_sta, n_events) column-major exactly as in the reference
 *     (src/cls_forward.f90:62-69): element (j, i) at [i * n_sta + j], station index fastest.
 *   - hypocentre vectors are xyz-interleaved, 3 * n_events long (src/cls_forward.f90:109-111).
 *   - event ids (evt_id) are 1-based like the reference.
 *   - host pointers unless a parameter name starts with d_ (device pointer, same device as the handle).
 *   - a handle is not thread-safe; calls are synchronous unless stated (the reference is single-threaded
 *     per rank with strictly synchronous calls).
 *   - there is NO CPU fallback: every entry point fails with HTM_ENODEVICE when no gfx950 device is usable.
 */
#ifndef HTM_HIP_H
#define HTM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HTM_OK          0
#define HTM_EINVAL     -1   /* bad argument */
#define HTM_ENODEVICE  -2   /* no usable HIP device */
#define HTM_EHIP       -3   /* a HIP runtime call failed */
#define HTM_ESTATE     -4   /* call not valid in the handle's current state */
#define HTM_EOVERFLOW  -5   /* a record buffer overflowed (lock-step mode without drain) */
#define HTM_EDESYNC    -6   /* ranks disagree on the iteration number in the swap exchange */

typedef struct htm_forward htm_forward;
typedef struct htm_chains  htm_chains;

const char *htm_last_error(void);
int         htm_abi_version(void);          /* bumped when this header changes incompatibly */
int         htm_device_count(int *n);       /* number of visible HIP devices */
/* The GPU behind a device ordinal as (PCI domain << 16 | bus << 8 | device): the same number in every process that sees this
 * GPU, whatever ordinal HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES gave it there -- what ranks compare to find out that they
 * share a GPU (htm_chains_share_gpu; the reference has one MPI rank per CPU core and no such notion, src/mod_mpi.f90). */
int         htm_device_physical_id(int device, int *id);

/* ------------------------------------------------------------------------------------------------
 * `type forward`                                                  reference: src/cls_forward.f90
 * ---------------------------------------------------------------------------------------------- */

/* constructor `forward(n_sta, n_events, sta_x, sta_y, sta_z, obs, use_amp, use_time)`  (:40-96).
 * `obs` is passed as its four arrays (what obs%get_t_obs() .. get_a_stdv() return, :71-74).  All inputs
 * are copied (the reference deep-copies too, :50-74); the missing-data rule of :76-92 is applied here. */
int htm_forward_create(int n_sta, int n_events,
                       const double *sta_x, const double *sta_y, const double *sta_z,
                       const double *t_obs, const double *t_stdv,
                       const double *a_obs, const double *a_stdv,
                       int use_time, int use_amp, int device, htm_forward **out);
int htm_forward_destroy(htm_forward *h);

/* fp32 forward / fp64 accept (BASELINE configs[4]; the reference is fp64 throughout, so this mode has no reference
 * counterpart and carries a STATISTICAL tolerance): forward_fp32 = 1 makes every kernel of this
 * handle compute the synthetic travel times / amplitudes (src/cls_forward.f90:115-118, :201-204 -- distance, sqrt,
 * division, log) in single precision and stream the observations as float (half the bytes of a full
 * evaluation); the weighted demean sums (:125-132), residuals, the misfit sum (:281-299) and the Metropolis
 * decision (src/cls_mcmc.f90:194-199) stay fp64.  n_sta <= 128.  Call before htm_chains_create. */
int htm_forward_set_precision(htm_forward *h, int forward_fp32);

/* Run all work of this handle (and of chain sets created from it) on the caller's HIP stream, e.g.
 * torch's current stream so that RCCL collectives enqueued by torch.distributed order with our kernels.
 * hip_stream is used as given: NULL is HIP's default (null) stream -- which is what torch's default
 * stream is.  htm_forward_reset_stream goes back to the handle's own non-blocking stream. */
int htm_forward_set_stream(htm_forward *h, void *hip_stream);
int htm_forward_reset_stream(htm_forward *h);

/* `calc_log_likelihood(hypo, t_corr, vs, a_corr, qs, log_likelihood)`  (:268-303) */
int htm_forward_loglik_full(htm_forward *h, const double *hypo, const double *t_corr, double vs,
                            const double *a_corr, double qs, double *log_likelihood);

/* `partially_update_log_likelihood(evt_id, hypo_old, log_likelihood_old, hypo, t_corr, vs, a_corr, qs,
 * log_likelihood)`  (:307-362).  Only event evt_id of hypo_old / hypo is read by the reference, so the two
 * models are passed as that event's xyz triplets. */
int htm_forward_loglik_partial(htm_forward *h, int evt_id, const double hypo_old_xyz[3],
                               double log_likelihood_old, const double hypo_xyz[3],
                               const double *t_corr, double vs, const double *a_corr, double qs,
                               double *log_likelihood);

/* `calc_travel_time` (:100-138) / `calc_amp` (:183-222): out arrays are (n_sta, n_events), demeaned */
int htm_forward_travel_time(htm_forward *h, const double *hypo, const double *t_corr, double vs,
                            double *t_syn);
int htm_forward_amp(htm_forward *h, const double *hypo, const double *a_corr, double qs, double vs,
                    double *a_syn);
/* `calc_travel_time_single` (:142-179) / `calc_amp_single` (:226-264): out arrays are n_sta long */
int htm_forward_travel_time_single(htm_forward *h, int evt_id, const double *hypo, const double *t_corr,
                                   double vs, double *t_syn);
int htm_forward_amp_single(htm_forward *h, int evt_id, const double *hypo, const double *a_corr,
                           double qs, double vs, double *a_syn);

/* Batched full evaluation: n_models independent (hypo, t_corr, vs, a_corr, qs) sets in ONE launch --
 * what a chain-parallel caller needs instead of n_models calls of :268-303.  Stacked model-major:
 * hypo [n_models][3E], t_corr/a_corr [n_models][S], vs/qs/out [n_models]. */
int htm_forward_loglik_full_batch(htm_forward *h, int n_models, const double *hypo, const double *t_corr,
                                  const double *vs, const double *a_corr, const double *qs,
                                  double *log_likelihood);
/* Same with every array already resident in HBM (device pointers).  Asynchronous on the handle's stream;
 * d_log_likelihood is valid after htm_forward_sync().  This is the kernel bench.py prices against the
 * HBM roofline. */
int htm_forward_loglik_full_batch_dev(htm_forward *h, int n_models, const double *d_hypo,
                                      const double *d_t_corr, const double *d_vs, const double *d_a_corr,
                                      const double *d_qs, double *d_log_likelihood);
int htm_forward_sync(htm_forward *h);

/* Wall-clock-free timing of the last htm_forward_loglik_full_batch_dev launches: brackets `reps`
 * launches with HIP events on the handle's stream and returns the average kernel time in microseconds. */
int htm_forward_time_full_batch_dev(htm_forward *h, int n_models, const double *d_hypo,
                                    const double *d_t_corr, const double *d_vs, const double *d_a_corr,
                                    const double *d_qs, double *d_log_likelihood, int reps,
                                    double *avg_us);

/* ------------------------------------------------------------------------------------------------
 * Device-resident chains: `type mcmc` x n_chains inside `type parallel`
 *                          reference: src/cls_mcmc.f90, src/cls_parallel.f90, src/hypo_tremor_mcmc.f90
 * ---------------------------------------------------------------------------------------------- */

/* One `type model` (src/cls_model.f90:5-27) per parameter group, stacked chain-major.  Every pointer is
 * [n_chains][nx] with nx = 3*n_events (hypo), n_sta (t_corr, a_corr) or 1 (vs, qs).  mu/sigma/step_size/
 * prior_type may be NULL for a group that is never perturbed (solve_* = 0; the reference leaves them
 * unset too, src/hypo_tremor_mcmc.f90:133-137). */
typedef struct {
    const double  *x;
    const double  *mu;
    const double  *sigma;
    const double  *step_size;
    const int32_t *prior_type;   /* 0 Gaussian, 1 Rayleigh (src/cls_model.f90:9) */
} htm_model_init;

typedef struct {
    int            n_chains;      /* chains on this rank            (para%get_n_chains()) */
    int            n_procs;       /* ranks in the job               (mpi_comm_size)       */
    int            rank;          /* this rank                      (mpi_comm_rank)       */
    htm_model_init hypo, t_corr, vs, a_corr, qs;
    const double  *temp;          /* [n_chains] initial temperatures (src/hypo_tremor_mcmc.f90:202-208) */
    int            solve_vs, solve_t_corr, solve_qs, solve_a_corr;   /* -> proposal mix, src/cls_mcmc.f90:91-108 */
    uint32_t       rng_state[4];  /* mod_random's (x, y, z, w) after the rank's set-up draws */
    int            n_burn;        /* parameters are recorded only for i > n_burn (src/hypo_tremor_mcmc.f90:272) */
    int            n_interval;    /* record when mod(i, n_interval) == 1          (:271) */
    int            lik_capacity;  /* record-buffer sizes (records held on the device between drains); */
    int            sample_capacity; /* 0 = defaults */
} htm_chains_init;

int htm_chains_create(htm_forward *h, const htm_chains_init *init, htm_chains **out);
int htm_chains_destroy(htm_chains *hc);

/* Single-rank job (n_procs == 1): run n_iter iterations of the main loop
 * (src/hypo_tremor_mcmc.f90:236-284: propose -> forward -> judge per chain, then swap_temperature),
 * entirely on the device.  Synchronous; records are drained into host memory as needed. */
int htm_chains_run(htm_chains *hc, int n_iter);

/* Lock-step multi-rank iteration (n_procs >= 1).  step_begin enqueues propose -> forward -> judge for all
 * chains of this rank and fills this rank's swap record; the caller all-gathers the records of all ranks
 * (RCCL through torch.distributed, or MPI) into a device buffer [n_procs][record_bytes] and hands it to
 * step_end, which enqueues the temperature swap of src/cls_parallel.f90:100-216.  Nothing here blocks the
 * host; all work is stream-ordered. */
int htm_chains_step_begin(htm_chains *hc);
int htm_chains_swap_record(htm_chains *hc, void **d_record, size_t *record_bytes);
int htm_chains_step_end(htm_chains *hc, const void *d_gathered_records);
/* The same loop driven from C: n_iter lock-step iterations, each = step_begin + all-gather of the swap
 * records + step_end, all enqueued on the handle's stream without host synchronisation.  `allgather` has
 * ncclAllGather's signature (RCCL: pass &ncclAllGather and the rank's ncclComm_t; count is in elements of
 * dtype 8 = ncclFloat64); d_gathered must hold n_procs * record_bytes.  Returns the function's first
 * non-zero status as HTM_EHIP. */
typedef int (*htm_allgather_fn)(const void *sendbuff, void *recvbuff, size_t count, int dtype, void *comm,
                                void *stream);
int htm_chains_run_lockstep(htm_chains *hc, int n_iter, htm_allgather_fn allgather, void *comm,
                            void *d_gathered);
/* Host-staged form of the same exchange for MPI programs (the Fortran driver): after step_begin,
 * swap_record_host waits for the iteration and copies this rank's record (4 + 2*n_chains doubles) to host memory;
 * the caller all-gathers the records of all ranks (MPI_Allgather) and hands the n_procs records back with
 * step_end_host, which stages them in device memory.  Costs two small PCIe copies per iteration: a
 * compatibility path -- the RCCL path above keeps everything on the device. */
int htm_chains_swap_record_host(htm_chains *hc, double *record);
int htm_chains_step_end_host(htm_chains *hc, const double *gathered_records);
/* Persistent lock-step: the swap exchange of src/cls_parallel.f90:100-216 INSIDE the kernel.  Each rank owns an
 * inbox in its GPU's memory; a rank's kernel writes its swap record straight into every rank's inbox (peer-mapped
 * memory over xGMI, tagged 8-byte granules, system-scope stores) and polls its own -- no kernel boundary and no
 * collective per iteration, so the kernel stays resident over all iterations like the single-rank loop.
 *   xchg_handle   allocates the inbox and returns its IPC handle (HTM_XCHG_HANDLE_BYTES bytes);
 *   xchg_connect  takes the handles of ALL ranks in rank order (exchanged by the caller: MPI_Allgather, or
 *                 torch.distributed's all-gather over RCCL), `handle_bytes` apart, and maps the peers' inboxes
 *                 (n_procs == 1: handles may be NULL);
 *   xchg_probe    (optional, collective) a one-wave kernel writes a token into every rank's inbox and waits up to
 *                 `seconds` for all ranks' tokens in its own: proves at set-up that peer writes reach a polling kernel
 *                 (the call count of the set is part of the token: a repeated probe never passes on an earlier one's tokens);
 *   run_lockstep_direct  n_iter lock-step iterations; synchronous; every rank must call it with the same n_iter.
 * The transport-agnostic step_begin / step_end protocol above stays available (and is what a rank falls back to
 * when peer mapping is refused). */
#define HTM_XCHG_HANDLE_BYTES 64
int htm_chains_xchg_handle(htm_chains *hc, void *handle, size_t handle_bytes);
int htm_chains_xchg_connect(htm_chains *hc, const void *handles, size_t handle_bytes);
int htm_chains_xchg_probe(htm_chains *hc, unsigned token, double seconds);
int htm_chains_run_lockstep_direct(htm_chains *hc, int n_iter);
/* RCCL from a C / Fortran host (no Python): a communicator over librccl, bound at run time with dlopen (inside a
 * PyTorch process its copy is reused).  Rank 0 calls htm_comm_unique_id and broadcasts the HTM_COMM_ID_BYTES bytes
 * (MPI_Bcast in hypo_tremor_mcmc_hip_mpi); every rank then calls htm_comm_create (collective = ncclCommInitRank).
 * htm_chains_run_lockstep_comm = htm_chains_run_lockstep with ncclAllGather on that communicator: the swap of
 * src/cls_parallel.f90:100-216 as one RCCL all-gather per iteration, enqueued on the chains' stream without host
 * synchronisation.  htm_comm_allgather moves any bytes (e.g. the inboxes' IPC handles of the persistent lock-step).
 * RCCL wants one device per rank: two ranks on one GPU are refused by ncclCommInitRank. */
typedef struct htm_comm htm_comm;
#define HTM_COMM_ID_BYTES 128
int htm_comm_unique_id(void *id, size_t id_bytes);
int htm_comm_create(const void *id, size_t id_bytes, int rank, int n_ranks, int device, htm_comm **out);
int htm_comm_destroy(htm_comm *c);
int htm_comm_allgather(htm_comm *c, const void *d_send, void *d_recv, size_t bytes_per_rank, void *hip_stream);
int htm_chains_run_lockstep_comm(htm_chains *hc, int n_iter, htm_comm *c);
int htm_chains_sync(htm_chains *hc);        /* wait + raise device-side error flags */
int htm_chains_drain(htm_chains *hc);       /* sync + move device record buffers to host memory */

int htm_chains_iterations_done(htm_chains *hc, int *n);

/* Checkpoint / resume of a chain set (the reference has none, a 4 M-iteration production run
 * restarts from scratch).  The blob holds everything the main loop carries from one iteration to the next:
 * iteration counter, mod_random state at the consumed position, the five parameter vectors, temperatures,
 * log-likelihoods and proposal counters of every chain.  Priors, step sizes, observations and record
 * buffers are NOT in it: load into a chain set created from the same inputs (shapes are checked).
 * Continuing after load gives the bits of the uninterrupted run.  Not valid while a lock-step iteration is
 * in flight (between step_begin and step_end).  A load clears this rank's swap-record inbox (its tags are iteration
 * numbers, and iterations are about to be run again): for a multi-rank job every rank loads, then the ranks meet at a
 * barrier before the next htm_chains_run_lockstep_direct. */
int htm_chains_checkpoint_size(htm_chains *hc, size_t *bytes);
int htm_chains_checkpoint_save(htm_chains *hc, void *blob, size_t bytes);
int htm_chains_checkpoint_load(htm_chains *hc, const void *blob, size_t bytes);

/* chain state (`pt%get_mc(j)` + getters of src/cls_mcmc.f90:252-420); chain is 0-based; NULLs skipped */
int htm_chains_get_state(htm_chains *hc, int chain, double *hypo, double *t_corr, double *vs,
                         double *a_corr, double *qs, double *temp, double *log_likelihood,
                         int32_t n_propose[7], int32_t n_accept[7]);
int htm_chains_get_rng(htm_chains *hc, uint32_t state[4]);
/* Several ranks share this rank's GPU (`ranks_on_this_gpu` of them, this one included): every rank's persistent launch must be
 * resident at once (master and workers of a launch wait for each other), so each takes its share of the CUs.  Call before the
 * chain set's first launch; the environment variable HTM_RANKS_PER_GPU, if set, stands instead.  (The reference has no
 * counterpart: `mpirun -np N` ranks are CPU processes, src/cls_parallel.f90.) */
int htm_chains_share_gpu(htm_chains *hc, int ranks_on_this_gpu);
/* the chain's current log-likelihood alone (one 8-byte copy): what the reference's progress report prints every 1 000
 * iterations for chain 1 (`mc%one_step_summary`, src/cls_mcmc.f90:230-237) */
int htm_chains_get_loglik(htm_chains *hc, int chain, double *log_likelihood);

/* What the reference streams to likelihoodRR.out (src/hypo_tremor_mcmc.f90:279): (iteration, value) in
 * file order, plus the chain index that produced it. */
int htm_chains_lik_count(htm_chains *hc, int *n);
int htm_chains_lik_read(htm_chains *hc, int32_t *iter, int32_t *chain, double *log_likelihood);
/* What goes to vs/hypo/t_corr/qs/a_corr.RR.out (:273-277), sample k in file order */
int htm_chains_sample_count(htm_chains *hc, int *n);
int htm_chains_sample_read(htm_chains *hc, int k, int32_t *iter, int32_t *chain, double *vs, double *qs,
                           double *hypo, double *t_corr, double *a_corr);
int htm_chains_clear_records(htm_chains *hc);

/* Per-step trace for parity debugging: rows {iter, chain, type(1..7), index(1-based), prior_ok, accepted,
 * used_full, 0} and {x_new, loglik_proposed, loglik_after, temp}.  capacity 0 disables. */
int htm_chains_enable_steplog(htm_chains *hc, int capacity);
int htm_chains_steplog_read(htm_chains *hc, int *n, int32_t *irows, double *drows);

/* Kernel timing collected with HIP events inside htm_chains_run (stream-ordered, no extra syncs):
 * total device time of the last run in microseconds and the number of graph replays issued. */
int htm_chains_last_run_stats(htm_chains *hc, double *device_us, int *graph_launches,
                              int64_t *full_evals, int64_t *partial_evals);

/* Health of the in-kernel hand-off since the chain set was created: *orders_put_aside = how many times worker block 0 put an
 * order aside because the commit it names did not show within 20 us (0 in a healthy run: the chain waves take such orders
 * back themselves; the other worker blocks behave alike).  Synchronises with the handle's stream. */
int htm_chains_handoff_stats(htm_chains *hc, int64_t *orders_put_aside);

/* Which main loop the single-rank / lock-step launches of this chain set run on, and the health of its speculation:
 * *single_rank_loop / *lockstep_loop = 5 / 6 the pipelined master (csrc/htm_pipe.hpp), 3 / 4 the free-running chain master
 * (csrc/htm_flow.hpp), 0 / 2 the loop with workgroup barriers, -1 the two-kernel path; *flushes = how often the pipelined
 * master threw its speculative records away since the chain set was created (rare at production sizes).  No reference
 * counterpart (diagnostics).  Synchronises with the handle's stream. */
int htm_chains_master_stats(htm_chains *hc, int *single_rank_loop, int *lockstep_loop, int64_t *flushes);

/* Same work as htm_chains_run, but every kernel is launched eagerly and bracketed by its own pair of HIP
 * events on the handle's stream, so that the average duration of each kernel comes from the run itself:
 * k_step (proposals + partial updates + judge + swap; may cover several iterations per launch) and k_full
 * (batched full evaluations).  Returns sums in microseconds and launch counts; the chains advance by
 * n_iter iterations exactly as with htm_chains_run. */
int htm_chains_profile(htm_chains *hc, int n_iter, double *step_us, int *step_launches, double *full_us,
                       int *full_launches, int64_t *full_evals, int64_t *partial_evals);

/* self-test of the wave-level reduction and RNG device code against straightforward device loops;
 * returns 0 when they agree bit for bit */
/* ------------------------------------------------------------------------------------------------
 * Step-6 order statistics       reference: src/cls_statistics.f90:216-264, :345-431
 * The reference sorts every parameter's n_mod recorded samples (quick_sort, src/mod_sort.f90) and prints
 * the elements il = int(0.025*n_mod), im = int(0.5*n_mod), iu = int(0.975*n_mod) (1-based, single-
 * precision products) of the sorted column.  htm_quantiles returns exactly those elements without sorting:
 * samples [n_mod][n_par] row-major (one recorded model per row), ranks_1based[3] = {il, im, iu} or any other
 * three ranks in 1..n_mod, out [n_par][3].  Host pointers; synchronous.  The _dev form takes device
 * pointers (ld = row stride in doubles) and is asynchronous on `hip_stream` (NULL = the null stream). */
int htm_quantiles(int device, const double *samples, long n_mod, long n_par, const int ranks_1based[3],
                  double *out);
int htm_quantiles_dev(int device, const double *d_samples, long n_mod, long n_par, long ld,
                      const int ranks_1based[3], double *d_out, void *hip_stream);

/* ------------------------------------------------------------------------------------------------
 * Step 4, `hypo_tremor_select`   reference: src/cls_selector.f90:75-132, src/mod_regress.f90
 * For every detected window: the station of largest amplitude is taken as the epicentre (depth z_guess), the
 * amplitudes are corrected for geometrical spreading (+ ln d), and arrival time and amplitude are regressed
 * against distance with weights 1/err^2.  t, t_err, a, a_err: (n_sta, n_win) column-major = the columns 4-7 of the
 * opt_data.NNNNNN.dat files; out[n_win][6] = {vs, b, t0, a0, cc_t, cc_a}: the columns of a regress.dat row after
 * the window id (src/hypo_tremor_select.f90:124-125).  Host pointers; synchronous. */
int htm_select_regress(int device, int n_sta, int n_win, const double *sta_x, const double *sta_y,
                       const double *sta_z, double z_guess, const double *t, const double *t_err,
                       const double *a, const double *a_err, double *out);

int htm_selftest(int device);

/* y[i] = fn(x[i]) for n host values, fn = the forward model's own fp64 routines: which = 0 the logarithm of the amplitude
 * term (reference src/cls_forward.f90:204, `log(d)`), 1 the square root of the squared distance (:115-117), 2 the device
 * library's sqrt (what 1 must equal), 3 the device library's log -- the one used by the Rayleigh prior ratio
 * (reference src/cls_model.f90:184-185, `log(x_new - mu) - log(x_old - mu)`: a term of the Metropolis decision, so the test
 * compares it with the host libm the reference links).  Lets a test measure them against a wider-precision result (stated
 * bound of the logarithm: < 1 ulp).  The wave-level sums of the likelihood (cls_forward.f90:125-132, :210-217, :281-299) have
 * their own modes: 4 = every lane's sum of its wave's 64 values on the matrix pipe (n a multiple of 64), 5 / 6 = x is four blocks
 * of n / 4 values, a wave's four sums taken at once (5) or one by one (6) -- a test compares the associations bit for bit
 * (n a multiple of 256).  Synchronous. */
int htm_selftest_math(int device, int which, const double *x, double *y, int n);

/* mod_random's generator (reference src/mod_random.f90:60-74) is linear over GF(2): the state after n draws is
 * T^n * state.  Host-only (no device needed): used to seek in a rank's stream and by the tests that pin the
 * jump tables of the device generator.  state = {x, y, z, w} as in mod_random.f90:30. */
int htm_rng_jump(const uint32_t state_in[4], unsigned long long n_draws, uint32_t state_out[4]);

#ifdef __cplusplus
}
#endif
#endif /* HTM_HIP_H */
