/*
 * htm_oracle.h -- CPU restatement of HypoTremorMCMC step 5 (TEST INFRASTRUCTURE ONLY).
 *
 * This is the parity oracle for the HIP path.  It is NOT part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The product library
 * (hypotremormcmc_amd/lib/libhtm_hip.so) never links, loads or calls anything in oracle/.
 *
 * Every function cites the reference file:line (relative to the akuhara/HypoTremorMCMC tree) whose
 * algorithm it restates.  Parity is PINNED: the reference has no tests of its own, so the
 * restatement is checked against fixtures produced by the reference itself, compiled unmodified with
 * AMD flang into oracle/_ref (recipe: oracle/Makefile; fixtures + generator: tests/golden/).
 *
 * Plain C99, fp64 everywhere, compiled WITHOUT -ffast-math / FMA contraction so that every per-element
 * expression rounds exactly like the Fortran one.
 */
#ifndef HTM_ORACLE_H
#define HTM_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- mod_random (src/mod_random.f90:30-112): process-global xorshift128, here an explicit state ---- */
typedef struct { uint32_t x, y, z, w; } orc_rng;
void   orc_rng_seed(orc_rng *s, int32_t i1, int32_t i2, int32_t i3, int32_t i4, int32_t rank); /* :39-55 */
double orc_rand_u(orc_rng *s);   /* [0,1)  :60-74  */
double orc_rand_u2(orc_rng *s);  /* (0,1)  :78-92  */
double orc_rand_g(orc_rng *s);   /* :95-102  */
double orc_rand_r(orc_rng *s);   /* :106-112 */

/* ---- cls_model (src/cls_model.f90:5-27): parameter vector with priors and step sizes ---- */
typedef struct {
    int      nx;
    int32_t *prior_type;   /* 0 Gaussian, 1 Rayleigh */
    double  *x, *mu, *sigma, *step_size;
} orc_model;

/* ---- cls_forward (src/cls_forward.f90) ---- */
typedef struct orc_forward orc_forward;
/* arrays t_obs.. are (n_sta, n_events) column-major: element (j,i) at [i*n_sta + j] */
orc_forward *orc_forward_create(int n_sta, int n_events, const double *sta_x, const double *sta_y,
                                const double *sta_z, const double *t_obs, const double *t_stdv,
                                const double *a_obs, const double *a_stdv, int use_time, int use_amp);
void   orc_forward_destroy(orc_forward *f);
void   orc_forward_travel_time(const orc_forward *f, const double *hypo, const double *t_corr, double vs,
                               double *t_syn);                                   /* :100-138 */
void   orc_forward_amp(const orc_forward *f, const double *hypo, const double *a_corr, double qs,
                       double vs, double *a_syn);                                /* :183-222 */
void   orc_forward_travel_time_single(const orc_forward *f, int evt_id, const double *hypo,
                                      const double *t_corr, double vs, double *t_syn); /* :142-179 */
void   orc_forward_amp_single(const orc_forward *f, int evt_id, const double *hypo, const double *a_corr,
                              double qs, double vs, double *a_syn);              /* :226-264 */
double orc_forward_loglik_full(const orc_forward *f, const double *hypo, const double *t_corr, double vs,
                               const double *a_corr, double qs);                 /* :268-303 */
double orc_forward_loglik_partial(const orc_forward *f, int evt_id, const double *hypo_old,
                                  double loglik_old, const double *hypo, const double *t_corr, double vs,
                                  const double *a_corr, double qs);              /* :307-362 */

/* ---- whole step-5 job: n_procs simulated MPI ranks x n_chains chains, run in lock step ---- */
typedef struct {
    int    n_procs, n_chains, n_cool;
    int    n_iter, n_burn, n_interval;
    double temp_high;
    double prior_z, prior_width_z, prior_width_xy;
    double prior_vs, prior_width_vs, prior_qs, prior_width_qs;
    double prior_t_corr, prior_width_t_corr, prior_a_corr, prior_width_a_corr;
    double step_size_z, step_size_xy, step_size_vs, step_size_qs, step_size_t_corr, step_size_a_corr;
    int    solve_vs, solve_t_corr, solve_qs, solve_a_corr;
    int    use_time, use_amp;
} orc_params;

typedef struct orc_job orc_job;

/* Builds ranks/chains exactly as src/hypo_tremor_mcmc.f90:72,:98,:120-211 does (RNG draws included). */
orc_job *orc_job_create(const orc_params *p, int n_sta, int n_events, const double *sta_x,
                        const double *sta_y, const double *sta_z, const double *t_obs,
                        const double *t_stdv, const double *a_obs, const double *a_stdv);
void     orc_job_destroy(orc_job *job);
/* Runs iterations i = i_done+1 .. i_done+n (main loop src/hypo_tremor_mcmc.f90:236-284). */
void     orc_job_run(orc_job *job, int n);

/* Per-rank lock-step mode: the same job advanced one rank at a time with the swap decided from
 * all-gathered records -- the protocol the HIP path uses across GPUs.  Record layout, 8-byte
 * words: [0] i1, [1] i2 (global chain indices chosen by rank 0, -1 elsewhere), [2] this rank's pending
 * judge_swap draw (peeked, consumed only if the rank turns out to be rank1), [3] iteration,
 * [4+2c] temperature and [5+2c] log-likelihood of chain c.  Do not mix with orc_job_run on one job. */
int      orc_job_record_words(const orc_job *job);
void     orc_job_rank_begin(orc_job *job, int rank, double *record);
int      orc_job_rank_end(orc_job *job, int rank, const double *gathered); /* 0 ok, -6 iteration mismatch */

/* Recorded output (what the reference writes to likelihoodRR.out / hypo.RR.out / ...) */
int      orc_job_n_lik(const orc_job *job, int rank);
void     orc_job_get_lik(const orc_job *job, int rank, int32_t *iter, double *lik);
int      orc_job_n_samples(const orc_job *job, int rank);
/* sample k of rank: iter, vs, qs, hypo[3E], t_corr[S], a_corr[S] */
void     orc_job_get_sample(const orc_job *job, int rank, int k, int32_t *iter, double *vs, double *qs,
                            double *hypo, double *t_corr, double *a_corr);
/* proposal_count.txt content (src/cls_parallel.f90:244-281): sums over all ranks and chains */
void     orc_job_get_counts(const orc_job *job, int32_t n_propose[7], int32_t n_accept[7]);
/* chain state access */
void     orc_job_get_chain(const orc_job *job, int rank, int chain, double *hypo, double *t_corr,
                           double *vs, double *a_corr, double *qs, double *temp, double *loglik,
                           int32_t n_propose[7], int32_t n_accept[7]);
void     orc_job_get_priors(const orc_job *job, int rank, int chain, double *hypo_mu, double *hypo_sigma,
                            double *hypo_step, int32_t *hypo_ptype);
void     orc_job_get_rng(const orc_job *job, int rank, uint32_t state[4]);
/* per-step debug trace (optional): rows of {iter, rank, chain, type, index(1-based), prior_ok, accepted,
 * used_full} + {x_new, loglik_proposed, loglik_current_after, temp}.  cap = max rows kept (0 disables). */
void     orc_job_enable_steplog(orc_job *job, int cap);
int      orc_job_steplog_n(const orc_job *job);
void     orc_job_get_steplog(const orc_job *job, int32_t *irows /* n x 8 */, double *drows /* n x 4 */);

/* step 4 (hypo_tremor_select): regressions of one window set, out[n_win][6] = {vs, b, t0, a0, cc_t, cc_a}
 * (src/cls_selector.f90:75-132, src/mod_regress.f90) */
void     orc_select_regress(int n_sta, int n_win, const double *sta_x, const double *sta_y, const double *sta_z,
                            double z_guess, const double *t, const double *t_err, const double *a,
                            const double *a_err, double *out);

#ifdef __cplusplus
}
#endif
#endif
