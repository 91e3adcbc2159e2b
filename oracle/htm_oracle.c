/*
 * htm_oracle.c -- CPU restatement of HypoTremorMCMC step 5.  TEST INFRASTRUCTURE ONLY (see htm_oracle.h).
 *
 * Follows the reference's operation order expression by expression (serial left-to-right sums, the same
 * parenthesisation, libm sqrt/log/cos) so that it tracks the flang-compiled reference to rounding level.
 * Differences by design: chain objects are updated in place (one scalar per proposal) instead of being
 * deep-copied (src/hypo_tremor_mcmc.f90:238,:265; src/cls_mcmc.f90:128-132,:209-213) -- the arithmetic and
 * the random-number consumption order are unchanged; MPI ranks are simulated in one process, in lock step.
 */
#include "htm_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------------ */
/* mod_random                                                                                       */
/* ------------------------------------------------------------------------------------------------ */

/* src/mod_random.f90:49-52 -- default-integer (int32) arithmetic that wraps; uint32 gives the same bits */
static uint32_t seed_word(uint32_t i, uint32_t j1)
{
    uint32_t j2 = j1 * j1;
    uint32_t j4 = j2 * j2;
    return i * j4 + 1000u * i * j2 + i;
}

void orc_rng_seed(orc_rng *s, int32_t i1, int32_t i2, int32_t i3, int32_t i4, int32_t rank)
{
    uint32_t j1 = (uint32_t)rank + 1u;
    s->x = seed_word((uint32_t)i1, j1);
    s->y = seed_word((uint32_t)i2, j1);
    s->z = seed_word((uint32_t)i3, j1);
    s->w = seed_word((uint32_t)i4, j1);
}

/* src/mod_random.f90:63-71 -- ishft is a logical shift */
static uint32_t rng_next(orc_rng *s)
{
    uint32_t t = s->x ^ (s->x << 11);
    s->x = s->y;
    s->y = s->z;
    s->z = s->w;
    s->w = (s->w ^ (s->w >> 19)) ^ (t ^ (t >> 8));
    return s->w;
}

double orc_rand_u(orc_rng *s)
{
    /* :72  (dble(w) + r31) / r32, w interpreted as a signed int32 */
    return ((double)(int32_t)rng_next(s) + 2147483648.0) / 4294967296.0;
}

double orc_rand_u2(orc_rng *s)
{
    /* :90  (dble(w) + r31 + 0.5d0) / r32 */
    return ((double)(int32_t)rng_next(s) + 2147483648.0 + 0.5) / 4294967296.0;
}

double orc_rand_g(orc_rng *s)
{
    /* :98-100 */
    const double pi2 = 2.0 * acos(-1.0);
    double v1 = orc_rand_u2(s);
    double v2 = orc_rand_u2(s);
    return sqrt(-2.0 * log(v1)) * cos(pi2 * v2);
}

double orc_rand_r(orc_rng *s)
{
    /* :109-110 */
    double u = orc_rand_u2(s);
    return sqrt(-2.0 * log(u));
}

/* ------------------------------------------------------------------------------------------------ */
/* cls_forward                                                                                      */
/* ------------------------------------------------------------------------------------------------ */

struct orc_forward {
    int     n_sta, n_events;
    double *sta_x, *sta_y, *sta_z;
    double *t_obs, *t_stdv, *t_precision, *log_t_stdv;
    double *a_obs, *a_stdv, *a_precision, *log_a_stdv;
    int     use_time, use_amp;
    double  log_2pi_half;
};

static double *dup_d(const double *src, size_t n)
{
    double *p = (double *)malloc(n * sizeof(double));
    memcpy(p, src, n * sizeof(double));
    return p;
}

orc_forward *orc_forward_create(int n_sta, int n_events, const double *sta_x, const double *sta_y,
                                const double *sta_z, const double *t_obs, const double *t_stdv,
                                const double *a_obs, const double *a_stdv, int use_time, int use_amp)
{
    orc_forward *f = (orc_forward *)calloc(1, sizeof(*f));
    size_t n = (size_t)n_sta * (size_t)n_events;
    f->n_sta = n_sta;
    f->n_events = n_events;
    f->sta_x = dup_d(sta_x, n_sta);
    f->sta_y = dup_d(sta_y, n_sta);
    f->sta_z = dup_d(sta_z, n_sta);
    f->t_obs = dup_d(t_obs, n);
    f->t_stdv = dup_d(t_stdv, n);
    f->a_obs = dup_d(a_obs, n);
    f->a_stdv = dup_d(a_stdv, n);
    f->t_precision = (double *)malloc(n * sizeof(double));
    f->a_precision = (double *)malloc(n * sizeof(double));
    f->log_t_stdv = (double *)malloc(n * sizeof(double));
    f->log_a_stdv = (double *)malloc(n * sizeof(double));
    f->use_time = use_time;
    f->use_amp = use_amp;
    f->log_2pi_half = 0.5 * log(2.0 * acos(-1.0)); /* src/cls_forward.f90:5 */
    /* src/cls_forward.f90:76-92 -- missing-data rule keyed on t_stdv only; log-stdv := 1.0 (sic) */
    for (size_t k = 0; k < n; ++k) {
        if (f->t_stdv[k] > 1.e-16) {
            f->log_t_stdv[k] = log(f->t_stdv[k]);
            f->t_precision[k] = 1.0 / (f->t_stdv[k] * f->t_stdv[k]);
            f->log_a_stdv[k] = log(f->a_stdv[k]);
            f->a_precision[k] = 1.0 / (f->a_stdv[k] * f->a_stdv[k]);
        } else {
            f->log_t_stdv[k] = 1.0;
            f->t_stdv[k] = 1.0;
            f->t_precision[k] = 1.0;
            f->log_a_stdv[k] = 1.0;
            f->a_stdv[k] = 1.0;
            f->a_precision[k] = 1.0;
        }
    }
    return f;
}

void orc_forward_destroy(orc_forward *f)
{
    if (!f) return;
    free(f->sta_x); free(f->sta_y); free(f->sta_z);
    free(f->t_obs); free(f->t_stdv); free(f->t_precision); free(f->log_t_stdv);
    free(f->a_obs); free(f->a_stdv); free(f->a_precision); free(f->log_a_stdv);
    free(f);
}

/* one event's travel times incl. demean: src/cls_forward.f90:113-120 + :124-133 (== :155-173) */
static void tt_event(const orc_forward *f, int i /*0-based*/, double x, double y, double z,
                     const double *t_corr, double beta, double *t_syn /* n_sta */)
{
    const int S = f->n_sta;
    const double *prec = f->t_precision + (size_t)i * S;
    const double *obs = f->t_obs + (size_t)i * S;
    for (int j = 0; j < S; ++j) {
        double dx = x - f->sta_x[j], dy = y - f->sta_y[j], dz = z - f->sta_z[j];
        t_syn[j] = sqrt(dx * dx + dy * dy + dz * dz) / beta - t_corr[j];
    }
    double num = 0.0, den = 0.0;
    for (int j = 0; j < S; ++j) num += prec[j] * (t_syn[j] - obs[j]);
    for (int j = 0; j < S; ++j) den += prec[j];
    double t_mean = num / den;
    for (int j = 0; j < S; ++j) t_syn[j] = t_syn[j] - t_mean;
}

/* one event's log-amplitudes incl. demean: src/cls_forward.f90:199-205 + :209-218 (== :242-259) */
static void amp_event(const orc_forward *f, int i, double x, double y, double z, const double *a_corr,
                      double q, double beta, double *a_syn)
{
    const int S = f->n_sta;
    const double pi = acos(-1.0);
    const double freq = 5.0; /* :190 */
    const double *prec = f->a_precision + (size_t)i * S;
    const double *obs = f->a_obs + (size_t)i * S;
    for (int j = 0; j < S; ++j) {
        double dx = x - f->sta_x[j], dy = y - f->sta_y[j], dz = z - f->sta_z[j];
        double d = sqrt(dx * dx + dy * dy + dz * dz);
        a_syn[j] = -(d * pi * freq / (q * beta)) - log(d) - a_corr[j]; /* :204 */
    }
    double num = 0.0, den = 0.0;
    for (int j = 0; j < S; ++j) num += prec[j] * (a_syn[j] - obs[j]);
    for (int j = 0; j < S; ++j) den += prec[j];
    double a_mean = num / den;
    for (int j = 0; j < S; ++j) a_syn[j] = a_syn[j] - a_mean;
}

void orc_forward_travel_time(const orc_forward *f, const double *hypo, const double *t_corr, double vs,
                             double *t_syn)
{
    for (int i = 0; i < f->n_events; ++i)
        tt_event(f, i, hypo[3 * i], hypo[3 * i + 1], hypo[3 * i + 2], t_corr, vs,
                 t_syn + (size_t)i * f->n_sta);
}

void orc_forward_amp(const orc_forward *f, const double *hypo, const double *a_corr, double qs, double vs,
                     double *a_syn)
{
    for (int i = 0; i < f->n_events; ++i)
        amp_event(f, i, hypo[3 * i], hypo[3 * i + 1], hypo[3 * i + 2], a_corr, qs, vs,
                  a_syn + (size_t)i * f->n_sta);
}

void orc_forward_travel_time_single(const orc_forward *f, int evt_id, const double *hypo,
                                    const double *t_corr, double vs, double *t_syn)
{
    int i = evt_id - 1;
    tt_event(f, i, hypo[3 * i], hypo[3 * i + 1], hypo[3 * i + 2], t_corr, vs, t_syn);
}

void orc_forward_amp_single(const orc_forward *f, int evt_id, const double *hypo, const double *a_corr,
                            double qs, double vs, double *a_syn)
{
    int i = evt_id - 1;
    amp_event(f, i, hypo[3 * i], hypo[3 * i + 1], hypo[3 * i + 2], a_corr, qs, vs, a_syn);
}

double orc_forward_loglik_full(const orc_forward *f, const double *hypo, const double *t_corr, double vs,
                               const double *a_corr, double qs)
{
    const int S = f->n_sta, E = f->n_events;
    double *syn = (double *)malloc((size_t)S * E * sizeof(double));
    double L = 0.0;
    if (f->use_time) { /* :279-289 */
        orc_forward_travel_time(f, hypo, t_corr, vs, syn);
        for (int i = 0; i < E; ++i)
            for (int j = 0; j < S; ++j) {
                size_t k = (size_t)i * S + j;
                double r = f->t_obs[k] - syn[k];
                L = L - r * r / (2.0 * (f->t_stdv[k] * f->t_stdv[k])) - f->log_2pi_half - f->log_t_stdv[k];
            }
    }
    if (f->use_amp) { /* :290-300 */
        orc_forward_amp(f, hypo, a_corr, qs, vs, syn);
        for (int i = 0; i < E; ++i)
            for (int j = 0; j < S; ++j) {
                size_t k = (size_t)i * S + j;
                double r = f->a_obs[k] - syn[k];
                L = L - r * r / (2.0 * (f->a_stdv[k] * f->a_stdv[k])) - f->log_2pi_half - f->log_a_stdv[k];
            }
    }
    free(syn);
    return L;
}

double orc_forward_loglik_partial(const orc_forward *f, int evt_id, const double *hypo_old,
                                  double loglik_old, const double *hypo, const double *t_corr, double vs,
                                  const double *a_corr, double qs)
{
    const int S = f->n_sta;
    const int i = evt_id - 1;
    double *syn = (double *)malloc((size_t)S * sizeof(double));
    double L = loglik_old; /* :319 */
    if (f->use_time) {
        orc_forward_travel_time_single(f, evt_id, hypo_old, t_corr, vs, syn); /* :321-328 */
        for (int j = 0; j < S; ++j) {
            size_t k = (size_t)i * S + j;
            double r = f->t_obs[k] - syn[j];
            L = L + r * r / (2.0 * (f->t_stdv[k] * f->t_stdv[k])) + f->log_2pi_half + f->log_t_stdv[k];
        }
        orc_forward_travel_time_single(f, evt_id, hypo, t_corr, vs, syn); /* :330-337 */
        for (int j = 0; j < S; ++j) {
            size_t k = (size_t)i * S + j;
            double r = f->t_obs[k] - syn[j];
            L = L - r * r / (2.0 * (f->t_stdv[k] * f->t_stdv[k])) - f->log_2pi_half - f->log_t_stdv[k];
        }
    }
    if (f->use_amp) {
        orc_forward_amp_single(f, evt_id, hypo_old, a_corr, qs, vs, syn); /* :341-348 */
        for (int j = 0; j < S; ++j) {
            size_t k = (size_t)i * S + j;
            double r = f->a_obs[k] - syn[j];
            L = L + r * r / (2.0 * (f->a_stdv[k] * f->a_stdv[k])) + f->log_2pi_half + f->log_a_stdv[k];
        }
        orc_forward_amp_single(f, evt_id, hypo, a_corr, qs, vs, syn); /* :350-357 */
        for (int j = 0; j < S; ++j) {
            size_t k = (size_t)i * S + j;
            double r = f->a_obs[k] - syn[j];
            L = L - r * r / (2.0 * (f->a_stdv[k] * f->a_stdv[k])) - f->log_2pi_half - f->log_a_stdv[k];
        }
    }
    free(syn);
    return L;
}

/* ------------------------------------------------------------------------------------------------ */
/* cls_model                                                                                        */
/* ------------------------------------------------------------------------------------------------ */

static void model_alloc(orc_model *m, int nx)
{
    m->nx = nx;
    m->prior_type = (int32_t *)calloc(nx, sizeof(int32_t));
    m->x = (double *)calloc(nx, sizeof(double));
    m->mu = (double *)calloc(nx, sizeof(double));
    m->sigma = (double *)calloc(nx, sizeof(double));
    m->step_size = (double *)calloc(nx, sizeof(double));
}

static void model_free(orc_model *m)
{
    free(m->prior_type); free(m->x); free(m->mu); free(m->sigma); free(m->step_size);
}

/* src/cls_model.f90:139-158 */
static void model_generate(orc_model *m, orc_rng *rng)
{
    for (int i = 0; i < m->nx; ++i) {
        if (m->prior_type[i] == 0)
            m->x[i] = m->mu[i] + orc_rand_g(rng) * m->sigma[i];
        else
            m->x[i] = m->mu[i] + orc_rand_r(rng) * m->sigma[i];
    }
}

/* src/cls_model.f90:162-190 -- returns x_new, does NOT store it (the caller keeps old and new apart) */
static double model_perturb(const orc_model *m, int i /*0-based*/, orc_rng *rng, double *log_prior_ratio,
                            int *prior_ok)
{
    *prior_ok = 1;
    double x_old = m->x[i];
    double x_new = x_old + orc_rand_g(rng) * m->step_size[i];
    double a = x_new - m->mu[i], b = x_old - m->mu[i];
    *log_prior_ratio = -(a * a - b * b) / (2.0 * m->sigma[i] * m->sigma[i]);
    if (m->prior_type[i] == 1) {
        if (x_new <= m->mu[i]) {
            *log_prior_ratio = (double)-1.0e+30f; /* :180, a default-real literal */
            *prior_ok = 0;
        } else {
            *log_prior_ratio = *log_prior_ratio + log(x_new - m->mu[i]) - log(x_old - m->mu[i]);
        }
    }
    return x_new;
}

/* ------------------------------------------------------------------------------------------------ */
/* cls_mcmc + cls_parallel + program main                                                            */
/* ------------------------------------------------------------------------------------------------ */

typedef struct {
    orc_model hypo, t_corr, vs, a_corr, qs;
    double    log_likelihood; /* src/cls_mcmc.f90:88  = -9d300 */
    double    temp;
    int32_t   n_propose[7], n_accept[7];
    int       i_iter;
} orc_chain;

typedef struct {
    int32_t  iter;
    double   vs, qs;
    double  *hypo, *t_corr, *a_corr;
} orc_sample;

typedef struct {
    orc_rng     rng;
    orc_chain  *chains;
    /* likelihoodRR.out */
    int         n_lik, cap_lik;
    int32_t    *lik_iter;
    double     *lik_val;
    /* hypo.RR.out & friends */
    int         n_smp, cap_smp;
    orc_sample *smp;
    int         iter;      /* per-rank lock-step mode: iterations completed by this rank */
} orc_rank;

struct orc_job {
    orc_params   p;
    int          n_sta, n_events;
    orc_forward *fwd;
    orc_rank    *ranks;
    int          i_done;
    double       p_vs, p_t_corr, p_qs, p_a_corr; /* src/cls_mcmc.f90:91-106 */
    double      *scratch_hypo;
    /* step log */
    int          slog_cap, slog_n;
    int32_t     *slog_i;
    double      *slog_d;
};

static const double EPS = 2.220446049250313e-16; /* epsilon(1.d0) */

orc_job *orc_job_create(const orc_params *p, int n_sta, int n_events, const double *sta_x,
                        const double *sta_y, const double *sta_z, const double *t_obs,
                        const double *t_stdv, const double *a_obs, const double *a_stdv)
{
    orc_job *job = (orc_job *)calloc(1, sizeof(*job));
    job->p = *p;
    job->n_sta = n_sta;
    job->n_events = n_events;
    job->fwd = orc_forward_create(n_sta, n_events, sta_x, sta_y, sta_z, t_obs, t_stdv, a_obs, a_stdv,
                                  p->use_time, p->use_amp);
    job->scratch_hypo = (double *)malloc(3 * (size_t)n_events * sizeof(double));
    job->p_vs = p->solve_vs ? 0.025 : 0.0;
    job->p_t_corr = p->solve_t_corr ? 0.025 : 0.0;
    job->p_qs = p->solve_qs ? 0.025 : 0.0;
    job->p_a_corr = p->solve_a_corr ? 0.025 : 0.0;

    /* obs%make_initial_guess: src/cls_obs_data.f90:120-134 (maxloc = first maximum) */
    double *x_mu = (double *)malloc(n_events * sizeof(double));
    double *y_mu = (double *)malloc(n_events * sizeof(double));
    for (int i = 0; i < n_events; ++i) {
        int ista = 0;
        for (int j = 1; j < n_sta; ++j)
            if (a_obs[(size_t)i * n_sta + j] > a_obs[(size_t)i * n_sta + ista]) ista = j;
        x_mu[i] = sta_x[ista];
        y_mu[i] = sta_y[ista];
    }

    job->ranks = (orc_rank *)calloc(p->n_procs, sizeof(orc_rank));
    for (int r = 0; r < p->n_procs; ++r) {
        orc_rank *rk = &job->ranks[r];
        orc_rng_seed(&rk->rng, 5551111, 453222, 4444431, 6765, r); /* src/hypo_tremor_mcmc.f90:72 */
        rk->chains = (orc_chain *)calloc(p->n_chains, sizeof(orc_chain));
        for (int j = 0; j < p->n_chains; ++j) { /* src/hypo_tremor_mcmc.f90:120-211 */
            orc_chain *c = &rk->chains[j];
            model_alloc(&c->t_corr, n_sta);
            if (p->solve_t_corr) {
                for (int i = 0; i < n_sta; ++i) {
                    c->t_corr.mu[i] = p->prior_t_corr;
                    c->t_corr.sigma[i] = p->prior_width_t_corr;
                    c->t_corr.step_size[i] = p->step_size_t_corr;
                }
                model_generate(&c->t_corr, &rk->rng);
            } else {
                for (int i = 0; i < n_sta; ++i) c->t_corr.x[i] = p->prior_t_corr;
            }
            model_alloc(&c->a_corr, n_sta);
            if (p->solve_a_corr) {
                for (int i = 0; i < n_sta; ++i) {
                    c->a_corr.mu[i] = p->prior_a_corr;
                    c->a_corr.sigma[i] = p->prior_width_a_corr;
                    c->a_corr.step_size[i] = p->step_size_a_corr;
                }
                model_generate(&c->a_corr, &rk->rng);
            } else {
                for (int i = 0; i < n_sta; ++i) c->a_corr.x[i] = p->prior_a_corr;
            }
            model_alloc(&c->hypo, 3 * n_events);
            for (int i = 0; i < n_events; ++i) {
                c->hypo.mu[3 * i] = x_mu[i];
                c->hypo.sigma[3 * i] = p->prior_width_xy;
                c->hypo.mu[3 * i + 1] = y_mu[i];
                c->hypo.sigma[3 * i + 1] = p->prior_width_xy;
                c->hypo.mu[3 * i + 2] = p->prior_z;
                c->hypo.sigma[3 * i + 2] = p->prior_width_z;
                c->hypo.prior_type[3 * i + 2] = 1;
                c->hypo.step_size[3 * i] = p->step_size_xy;
                c->hypo.step_size[3 * i + 1] = p->step_size_xy;
                c->hypo.step_size[3 * i + 2] = p->step_size_z;
            }
            model_generate(&c->hypo, &rk->rng);
            model_alloc(&c->vs, 1);
            c->vs.mu[0] = p->prior_vs; c->vs.sigma[0] = p->prior_width_vs;
            c->vs.step_size[0] = p->step_size_vs; c->vs.x[0] = p->prior_vs;
            model_alloc(&c->qs, 1);
            c->qs.mu[0] = p->prior_qs; c->qs.sigma[0] = p->prior_width_qs;
            c->qs.step_size[0] = p->step_size_qs; c->qs.x[0] = p->prior_qs;
            c->log_likelihood = -9.e+300;
            if (j + 1 <= p->n_cool) {
                c->temp = 1.0;
            } else { /* :205-206 */
                c->temp = exp((orc_rand_u(&rk->rng) * (1.0 - EPS) + EPS) * log(p->temp_high));
            }
        }
    }
    free(x_mu);
    free(y_mu);
    return job;
}

void orc_job_destroy(orc_job *job)
{
    if (!job) return;
    for (int r = 0; r < job->p.n_procs; ++r) {
        orc_rank *rk = &job->ranks[r];
        for (int j = 0; j < job->p.n_chains; ++j) {
            orc_chain *c = &rk->chains[j];
            model_free(&c->hypo); model_free(&c->t_corr); model_free(&c->vs);
            model_free(&c->a_corr); model_free(&c->qs);
        }
        free(rk->chains);
        free(rk->lik_iter); free(rk->lik_val);
        for (int k = 0; k < rk->n_smp; ++k) {
            free(rk->smp[k].hypo); free(rk->smp[k].t_corr); free(rk->smp[k].a_corr);
        }
        free(rk->smp);
    }
    free(job->ranks);
    orc_forward_destroy(job->fwd);
    free(job->scratch_hypo);
    free(job->slog_i); free(job->slog_d);
    free(job);
}

static void record(orc_job *job, orc_rank *rk, const orc_chain *c, int i)
{
    const int S = job->n_sta, E = job->n_events;
    if (i > job->p.n_burn) { /* src/hypo_tremor_mcmc.f90:272-278 */
        if (rk->n_smp == rk->cap_smp) {
            rk->cap_smp = rk->cap_smp ? 2 * rk->cap_smp : 16;
            rk->smp = (orc_sample *)realloc(rk->smp, rk->cap_smp * sizeof(orc_sample));
        }
        orc_sample *s = &rk->smp[rk->n_smp++];
        s->iter = i;
        s->vs = c->vs.x[0];
        s->qs = c->qs.x[0];
        s->hypo = dup_d(c->hypo.x, 3 * (size_t)E);
        s->t_corr = dup_d(c->t_corr.x, S);
        s->a_corr = dup_d(c->a_corr.x, S);
    }
    if (rk->n_lik == rk->cap_lik) { /* :279 */
        rk->cap_lik = rk->cap_lik ? 2 * rk->cap_lik : 64;
        rk->lik_iter = (int32_t *)realloc(rk->lik_iter, rk->cap_lik * sizeof(int32_t));
        rk->lik_val = (double *)realloc(rk->lik_val, rk->cap_lik * sizeof(double));
    }
    rk->lik_iter[rk->n_lik] = i;
    rk->lik_val[rk->n_lik] = c->log_likelihood;
    rk->n_lik++;
}

/* one chain, one iteration: propose (src/cls_mcmc.f90:115-172) -> forward (src/hypo_tremor_mcmc.f90:245-259)
 * -> judge (src/cls_mcmc.f90:176-226) */
static void chain_step(orc_job *job, int r, int jc, int i)
{
    orc_rank *rk = &job->ranks[r];
    orc_chain *c = &rk->chains[jc];
    orc_rng *rng = &rk->rng;
    const int S = job->n_sta, E = job->n_events;

    double a_select = orc_rand_u(rng);
    int prior_ok = 1, evt_id = -999, itype, idx = 0;
    double lpr = 0.0, x_new;
    orc_model *m;
    if (a_select < job->p_vs) {
        m = &c->vs; idx = 0; itype = 1;
    } else if (a_select < job->p_vs + job->p_t_corr) {
        idx = (int)(orc_rand_u(rng) * S); m = &c->t_corr; itype = 2;
    } else if (a_select < job->p_vs + job->p_t_corr + job->p_qs) {
        m = &c->qs; idx = 0; itype = 3;
    } else if (a_select < job->p_vs + job->p_t_corr + job->p_qs + job->p_a_corr) {
        idx = (int)(orc_rand_u(rng) * S); m = &c->a_corr; itype = 4;
    } else {
        int id = (int)(orc_rand_u(rng) * E) + 1;
        int icmp = (int)(orc_rand_u(rng) * 3);
        idx = 3 * id - icmp - 1; /* 0-based index of Fortran element 3*id-icmp */
        m = &c->hypo; itype = 5 + icmp; evt_id = id;
    }
    x_new = model_perturb(m, idx, rng, &lpr, &prior_ok);

    double L_new = 0.0;
    int used_full = 0;
    double x_keep = m->x[idx];
    if (prior_ok) {
        if (evt_id > 0 && i > 1) {
            memcpy(job->scratch_hypo, c->hypo.x, 3 * (size_t)E * sizeof(double));
            job->scratch_hypo[idx] = x_new;
            L_new = orc_forward_loglik_partial(job->fwd, evt_id, c->hypo.x, c->log_likelihood,
                                               job->scratch_hypo, c->t_corr.x, c->vs.x[0], c->a_corr.x,
                                               c->qs.x[0]);
        } else {
            m->x[idx] = x_new;
            L_new = orc_forward_loglik_full(job->fwd, c->hypo.x, c->t_corr.x, c->vs.x[0], c->a_corr.x,
                                            c->qs.x[0]);
            m->x[idx] = x_keep;
            used_full = 1;
        }
    }

    /* judge */
    if (c->temp < 1.0 + EPS) c->n_propose[itype - 1]++;
    int accepted = 0;
    if (prior_ok) {
        double ratio = (L_new - c->log_likelihood) / c->temp;
        ratio = ratio + lpr;
        double rr = orc_rand_u(rng);
        if (rr >= EPS) {
            if (log(rr) <= ratio) accepted = 1;
        }
    }
    if (accepted) {
        m->x[idx] = x_new;
        c->log_likelihood = L_new;
        if (c->temp < 1.0 + EPS) c->n_accept[itype - 1]++;
    }
    c->i_iter++;

    if (job->slog_n < job->slog_cap) {
        int32_t *ir = job->slog_i + 8 * (size_t)job->slog_n;
        double *dr = job->slog_d + 4 * (size_t)job->slog_n;
        ir[0] = i; ir[1] = r; ir[2] = jc; ir[3] = itype; ir[4] = idx + 1; ir[5] = prior_ok;
        ir[6] = accepted; ir[7] = used_full;
        dr[0] = x_new; dr[1] = L_new; dr[2] = c->log_likelihood; dr[3] = c->temp;
        job->slog_n++;
    }

    /* recording: src/hypo_tremor_mcmc.f90:270-280 */
    if (c->temp < 1.0 + EPS && (i % job->p.n_interval) == 1) record(job, rk, c, i);
}

/* src/cls_parallel.f90:285-302 */
static int judge_swap(double temp1, double temp2, double l1, double l2, orc_rng *rng)
{
    double del_s = (l2 - l1) * (1.0 / temp1 - 1.0 / temp2);
    double r = orc_rand_u(rng);
    if (r >= EPS) {
        if (log(r) <= del_s) return 1;
    }
    return 0;
}

/* src/cls_parallel.f90:100-216 with :220-240; the MPI exchange is a plain memory access here */
static void swap_temperature(orc_job *job)
{
    const int n_proc = job->p.n_procs, n_chain = job->p.n_chains;
    /* with a single chain in the whole job the reference's select_pair never terminates
     * (src/cls_parallel.f90:226-230).  Defined here (and in the HIP path) as: no swap, no draws. */
    if (n_proc * n_chain < 2) return;
    orc_rng *rng0 = &job->ranks[0].rng;
    int i1 = (int)(orc_rand_u(rng0) * n_proc * n_chain);
    int i2;
    for (;;) {
        i2 = (int)(orc_rand_u(rng0) * n_proc * n_chain);
        if (i1 != i2) break;
    }
    int rank1 = i1 / n_chain, rank2 = i2 / n_chain;
    int chain1 = i1 % n_chain, chain2 = i2 % n_chain; /* 0-based */
    orc_chain *c1 = &job->ranks[rank1].chains[chain1];
    orc_chain *c2 = &job->ranks[rank2].chains[chain2];
    /* judge_swap always consumes rank1's stream (also when rank1 == rank2) */
    if (judge_swap(c1->temp, c2->temp, c1->log_likelihood, c2->log_likelihood, &job->ranks[rank1].rng)) {
        double t = c1->temp;
        c1->temp = c2->temp;
        c2->temp = t;
    }
}

void orc_job_run(orc_job *job, int n)
{
    for (int k = 0; k < n; ++k) {
        int i = ++job->i_done;
        for (int r = 0; r < job->p.n_procs; ++r)
            for (int j = 0; j < job->p.n_chains; ++j) chain_step(job, r, j, i);
        swap_temperature(job);
    }
}

int orc_job_record_words(const orc_job *job) { return 4 + 2 * job->p.n_chains; }

void orc_job_rank_begin(orc_job *job, int rank, double *rec)
{
    orc_rank *rk = &job->ranks[rank];
    const int n_proc = job->p.n_procs, n_chain = job->p.n_chains;
    const int i = rk->iter + 1;
    for (int j = 0; j < n_chain; ++j) chain_step(job, rank, j, i);
    rec[0] = rec[1] = -1.0;
    rec[2] = 0.0;
    rec[3] = (double)i;
    if (n_proc * n_chain > 1) {
        if (rank == 0) { /* select_pair, src/cls_parallel.f90:226-230 */
            int i1 = (int)(orc_rand_u(&rk->rng) * n_proc * n_chain), i2;
            for (;;) {
                i2 = (int)(orc_rand_u(&rk->rng) * n_proc * n_chain);
                if (i1 != i2) break;
            }
            rec[0] = (double)i1;
            rec[1] = (double)i2;
        }
        orc_rng peek = rk->rng; /* the rand_u() of judge_swap (:294), not consumed yet */
        rec[2] = orc_rand_u(&peek);
    }
    for (int c = 0; c < n_chain; ++c) {
        rec[4 + 2 * c] = rk->chains[c].temp;
        rec[5 + 2 * c] = rk->chains[c].log_likelihood;
    }
}

int orc_job_rank_end(orc_job *job, int rank, const double *g)
{
    orc_rank *rk = &job->ranks[rank];
    const int n_proc = job->p.n_procs, n_chain = job->p.n_chains, RW = 4 + 2 * n_chain;
    const int i = rk->iter + 1;
    if (n_proc * n_chain > 1) {
        for (int r = 0; r < n_proc; ++r)
            if ((int)g[(size_t)r * RW + 3] != i) return -6;
        const int i1 = (int)g[0], i2 = (int)g[1];
        const int rank1 = i1 / n_chain, chain1 = i1 % n_chain, rank2 = i2 / n_chain, chain2 = i2 % n_chain;
        const double T1 = g[(size_t)rank1 * RW + 4 + 2 * chain1], L1 = g[(size_t)rank1 * RW + 5 + 2 * chain1];
        const double T2 = g[(size_t)rank2 * RW + 4 + 2 * chain2], L2 = g[(size_t)rank2 * RW + 5 + 2 * chain2];
        const double r = g[(size_t)rank1 * RW + 2];
        const double del_s = (L2 - L1) * (1.0 / T1 - 1.0 / T2);
        int acc = 0;
        if (r >= EPS) { if (log(r) <= del_s) acc = 1; }
        if (acc) {
            if (rank == rank1) rk->chains[chain1].temp = T2;
            if (rank == rank2) rk->chains[chain2].temp = T1;
        }
        if (rank == rank1) (void)orc_rand_u(&rk->rng);
    }
    rk->iter = i;
    return 0;
}

int orc_job_n_lik(const orc_job *job, int rank) { return job->ranks[rank].n_lik; }

void orc_job_get_lik(const orc_job *job, int rank, int32_t *iter, double *lik)
{
    const orc_rank *rk = &job->ranks[rank];
    memcpy(iter, rk->lik_iter, rk->n_lik * sizeof(int32_t));
    memcpy(lik, rk->lik_val, rk->n_lik * sizeof(double));
}

int orc_job_n_samples(const orc_job *job, int rank) { return job->ranks[rank].n_smp; }

void orc_job_get_sample(const orc_job *job, int rank, int k, int32_t *iter, double *vs, double *qs,
                        double *hypo, double *t_corr, double *a_corr)
{
    const orc_sample *s = &job->ranks[rank].smp[k];
    *iter = s->iter; *vs = s->vs; *qs = s->qs;
    memcpy(hypo, s->hypo, 3 * (size_t)job->n_events * sizeof(double));
    memcpy(t_corr, s->t_corr, job->n_sta * sizeof(double));
    memcpy(a_corr, s->a_corr, job->n_sta * sizeof(double));
}

void orc_job_get_counts(const orc_job *job, int32_t n_propose[7], int32_t n_accept[7])
{
    for (int t = 0; t < 7; ++t) n_propose[t] = n_accept[t] = 0;
    for (int r = 0; r < job->p.n_procs; ++r)
        for (int j = 0; j < job->p.n_chains; ++j)
            for (int t = 0; t < 7; ++t) {
                n_propose[t] += job->ranks[r].chains[j].n_propose[t];
                n_accept[t] += job->ranks[r].chains[j].n_accept[t];
            }
}

void orc_job_get_chain(const orc_job *job, int rank, int chain, double *hypo, double *t_corr, double *vs,
                       double *a_corr, double *qs, double *temp, double *loglik, int32_t n_propose[7],
                       int32_t n_accept[7])
{
    const orc_chain *c = &job->ranks[rank].chains[chain];
    if (hypo) memcpy(hypo, c->hypo.x, 3 * (size_t)job->n_events * sizeof(double));
    if (t_corr) memcpy(t_corr, c->t_corr.x, job->n_sta * sizeof(double));
    if (a_corr) memcpy(a_corr, c->a_corr.x, job->n_sta * sizeof(double));
    if (vs) *vs = c->vs.x[0];
    if (qs) *qs = c->qs.x[0];
    if (temp) *temp = c->temp;
    if (loglik) *loglik = c->log_likelihood;
    if (n_propose) memcpy(n_propose, c->n_propose, sizeof(c->n_propose));
    if (n_accept) memcpy(n_accept, c->n_accept, sizeof(c->n_accept));
}

void orc_job_get_priors(const orc_job *job, int rank, int chain, double *hypo_mu, double *hypo_sigma,
                        double *hypo_step, int32_t *hypo_ptype)
{
    const orc_chain *c = &job->ranks[rank].chains[chain];
    size_t n = 3 * (size_t)job->n_events;
    memcpy(hypo_mu, c->hypo.mu, n * sizeof(double));
    memcpy(hypo_sigma, c->hypo.sigma, n * sizeof(double));
    memcpy(hypo_step, c->hypo.step_size, n * sizeof(double));
    memcpy(hypo_ptype, c->hypo.prior_type, n * sizeof(int32_t));
}

void orc_job_get_rng(const orc_job *job, int rank, uint32_t state[4])
{
    const orc_rng *s = &job->ranks[rank].rng;
    state[0] = s->x; state[1] = s->y; state[2] = s->z; state[3] = s->w;
}

void orc_job_enable_steplog(orc_job *job, int cap)
{
    free(job->slog_i); free(job->slog_d);
    job->slog_cap = cap;
    job->slog_n = 0;
    job->slog_i = cap ? (int32_t *)malloc(8 * (size_t)cap * sizeof(int32_t)) : NULL;
    job->slog_d = cap ? (double *)malloc(4 * (size_t)cap * sizeof(double)) : NULL;
}

int orc_job_steplog_n(const orc_job *job) { return job->slog_n; }

void orc_job_get_steplog(const orc_job *job, int32_t *irows, double *drows)
{
    memcpy(irows, job->slog_i, 8 * (size_t)job->slog_n * sizeof(int32_t));
    memcpy(drows, job->slog_d, 4 * (size_t)job->slog_n * sizeof(double));
}

/* ====================================================================================================
 * Step 4 (`hypo_tremor_select`), SURVEY 8f-4: per detected window, weighted linear regressions of arrival
 * time and geometrically corrected log-amplitude against the distance from the station of largest
 * amplitude.  Restates src/cls_selector.f90:50-67 (distance table), :75-132 (eval_wave_propagation) and
 * src/mod_regress.f90:5-38 (linear_regression), :40-58 (weighted_corr) in their own operation order.
 * Arrays (n_sta, n_win) column-major like the step-5 observations.  out[i] = {vs, b, t0, a0, cc_t, cc_a}
 * in the order of a regress.dat row (src/hypo_tremor_select.f90:124-125).
 * ==================================================================================================== */
static void orc_linear_regression(int n, const double *x, const double *y, const double *w, double *a, double *b)
{
    double sumx = 0.0, sumy = 0.0, sumw = 0.0, sumxy = 0.0, sumx2 = 0.0;      /* mod_regress.f90:12-24 */
    for (int i = 0; i < n; ++i) {
        sumx = sumx + x[i] * w[i];
        sumy = sumy + y[i] * w[i];
        sumw = sumw + w[i];
        sumxy = sumxy + x[i] * y[i] * w[i];
        sumx2 = sumx2 + x[i] * x[i] * w[i];
    }
    const double d = sumw * sumx2 - sumx * sumx;                              /* :26-29 */
    *a = (sumw * sumxy - sumx * sumy) / d;
    *b = (sumx2 * sumy - sumx * sumxy) / d;
}

static double orc_weighted_corr(int n, const double *x, const double *y, const double *w)
{
    double sum_w = 0.0, sx = 0.0, sy = 0.0;                                   /* mod_regress.f90:47-49 */
    for (int i = 0; i < n; ++i) sum_w += w[i];
    for (int i = 0; i < n; ++i) sx += x[i] * w[i];
    for (int i = 0; i < n; ++i) sy += y[i] * w[i];
    const double mean_x = sx / sum_w, mean_y = sy / sum_w;
    double s_xx = 0.0, s_yy = 0.0, s_xy = 0.0;                                /* :51-53: sums WITHOUT the weights (sic) */
    for (int i = 0; i < n; ++i) s_xx += (x[i] - mean_x) * (x[i] - mean_x);
    for (int i = 0; i < n; ++i) s_yy += (y[i] - mean_y) * (y[i] - mean_y);
    for (int i = 0; i < n; ++i) s_xy += (x[i] - mean_x) * (y[i] - mean_y);
    return s_xy / sqrt(s_xx * s_yy);                                          /* :55 */
}

void orc_select_regress(int n_sta, int n_win, const double *sta_x, const double *sta_y, const double *sta_z,
                        double z_guess, const double *t, const double *t_err, const double *a,
                        const double *a_err, double *out)
{
    double *d = (double *)malloc(sizeof(double) * n_sta), *ac = (double *)malloc(sizeof(double) * n_sta);
    double *w = (double *)malloc(sizeof(double) * n_sta);
    for (int i = 0; i < n_win; ++i) {
        const double *ti = t + (size_t)i * n_sta, *te = t_err + (size_t)i * n_sta;
        const double *ai = a + (size_t)i * n_sta, *ae = a_err + (size_t)i * n_sta;
        int near = 0;                                                         /* maxloc(a): first maximum, cls_selector.f90:99 */
        for (int j = 1; j < n_sta; ++j) if (ai[j] > ai[near]) near = j;
        for (int j = 0; j < n_sta; ++j) {                                     /* :62-64: source at the nearest station's x, y and depth z_guess */
            const double dx = sta_x[j] - sta_x[near], dy = sta_y[j] - sta_y[near], dz = sta_z[j] - z_guess;
            d[j] = sqrt(dx * dx + dy * dy + dz * dz);
            ac[j] = ai[j] + log(d[j]);                                        /* :102-103 geometrical spreading */
        }
        double slope, intercept;
        for (int j = 0; j < n_sta; ++j) w[j] = 1.0 / (te[j] * te[j]);         /* :114 */
        orc_linear_regression(n_sta, d, ti, w, &slope, &intercept);
        const double vs = 1.0 / slope, t0 = intercept;                        /* :117-118 */
        const double cc_t = orc_weighted_corr(n_sta, d, ti, w);               /* :127-128 */
        for (int j = 0; j < n_sta; ++j) w[j] = 1.0 / (ae[j] * ae[j]);         /* :120 */
        orc_linear_regression(n_sta, d, ac, w, &slope, &intercept);
        const double b = -1.0 * slope, a0 = intercept;                        /* :123-124 */
        const double cc_a = orc_weighted_corr(n_sta, d, ac, w);               /* :129-130 */
        double *o = out + (size_t)i * 6;
        o[0] = vs; o[1] = b; o[2] = t0; o[3] = a0; o[4] = cc_t; o[5] = cc_a;
    }
    free(d); free(ac); free(w);
}
