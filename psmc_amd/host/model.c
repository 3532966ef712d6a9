/* model.c -- PSMC population parameters -> HMM parameters.
 *
 * The mathematics is psmc.tex:407-465 of the reference; the evaluation order of
 * every expression follows core.c so the doubles come out bit-identical
 * (psmc_update_intv core.c:6-19, psmc_update_hmm core.c:61-133, psmc_avg_t
 * core.c:135-162, psmc_cap_matrix aux.c:115-127).
 * params = [theta0, rho0, max_t, lambda_0..lambda_{n_free-1}, (dt)].
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "psmc_host.h"

psmc_model *psmc_model_new(const psmc_pattern *pat, const char *pattern_text, double alpha, int has_dt)
{
	psmc_model *m = (psmc_model *)calloc(1, sizeof(psmc_model));
	const int N = pat->n_states;
	m->pat.n_states = N; m->pat.n_free = pat->n_free;
	m->pat.group = (int *)malloc(sizeof(int) * (size_t)N);
	memcpy(m->pat.group, pat->group, sizeof(int) * (size_t)N);
	m->pattern_text = strdup(pattern_text);
	m->alpha = alpha; m->has_dt = has_dt;
	m->n_params = pat->n_free + PSMC_N_FIXED + (has_dt ? 1 : 0);
	m->params = (double *)calloc((size_t)m->n_params + 1, sizeof(double));
	m->t = (double *)calloc((size_t)N + 1, sizeof(double));
	m->sigma = (double *)calloc((size_t)N, sizeof(double));
	m->post_sigma = (double *)calloc((size_t)N, sizeof(double));
	m->a = (double *)calloc((size_t)N * N, sizeof(double));
	m->e = (double *)calloc((size_t)3 * N, sizeof(double));
	m->a0 = (double *)calloc((size_t)N, sizeof(double));
	for (int k = 0; k < N; ++k) m->e[2 * N + k] = 1.0; /* missing data emits with probability 1 (khmm.c:21) */
	return m;
}

void psmc_model_free(psmc_model *m)
{
	if (!m) return;
	psmc_pattern_free(&m->pat);
	free(m->pattern_text); free(m->fixed_t); free(m->params); free(m->t); free(m->sigma); free(m->post_sigma);
	free(m->a); free(m->e); free(m->a0);
	free(m);
}

/* boundaries t_0..t_n (n = N-1 = psmc's n) and t_{n+1} = "infinity" */
static void time_boundaries(const psmc_model *m, double max_t, double *t)
{
	const int n = m->pat.n_states - 1;
	if (m->fixed_t) {
		memcpy(t, m->fixed_t, sizeof(double) * (size_t)(n + 1));
	} else {
		const double beta = log(1.0 + max_t / m->alpha) / n; /* core.c:11 */
		for (int k = 0; k < n; ++k) t[k] = m->alpha * (exp(beta * k) - 1);
		t[n] = max_t;
	}
	t[n + 1] = PSMC_T_INFINITY;
}

/* ---- The coalescent intervals of one parameter set: everything psmc.tex:407-465 derives per interval BEFORE a matrix exists.
 *
 * Own layout: the transition matrix of PSMC is two rank-one triangles and a diagonal,
 *     a[k][l] = stay_k * (below_k * g_l)      l < k
 *             = stay_k * diag_k + (1 - stay_k) l = k
 *             = stay_k * (above_k * drop_l)   l > k
 * and this file keeps those FACTORS (the fast M-step and the structured device kernels work on them directly); the dense matrix of
 * psmc_model_update is their expansion.  What is NOT free is the arithmetic: a byte-identical .psmc needs every double to round
 * where the reference's does (core.c:61-133, psmc_update_hmm), so each quantity below is evaluated with core.c's association --
 * noted per line -- including the ones that look redundant (the running sum `start`, which is not t_k bit for bit).
 * lh3/psmc is Copyright (c) 2007-2009 Genome Research Ltd, 2009-2015 Broad Institute, MIT License; the full notice is in /NOTICE. */
typedef struct {
	int N;
	double *buf;
	double *lam;    /* lambda_k: relative population size in interval k */
	double *width;  /* tau_k = t_{k+1} - t_k */
	double *surv;   /* alpha_k: no coalescence before t_k; surv[N] = 0 */
	double *drop;   /* alpha_k - alpha_{k+1} */
	double *cum;    /* beta_k */
	double *g;      /* the l-dependent factor of q_{kl}, l < k ("q_aux", core.c:91-92); N - 1 entries */
	double *start;  /* widths summed in order up to interval k */
	double *mass;   /* C_pi * pi_k */
	double *pi, *sigma, *stay, *below, *above, *diag, *mean_t;
	double C_pi, C_sigma, dt;
} intervals;

enum { IV_ARRAYS = 15 };

static int intervals_alloc(intervals *v, int N)
{
	v->N = N;
	v->buf = (double *)calloc((size_t)IV_ARRAYS * (size_t)(N + 1), sizeof(double));
	if (!v->buf) { /* 15 short arrays: out of memory here is out of memory for the run, and going on with the previous parameters in place would be silent garbage */
		fprintf(stderr, "psmc: out of memory (model intervals, %d states)\n", N);
		abort();
	}
	double **slot[IV_ARRAYS] = {&v->lam, &v->width, &v->surv, &v->drop, &v->cum, &v->g, &v->start, &v->mass, &v->pi, &v->sigma, &v->stay,
	                            &v->below, &v->above, &v->diag, &v->mean_t};
	for (int i = 0; i < IV_ARRAYS; ++i) *slot[i] = v->buf + (size_t)i * (size_t)(N + 1);
	return 0;
}

/* the mean coalescence time inside an interval given that it happens there, with the reference's fallback when the closed form leaves
 * the interval or is not a number (core.c:111-114) */
static double interval_mean_time(double pi_k, double C_sigma, double sigma_k, double rho, double start, double width,
                                 double lam_k, double surv_k, double surv_k1)
{
	double at = -log(1.0 - pi_k / (C_sigma * sigma_k)) / rho;
	if (isnan(at) || at < start || at > start + width)
		at = start + (lam_k - width * surv_k1 / (surv_k - surv_k1));
	return at;
}

/* survival up to every boundary from the widths and sizes: alpha_k = alpha_{k-1} * exp(-tau_{k-1} / lambda_{k-1}) (core.c:84-86) */
static void survival_curve(int N, const double *width, const double *lam, double *surv)
{
	surv[0] = 1.0;
	for (int k = 1; k < N; ++k) surv[k] = surv[k - 1] * exp(-width[k - 1] / lam[k - 1]);
	surv[N] = 0.0;
}

/* fills v from m->params (and m->t from max_t); `full` = 0 stops after the survival curve (psmc_model_avg_t needs no more) */
static void intervals_compute(psmc_model *m, intervals *v, int refresh_t, int full)
{
	const int N = v->N, last = N - 1;
	const double rho = m->params[1];
	for (int k = 0; k < N; ++k) v->lam[k] = m->params[m->pat.group[k] + PSMC_N_FIXED];
	if (refresh_t) time_boundaries(m, m->params[2], m->t);
	v->dt = 0.0;
	if (m->has_dt) { v->dt = m->params[m->n_params - 1]; if (v->dt < 0) v->dt = 0; }
	for (int k = 0; k < N; ++k) v->width[k] = m->t[k + 1] - m->t[k];
	survival_curve(N, v->width, v->lam, v->surv);
	for (int k = 0; k < N; ++k) v->drop[k] = v->surv[k] - v->surv[k + 1];
	{ double run = 0.0; for (int k = 0; k < N; ++k) { v->start[k] = run; run += v->width[k]; } } /* core.c:131: sum_t += tau[k] */
	if (!full) return;
	v->cum[0] = 0.0; /* beta, core.c:88-89 */
	for (int k = 1; k < N; ++k) v->cum[k] = v->cum[k - 1] + v->lam[k - 1] * (1.0 / v->surv[k] - 1.0 / v->surv[k - 1]);
	for (int l = 0; l < last; ++l) v->g[l] = v->drop[l] * (v->cum[l] - v->lam[l] / v->surv[l]) + v->width[l];
	v->C_pi = 0.0; /* core.c:94-96 */
	for (int l = 0; l < N; ++l) v->C_pi += v->lam[l] * v->drop[l];
	v->C_sigma = 1.0 / (v->C_pi * rho) + 0.5;
	for (int k = 0; k < N; ++k) {
		const double d = v->drop[k], lam = v->lam[k];
		v->mass[k] = d * (v->start[k] + lam) - v->surv[k + 1] * v->width[k];                    /* core.c:102 */
		v->pi[k] = v->mass[k] / v->C_pi;
		v->sigma[k] = (d / (v->C_pi * rho) + v->pi[k] / 2.0) / v->C_sigma;                      /* core.c:104 */
		v->mean_t[k] = interval_mean_time(v->pi[k], v->C_sigma, v->sigma[k], rho, v->start[k], v->width[k], lam, v->surv[k], v->surv[k + 1]);
		v->below[k] = d / v->mass[k];                                                          /* core.c:116 */
		v->diag[k] = (d * d * (v->cum[k] - lam / v->surv[k]) + 2 * lam * d - 2 * v->surv[k + 1] * v->width[k]) / v->mass[k]; /* core.c:118 */
		v->above[k] = k < last ? v->g[k] / v->mass[k] : 0.0;                                   /* core.c:120 */
		v->stay[k] = v->pi[k] / (v->C_sigma * v->sigma[k]);                                    /* core.c:124 */
	}
}

/* model parameters -> HMM (a, e, a0): the factors above, expanded */
void psmc_model_update(psmc_model *m)
{
	const int N = m->pat.n_states;
	const double theta = m->params[0];
	intervals v;
	if (intervals_alloc(&v, N)) return;
	intervals_compute(m, &v, 1, 1);
	m->C_pi = v.C_pi; m->C_sigma = v.C_sigma;
	for (int k = 0; k < N; ++k) {
		double *row = m->a + (size_t)k * N;
		const double stay = v.stay[k];
		for (int l = 0; l < k; ++l) row[l] = stay * (v.below[k] * v.g[l]);        /* p_kl = stay * q_kl, q_kl rounded first (core.c:117,125) */
		row[k] = stay * v.diag[k] + (1.0 - stay);                                   /* core.c:126 */
		for (int l = k + 1; l < N; ++l) row[l] = stay * (v.drop[l] * v.above[k]);  /* core.c:121,125 */
		m->sigma[k] = v.sigma[k];
		m->a0[k] = v.sigma[k];
		m->e[k] = exp(-theta * (v.mean_t[k] + v.dt));                               /* core.c:128-129 */
		m->e[N + k] = 1.0 - m->e[k];
	}
	free(v.buf);
}

/* The factors of psmc_model_update's matrix instead of the matrix (fast M-step):
 *   a[k][l] = FL_k * qa_l (l < k),  FU_k * c_l (l > k),  D_k (l == k);  e[0][k] = exp(le0_k), e[1][k] = 1 - e[0][k]
 * out = log FL | log FU | log D | log qa | log c | le0 | log e1, N each (entries that do not exist -- FL_0, FU_n, qa_n, c_0 -- are 0).
 * Returns 0 when a factor is not positive (the reference's Q is then -HMM_INF, khmm.c:369-377), else 1.  Does not touch a/e/a0/sigma. */
int psmc_model_logfactors(psmc_model *m, double *out)
{
	const int N = m->pat.n_states, last = N - 1;
	double *lFL = out, *lFU = out + N, *lD = out + 2 * N, *lqa = out + 3 * N, *lc = out + 4 * N, *le0 = out + 5 * N, *le1 = out + 6 * N;
	const double theta = m->params[0];
	intervals v;
	int ok = 1;
	memset(out, 0, sizeof(double) * (size_t)(7 * N));
	if (intervals_alloc(&v, N)) return 0;
	intervals_compute(m, &v, 1, 1);
	for (int k = 0; k < N && ok; ++k) {
		const double stay = v.stay[k];
		const double D = stay * v.diag[k] + (1.0 - stay);
		if (!(D > 0.0)) ok = 0; else lD[k] = log(D);
		if (k > 0) { const double FL = stay * v.below[k]; if (!(FL > 0.0)) ok = 0; else lFL[k] = log(FL); }
		if (k < last) {
			const double FU = stay * v.above[k];
			if (!(FU > 0.0) || !(v.g[k] > 0.0)) ok = 0; else { lFU[k] = log(FU); lqa[k] = log(v.g[k]); }
		}
		if (k > 0) { if (!(v.drop[k] > 0.0)) ok = 0; else lc[k] = log(v.drop[k]); }
		const double x = -theta * (v.mean_t[k] + v.dt), e1 = 1.0 - exp(x);
		le0[k] = x;
		if (!(e1 > 0.0) || !(exp(x) > 0.0)) ok = 0; else le1[k] = log(e1);
	}
	free(v.buf);
	return ok;
}

/* mean coalescence time per interval for the output's time axis (core.c:135-162): from the boundaries, normalisers and sigma the last
 * psmc_model_update left in *m */
void psmc_model_avg_t(const psmc_model *m, double *avg_t)
{
	const int N = m->pat.n_states;
	const double rho = m->params[1];
	intervals v;
	if (intervals_alloc(&v, N)) return;
	intervals_compute((psmc_model *)m, &v, 0, 0); /* (no refresh: *m is only read) */
	for (int k = 0; k < N; ++k) {
		const double pi_k = (v.drop[k] * (v.start[k] + v.lam[k]) - v.surv[k + 1] * v.width[k]) / m->C_pi;
		avg_t[k] = interval_mean_time(pi_k, m->C_sigma, m->sigma[k], rho, v.start[k], v.width[k], v.lam[k], v.surv[k], v.surv[k + 1]);
		avg_t[k] += v.dt;
	}
	free(v.buf);
}

/* -C: fold all transitions into states >= k0 onto state k0 (aux.c:115-127) */
void psmc_model_cap(psmc_model *m, int k0)
{
	const int N = m->pat.n_states;
	for (int k = 0; k < N; ++k) {
		double s = 0.0, *row = m->a + (size_t)k * N;
		for (int l = k0; l < N; ++l) { s += row[l]; row[l] = 0.0; }
		row[k0] = s;
	}
}
