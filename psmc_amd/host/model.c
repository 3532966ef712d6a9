/* model.c -- PSMC population parameters -> HMM parameters.
 *
 * The mathematics is psmc.tex:407-465 of the reference; the evaluation order of
 * every expression follows core.c so the doubles come out bit-identical
 * (psmc_update_intv core.c:6-19, psmc_update_hmm core.c:61-133, psmc_avg_t
 * core.c:135-162, psmc_cap_matrix aux.c:115-127).
 * params = [theta0, rho0, max_t, lambda_0..lambda_{n_free-1}, (dt)].
 */
#include <math.h>
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

static double interval_mean_time(double pik, double C_sigma, double sigma_k, double rho, double sum_t, double tau_k,
                                 double lak, double alpha_k, double alpha_k1)
{	/* core.c:112-114: average coalescent time inside interval k, with the reference's fallback */
	double avg = -log(1.0 - pik / (C_sigma * sigma_k)) / rho;
	if (isnan(avg) || avg < sum_t || avg > sum_t + tau_k)
		avg = sum_t + (lak - tau_k * alpha_k1 / (alpha_k - alpha_k1));
	return avg;
}

/* model parameters -> HMM (a, e, a0): RESTATED from lh3/psmc core.c:61-133 (psmc_update_hmm), expression shapes kept on
 * purpose -- every product and quotient rounds where the reference's does, which the byte-identical .psmc contract
 * needs.  lh3/psmc is Copyright (c) 2007-2009 Genome Research Ltd, 2009-2015 Broad Institute, MIT License; the full
 * notice is in /NOTICE.  Own here: the flat layout, the O(N) log factors (psmc_model_logfactors) and everything around. */
void psmc_model_update(psmc_model *m)
{
	const int N = m->pat.n_states, n = N - 1;
	double *lambda = (double *)malloc(sizeof(double) * (size_t)(N));
	double *alpha = (double *)malloc(sizeof(double) * (size_t)(N + 1));
	double *beta = (double *)malloc(sizeof(double) * (size_t)(N));
	double *q_aux = (double *)malloc(sizeof(double) * (size_t)(N));
	double *q = (double *)malloc(sizeof(double) * (size_t)(N));
	double *tau = (double *)malloc(sizeof(double) * (size_t)(N));
	const double theta = m->params[0], rho = m->params[1], max_t = m->params[2];
	double dt = 0.0;
	for (int k = 0; k <= n; ++k) lambda[k] = m->params[m->pat.group[k] + PSMC_N_FIXED];
	time_boundaries(m, max_t, m->t);
	if (m->has_dt) { dt = m->params[m->n_params - 1]; if (dt < 0) dt = 0; }
	const double *t = m->t;
	for (int k = 0; k <= n; ++k) tau[k] = t[k + 1] - t[k];
	alpha[0] = 1.0;
	for (int k = 1; k <= n; ++k) alpha[k] = alpha[k - 1] * exp(-tau[k - 1] / lambda[k - 1]);
	alpha[n + 1] = 0.0;
	beta[0] = 0.0;
	for (int k = 1; k <= n; ++k) beta[k] = beta[k - 1] + lambda[k - 1] * (1.0 / alpha[k] - 1.0 / alpha[k - 1]);
	for (int l = 0; l < n; ++l) q_aux[l] = (alpha[l] - alpha[l + 1]) * (beta[l] - lambda[l] / alpha[l]) + tau[l];
	m->C_pi = 0.0;
	for (int l = 0; l <= n; ++l) m->C_pi += lambda[l] * (alpha[l] - alpha[l + 1]);
	m->C_sigma = 1.0 / (m->C_pi * rho) + 0.5;
	double sum_t = 0.0;
	for (int k = 0; k <= n; ++k) {
		const double ak1 = alpha[k] - alpha[k + 1], lak = lambda[k];
		const double cpik = ak1 * (sum_t + lak) - alpha[k + 1] * tau[k];
		const double pik = cpik / m->C_pi;
		m->sigma[k] = (ak1 / (m->C_pi * rho) + pik / 2.0) / m->C_sigma;
		const double avg_t = interval_mean_time(pik, m->C_sigma, m->sigma[k], rho, sum_t, tau[k], lak, alpha[k], alpha[k + 1]);
		/* q_{kl}: rank one below the diagonal, rank one above it (core.c:116-122) */
		double tmp = ak1 / cpik;
		int l;
		for (l = 0; l < k; ++l) q[l] = tmp * q_aux[l];
		q[l++] = (ak1 * ak1 * (beta[k] - lak / alpha[k]) + 2 * lak * ak1 - 2 * alpha[k + 1] * tau[k]) / cpik;
		if (k < n) {
			tmp = q_aux[k] / cpik;
			for (; l <= n; ++l) q[l] = (alpha[l] - alpha[l + 1]) * tmp;
		}
		/* p_{kl} and e_k(b) (core.c:124-130) */
		tmp = pik / (m->C_sigma * m->sigma[k]);
		double *row = m->a + (size_t)k * N;
		for (l = 0; l <= n; ++l) row[l] = tmp * q[l];
		row[k] = tmp * q[k] + (1.0 - tmp);
		m->a0[k] = m->sigma[k];
		m->e[k] = exp(-theta * (avg_t + dt));
		m->e[N + k] = 1.0 - m->e[k];
		sum_t += tau[k];
	}
	free(lambda); free(alpha); free(beta); free(q_aux); free(q); free(tau);
}

/* The factors of psmc_model_update's matrix instead of the matrix (fast M-step):
 *   a[k][l] = FL_k * qa_l (l < k),  FU_k * c_l (l > k),  D_k (l == k);  e[0][k] = exp(le0_k), e[1][k] = 1 - e[0][k]
 * with the same scalar recurrences as above.  out = log FL | log FU | log D | log qa | log c | le0 | log e1,
 * N each (entries that do not exist -- FL_0, FU_n, qa_n, c_0 -- are 0).  Returns 0 when a factor is not
 * positive (the reference's Q is then -HMM_INF, khmm.c:369-377), else 1.  Does not touch a/e/a0/sigma. */
int psmc_model_logfactors(psmc_model *m, double *out)
{
	const int N = m->pat.n_states, n = N - 1;
	double *w = (double *)malloc(sizeof(double) * (size_t)(5 * N + 1));
	double *lambda = w, *alpha = w + N, *beta = w + 2 * N + 1, *q_aux = w + 3 * N + 1, *tau = w + 4 * N + 1;
	double *lFL = out, *lFU = out + N, *lD = out + 2 * N, *lqa = out + 3 * N, *lc = out + 4 * N, *le0 = out + 5 * N, *le1 = out + 6 * N;
	const double theta = m->params[0], rho = m->params[1], max_t = m->params[2];
	double dt = 0.0;
	int ok = 1;
	memset(out, 0, sizeof(double) * (size_t)(7 * N));
	for (int k = 0; k <= n; ++k) lambda[k] = m->params[m->pat.group[k] + PSMC_N_FIXED];
	time_boundaries(m, max_t, m->t);
	if (m->has_dt) { dt = m->params[m->n_params - 1]; if (dt < 0) dt = 0; }
	const double *t = m->t;
	for (int k = 0; k <= n; ++k) tau[k] = t[k + 1] - t[k];
	alpha[0] = 1.0;
	for (int k = 1; k <= n; ++k) alpha[k] = alpha[k - 1] * exp(-tau[k - 1] / lambda[k - 1]);
	alpha[n + 1] = 0.0;
	beta[0] = 0.0;
	for (int k = 1; k <= n; ++k) beta[k] = beta[k - 1] + lambda[k - 1] * (1.0 / alpha[k] - 1.0 / alpha[k - 1]);
	for (int l = 0; l < n; ++l) q_aux[l] = (alpha[l] - alpha[l + 1]) * (beta[l] - lambda[l] / alpha[l]) + tau[l];
	double C_pi = 0.0;
	for (int l = 0; l <= n; ++l) C_pi += lambda[l] * (alpha[l] - alpha[l + 1]);
	const double C_sigma = 1.0 / (C_pi * rho) + 0.5;
	double sum_t = 0.0;
	for (int k = 0; k <= n && ok; ++k) {
		const double ak1 = alpha[k] - alpha[k + 1], lak = lambda[k];
		const double cpik = ak1 * (sum_t + lak) - alpha[k + 1] * tau[k];
		const double pik = cpik / C_pi;
		const double sigma_k = (ak1 / (C_pi * rho) + pik / 2.0) / C_sigma;
		const double avg_t = interval_mean_time(pik, C_sigma, sigma_k, rho, sum_t, tau[k], lak, alpha[k], alpha[k + 1]);
		const double tmp = pik / (C_sigma * sigma_k);
		const double qkk = (ak1 * ak1 * (beta[k] - lak / alpha[k]) + 2 * lak * ak1 - 2 * alpha[k + 1] * tau[k]) / cpik;
		const double D = tmp * qkk + (1.0 - tmp);
		if (!(D > 0.0)) ok = 0; else lD[k] = log(D);
		if (k > 0) { const double FL = tmp * (ak1 / cpik); if (!(FL > 0.0)) ok = 0; else lFL[k] = log(FL); }
		if (k < n) {
			const double FU = tmp * (q_aux[k] / cpik);
			if (!(FU > 0.0) || !(q_aux[k] > 0.0)) ok = 0; else { lFU[k] = log(FU); lqa[k] = log(q_aux[k]); }
		}
		if (k > 0) { if (!(ak1 > 0.0)) ok = 0; else lc[k] = log(ak1); }
		const double x = -theta * (avg_t + dt), e1 = 1.0 - exp(x);
		le0[k] = x;
		if (!(e1 > 0.0) || !(exp(x) > 0.0)) ok = 0; else le1[k] = log(e1);
		sum_t += tau[k];
	}
	free(w);
	return ok;
}

void psmc_model_avg_t(const psmc_model *m, double *avg_t)
{
	const int N = m->pat.n_states, n = N - 1;
	double *lambda = (double *)malloc(sizeof(double) * (size_t)N);
	double *alpha = (double *)malloc(sizeof(double) * (size_t)(N + 1));
	double *tau = (double *)malloc(sizeof(double) * (size_t)N);
	const double rho = m->params[1];
	double dt = 0.0, sum_t = 0.0;
	for (int k = 0; k <= n; ++k) lambda[k] = m->params[m->pat.group[k] + PSMC_N_FIXED];
	if (m->has_dt) { dt = m->params[m->n_params - 1]; if (dt < 0) dt = 0; }
	for (int k = 0; k <= n; ++k) tau[k] = m->t[k + 1] - m->t[k];
	alpha[0] = 1.0;
	for (int k = 1; k <= n; ++k) alpha[k] = alpha[k - 1] * exp(-tau[k - 1] / lambda[k - 1]);
	alpha[n + 1] = 0.0;
	for (int k = 0; k <= n; ++k) {
		const double ak1 = alpha[k] - alpha[k + 1], lak = lambda[k];
		const double pik = (ak1 * (sum_t + lak) - alpha[k + 1] * tau[k]) / m->C_pi;
		avg_t[k] = interval_mean_time(pik, m->C_sigma, m->sigma[k], rho, sum_t, tau[k], lak, alpha[k], alpha[k + 1]);
		avg_t[k] += dt;
		sum_t += tau[k];
	}
	free(lambda); free(alpha); free(tau);
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
