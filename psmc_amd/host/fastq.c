/* fastq.c -- psmc_model_logfactors restructured for SIMD (fast M-step only; not bit-identical to model.c).
 *
 * The scalar version (model.c) spends its time in ~9N libm calls per objective evaluation.  Here the
 * recurrences (alpha, beta, sum_t) are scalar scans and everything else is a loop over independent k, which
 * gcc vectorises with glibc's libmvec (4 logs / exps per call with AVX2).  This file is compiled with
 * -O3 -mavx2 -mfma -ffast-math -fopenmp-simd (see Makefile and the note on finite-math-only below); results agree with the
 * scalar version to ~1e-15 relative.  Same formulas as psmc_update_hmm, lh3/psmc core.c:61-133.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "psmc_host.h"

#define NMAX 136 /* 128 states + slack */

/* This file is compiled with -ffast-math (glibc only offers the libmvec variants of log / exp under __FAST_MATH__), and
 * that implies -ffinite-math-only: the compiler may assume NaN and infinity never occur and fold floating-point tests
 * that only they can fail.  Trial points of the direct search do produce them (lambda -> 0, 0/0 in sigma_k ...), so
 * every accept / reject decision below is taken on the BIT PATTERN, which no floating-point assumption can remove. */
static inline uint64_t bits_of(double x) { uint64_t u; memcpy(&u, &x, 8); return u; }
static inline int is_finite_bits(double x) { return ((bits_of(x) >> 52) & 0x7ff) != 0x7ff; }
static inline int pos_finite_bits(double x) { const uint64_t u = bits_of(x); return (u >> 63) == 0 && u != 0 && (u >> 52) != 0x7ff; }

int psmc_model_logfactors_simd(psmc_model *m, double *out)
{
	const int N = m->pat.n_states, n = N - 1;
	if (N > 128) return psmc_model_logfactors(m, out);
	double lambda[NMAX], tau[NMAX], ex[NMAX], alpha[NMAX], inva[NMAX], beta[NMAX], q_aux[NMAX], ak1[NMAX], sum_t[NMAX];
	double D[NMAX], FL[NMAX], FU[NMAX], QA[NMAX], CC[NMAX], X[NMAX], E1[NMAX], arg[NMAX], alt[NMAX], okv[NMAX];
	double *lFL = out, *lFU = out + N, *lD = out + 2 * N, *lqa = out + 3 * N, *lc = out + 4 * N, *le0 = out + 5 * N, *le1 = out + 6 * N;
	const double theta = m->params[0], rho = m->params[1], max_t = m->params[2];
	double dt = 0.0;
	if (m->has_dt) { dt = m->params[m->n_params - 1]; if (dt < 0) dt = 0; }
	for (int k = 0; k <= n; ++k) lambda[k] = m->params[m->pat.group[k] + PSMC_N_FIXED];
	/* time boundaries (psmc_update_intv, core.c:6-19) */
	double *t = m->t;
	if (m->fixed_t) memcpy(t, m->fixed_t, sizeof(double) * (size_t)(n + 1));
	else {
		const double b = log(1.0 + max_t / m->alpha) / n;
#pragma omp simd
		for (int k = 0; k < n; ++k) t[k] = m->alpha * (exp(b * k) - 1);
		t[n] = max_t;
	}
	t[n + 1] = PSMC_T_INFINITY;
#pragma omp simd
	for (int k = 0; k <= n; ++k) { tau[k] = t[k + 1] - t[k]; ex[k] = exp(-tau[k] / lambda[k]); }
	alpha[0] = 1.0; sum_t[0] = 0.0;
	for (int k = 1; k <= n; ++k) { alpha[k] = alpha[k - 1] * ex[k - 1]; sum_t[k] = sum_t[k - 1] + tau[k - 1]; }
	alpha[n + 1] = 0.0;
#pragma omp simd
	for (int k = 0; k <= n; ++k) { inva[k] = 1.0 / alpha[k]; ak1[k] = alpha[k] - alpha[k + 1]; }
	beta[0] = 0.0;
	for (int k = 1; k <= n; ++k) beta[k] = beta[k - 1] + lambda[k - 1] * (inva[k] - inva[k - 1]);
	double C_pi = 0.0;
#pragma omp simd reduction(+ : C_pi)
	for (int l = 0; l <= n; ++l) { q_aux[l] = ak1[l] * (beta[l] - lambda[l] * inva[l]) + tau[l]; C_pi += lambda[l] * ak1[l]; }
	const double C_sigma = 1.0 / (C_pi * rho) + 0.5, irho = 1.0 / rho;
#pragma omp simd
	for (int k = 0; k <= n; ++k) {
		const double lak = lambda[k];
		const double cpik = ak1[k] * (sum_t[k] + lak) - alpha[k + 1] * tau[k];
		const double pik = cpik / C_pi;
		const double sigma_k = (ak1[k] / (C_pi * rho) + pik / 2.0) / C_sigma;
		const double tmp = pik / (C_sigma * sigma_k);
		arg[k] = 1.0 - tmp;
		alt[k] = sum_t[k] + (lak - tau[k] * alpha[k + 1] / ak1[k]); /* the reference's fallback (core.c:113-114) */
		const double qkk = (ak1[k] * ak1[k] * (beta[k] - lak * inva[k]) + 2 * lak * ak1[k] - 2 * alpha[k + 1] * tau[k]) / cpik;
		D[k] = tmp * qkk + (1.0 - tmp);
		FL[k] = k > 0 ? tmp * (ak1[k] / cpik) : 1.0;
		FU[k] = k < n ? tmp * (q_aux[k] / cpik) : 1.0;
		QA[k] = k < n ? q_aux[k] : 1.0;
		CC[k] = k > 0 ? ak1[k] : 1.0;
	}
#pragma omp simd
	for (int k = 0; k <= n; ++k) {
		const int arg_ok = pos_finite_bits(arg[k]);
		double avg = -log(arg_ok ? arg[k] : 1.0) * irho;      /* NaN in the reference <=> arg <= 0 */
		const int inside = arg_ok && is_finite_bits(avg) && avg >= sum_t[k] && avg <= sum_t[k] + tau[k];
		if (!inside) avg = alt[k];
		X[k] = -theta * (avg + dt);
	}
#pragma omp simd
	for (int k = 0; k <= n; ++k) E1[k] = 1.0 - exp(X[k]);
#pragma omp simd
	for (int k = 0; k <= n; ++k)
		okv[k] = (pos_finite_bits(D[k]) && pos_finite_bits(FL[k]) && pos_finite_bits(FU[k]) && pos_finite_bits(QA[k]) &&
		          pos_finite_bits(CC[k]) && pos_finite_bits(E1[k]) && is_finite_bits(X[k]) && X[k] > -700.0) ? 0.0 : 1.0;
	int bad = 0;
	for (int k = 0; k <= n; ++k) bad += okv[k] != 0.0;
	if (bad) { memset(out, 0, sizeof(double) * (size_t)(7 * N)); return 0; }
#pragma omp simd
	for (int k = 0; k <= n; ++k) {
		lD[k] = log(D[k]); lFL[k] = log(FL[k]); lFU[k] = log(FU[k]); lqa[k] = log(QA[k]); lc[k] = log(CC[k]);
		le0[k] = X[k]; le1[k] = log(E1[k]);
	}
	return 1;
}
