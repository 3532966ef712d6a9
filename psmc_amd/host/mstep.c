/* mstep.c -- the M-step of one EM round: EM-Q function and its maximisation by
 * a Hooke-Jeeves direct search over (theta0, rho0, max_t, lambdas[, dt]).
 *
 * Bit-faithful to the reference because the search is driven by `<` between
 * nearly equal Q values: hmm_Q0 khmm.c:326-342, hmm_Q
 * khmm.c:363-382, kmin_hj kmin.c:48-107 (r=0.5, eps=1e-7, 50000 calls,
 * kmin.h:4-6), objective em.c:15-25, psmc_em em.c:27-78.  Note the reference's
 * quirk, kept here: the objective writes |x| into the model on EVERY call and
 * the search's best point is never copied back, so the parameters carried to
 * the next round are those of the LAST evaluated trial point (em.c:21-22,67).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>
#include "psmc_host.h"

static double now_ms(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e3 + t.tv_nsec * 1e-6; }

#define HJ_RADIUS 0.5
#define HJ_EPS 1e-7
#define HJ_MAXCALL 50000
#define Q_MINUS_INF (-1e300) /* -HMM_INF, khmm.h:29 */

double psmc_Q0(int n, const double *A, const double *E)
{
	double sum = 0.0;
	for (int k = 0; k < n; ++k) {
		double tot = 0.0;
		for (int b = 0; b < 2; ++b) tot += E[b * n + k];
		for (int b = 0; b < 2; ++b) sum += E[b * n + k] * log(E[b * n + k] / tot);
	}
	for (int k = 0; k < n; ++k) {
		const double *row = A + (size_t)k * n;
		double tot = 0.0;
		for (int l = 0; l < n; ++l) tot += row[l];
		for (int l = 0; l < n; ++l) sum += row[l] * log(row[l] / tot);
	}
	return sum;
}

double psmc_Q(int n, const double *a, const double *e, const double *A, const double *E, double Q0)
{
	double sum = 0.0;
	for (int b = 0; b < 2; ++b)
		for (int k = 0; k < n; ++k) {
			if (e[b * n + k] <= 0.0) return Q_MINUS_INF;
			sum += E[b * n + k] * log(e[b * n + k]);
		}
	for (int k = 0; k < n; ++k)
		for (int l = 0; l < n; ++l) {
			if (a[k * n + l] <= 0.0) return Q_MINUS_INF;
			sum += A[k * n + l] * log(a[k * n + l]);
		}
	return sum - Q0;
}

/* ---- Hooke-Jeeves direct search (Hooke & Jeeves 1961; Bell & Pike, CACM 9(9):684-685; Tomlin & Smith, CACM 12(11):637-638).
 *
 * The search compares nearly equal objective values with `<`, so a byte-identical .psmc needs the reference's search to the
 * last evaluation: the same trial points (each coordinate produced by the same additions, rounding included), in the same
 * order, with the same acceptance tests -- kmin.c:48-107 of lh3/psmc, Copyright (c) 2008 Heng Li, MIT License (full notice
 * in /NOTICE).  Own here: the decomposition into a search state with three moves (sweep, advance, shrink); the arithmetic
 * of each move is the reference's and is marked. */
typedef struct {
	psmc_objective f; void *data;
	int n, calls, max_calls;
	double *step;  /* signed step per axis: the sign remembers which direction last helped */
} hj_search;

static double hj_eval(hj_search *S, double *pt) { ++S->calls; return S->f(S->n, pt, S->data); }

/* Sweep: along every axis in turn move `pt` by the axis' step; if that is no better, turn the step round and try the other side;
 * if neither helps, put the coordinate back.  The three updates of pt[k] are kmin.c:54-63's -- pt[k] + d, then + (-d + -d), then
 * - (-d): not the same double as "the old value" in general, and the search's later points depend on it. */
static double hj_sweep(hj_search *S, double *pt, double f_pt)
{
	for (int k = 0; k < S->n; ++k) {
		pt[k] += S->step[k];
		double f_try = hj_eval(S, pt);
		if (f_try < f_pt) { f_pt = f_try; continue; }
		S->step[k] = 0.0 - S->step[k];
		pt[k] += S->step[k] + S->step[k];
		f_try = hj_eval(S, pt);
		if (f_try < f_pt) f_pt = f_try;
		else pt[k] -= S->step[k];
	}
	return f_pt;
}

/* Advance: the sweep's result becomes the base, and the next trial point lies as far beyond it as it lies beyond the old base
 * (kmin.c:78-83); every step now points the way its axis moved. */
static void hj_advance(hj_search *S, double *base, double *trial)
{
	for (int k = 0; k < S->n; ++k) {
		const double old = base[k];
		S->step[k] = trial[k] > base[k] ? fabs(S->step[k]) : 0.0 - fabs(S->step[k]);
		base[k] = trial[k];
		trial[k] = trial[k] + trial[k] - old;
	}
}

/* did the last sweep leave the pattern point on any axis by more than half a step?  (kmin.c:90-93) */
static int hj_left_pattern(const hj_search *S, const double *base, const double *trial)
{
	for (int k = 0; k < S->n; ++k)
		if (fabs(trial[k] - base[k]) > .5 * fabs(S->step[k])) return 1;
	return 0;
}

/* Minimises f from x (updated in place to the last accepted base point).  Returns the value of the last SWEEP, which is what
 * kmin_hj returns -- not necessarily f(x) (the caller, like em.c:67, only prints it). */
double psmc_hooke_jeeves(psmc_objective f, int n, double *x, void *data, double r, double eps, int max_calls)
{
	hj_search S = {f, data, n, 0, max_calls, (double *)calloc((size_t)n, sizeof(double))};
	double *trial = (double *)calloc((size_t)n, sizeof(double));
	for (int k = 0; k < n; ++k) { S.step[k] = fabs(x[k]) * r; if (S.step[k] == 0) S.step[k] = r; }
	double radius = r, f_base = hj_eval(&S, x), f_trial = f_base;
	for (;;) {
		memcpy(trial, x, sizeof(double) * (size_t)n);
		f_trial = hj_sweep(&S, trial, f_base);
		while (f_trial < f_base) { /* pattern moves for as long as they pay */
			hj_advance(&S, x, trial);
			f_base = f_trial;
			if (S.calls >= S.max_calls) break;
			f_trial = hj_eval(&S, trial);
			f_trial = hj_sweep(&S, trial, f_trial);
			if (f_trial >= f_base || !hj_left_pattern(&S, x, trial)) break;
		}
		if (radius < eps || S.calls >= S.max_calls) break;
		radius *= r; /* shrink: the same search, half the steps */
		for (int k = 0; k < n; ++k) S.step[k] *= r;
	}
	free(trial); free(S.step);
	return f_trial;
}

typedef struct {
	psmc_model *m; const double *A, *E; double Q0; int calls;
	int simd;          /* the CPU has AVX2: vectorised log factors (fastq.c) */
	double *sums, *lf; /* fast objective: SL | SU | DG | CL | CU (5N, from A, once per round); 7N log factors per call */
} q_ctx;

/* Fast objective: log a[k][l] = log FL_k + log qa_l (l<k), log FU_k + log c_l (l>k), so
 * sum_kl A[k][l] log a[k][l] needs 5N logarithms and the triangular row / column sums of A instead of
 * N*N logarithms and the matrix itself.  Equal to psmc_Q up to rounding (~1e-13 relative): the direct
 * search may take a different path, which PSMC_HIP_MODE=fast accepts by definition. */
static double neg_Q_fast(int n, double *x, void *data)
{
	q_ctx *c = (q_ctx *)data;
	const int N = c->m->pat.n_states;
	++c->calls;
	for (int i = 0; i < n; ++i) c->m->params[i] = fabs(x[i]);
	if (!(c->simd ? psmc_model_logfactors_simd(c->m, c->lf) : psmc_model_logfactors(c->m, c->lf))) return -Q_MINUS_INF;
	const double *lFL = c->lf, *lFU = lFL + N, *lD = lFU + N, *lqa = lD + N, *lc = lqa + N, *le0 = lc + N, *le1 = le0 + N;
	const double *SL = c->sums, *SU = SL + N, *DG = SU + N, *CL = DG + N, *CU = CL + N;
	double sum = 0.0;
	for (int k = 0; k < N; ++k) sum += c->E[k] * le0[k] + c->E[N + k] * le1[k];
	for (int k = 0; k < N; ++k) sum += SL[k] * lFL[k] + SU[k] * lFU[k] + DG[k] * lD[k] + CL[k] * lqa[k] + CU[k] * lc[k];
	return -(sum - c->Q0);
}

static double neg_Q(int n, double *x, void *data)
{	/* em.c:15-25 */
	q_ctx *c = (q_ctx *)data;
	++c->calls;
	for (int i = 0; i < n; ++i) c->m->params[i] = fabs(x[i]);
	psmc_model_update(c->m);
	return -psmc_Q(c->m->pat.n_states, c->m->a, c->m->e, c->A, c->E, c->Q0);
}

int psmc_em_round(psmc_model *m, const psmc_input *in, psmc_estep_backend *be, FILE *out)
{
	const int N = m->pat.n_states;
	double *A = (double *)calloc((size_t)N * N, sizeof(double)), *E = (double *)calloc((size_t)2 * N, sizeof(double));
	double LL = 0.0;
	(void)in;
	/* E-step on the device: em.c:33-55.  With the O(N) objective only the triangular sums of A are needed, and
	 * a backend that can produce them directly skips the n*n counts altogether. */
	const int factored = m->fast_mstep && be->estep_factored != 0;
	double *sums = factored ? (double *)calloc((size_t)5 * N, sizeof(double)) : 0;
	const double t_e0 = now_ms();
	int rc = factored ? be->estep_factored(be->self, m->a, m->e, m->a0, sums, E, &LL)
	                  : be->estep(be->self, m->a, m->e, m->a0, A, E, &LL, 0);
	const double t_e1 = now_ms();
	if (rc) { free(A); free(E); free(sums); return rc; }
	const int calls = psmc_em_mstep(m, factored ? 0 : A, E, sums, LL, out);
	if (getenv("PSMC_TIMING")) /* stderr only: the .psmc stream stays byte-identical */
		fprintf(stderr, "[psmc] E-step %.1f ms, M-step %.1f ms (%d objective calls%s)\n", t_e1 - t_e0, now_ms() - t_e1, calls,
		        m->fast_mstep ? ", O(N) objective" : "");
	free(A); free(E); free(sums);
	return 0;
}

/* The M-step half of psmc_em (em.c:56-74) given the sufficient statistics of the E-step: A n*n (or NULL when the
 * five triangular sums are given instead: fast M-step only), E 2*n, LL.  Prints the IT line; returns the number of
 * objective calls.  Touches nothing but *m: replicates can run it concurrently on their own models. */
int psmc_em_mstep(psmc_model *m, const double *A, const double *E, const double *sums_in, double LL, FILE *out)
{
	const int N = m->pat.n_states;
	const int factored = A == 0;
	double *sums = (double *)sums_in;
	/* M-step: em.c:56-68 */
	q_ctx c;
	c.m = m; c.A = A; c.E = E; c.calls = 0;
	c.Q0 = factored ? 0.0 : psmc_Q0(N, A, E); /* a constant of the search; needs the full matrix (only printed: QD line) */
	c.sums = c.lf = 0;
	c.simd = __builtin_cpu_supports("avx2") && !getenv("PSMC_NO_SIMD");
	if (factored) {
		c.sums = sums; c.lf = (double *)calloc((size_t)7 * N, sizeof(double));
	} else if (m->fast_mstep) {
		c.sums = (double *)calloc((size_t)5 * N, sizeof(double)); c.lf = (double *)calloc((size_t)7 * N, sizeof(double));
		for (int k = 0; k < N; ++k)
			for (int l = 0; l < N; ++l) {
				const double v = A[(size_t)k * N + l];
				if (l < k) { c.sums[k] += v; c.sums[3 * N + l] += v; }          /* SL_k, CL_l */
				else if (l > k) { c.sums[N + k] += v; c.sums[4 * N + l] += v; } /* SU_k, CU_l */
				else c.sums[2 * N + k] = v;
			}
	}
	m->lk = LL;
	double *x = (double *)calloc((size_t)m->n_params, sizeof(double));
	memcpy(x, m->params, sizeof(double) * (size_t)m->n_params);
	if (factored) { double *xx = (double *)malloc(sizeof(double) * (size_t)m->n_params); /* Q at the current parameters, same objective */
		memcpy(xx, m->params, sizeof(double) * (size_t)m->n_params);
		const int keep = c.calls; m->Q0 = -neg_Q_fast(m->n_params, xx, &c); c.calls = keep; free(xx);
	} else m->Q0 = psmc_Q(N, m->a, m->e, A, E, c.Q0);
	m->Q1 = -psmc_hooke_jeeves(m->fast_mstep ? neg_Q_fast : neg_Q, m->n_params, x, &c, HJ_RADIUS, HJ_EPS, HJ_MAXCALL);
	if (m->fast_mstep) { psmc_model_update(m); if (!factored) free(c.sums); free(c.lf); } /* a/e/a0/sigma of the LAST trial point, like em.c:21-22 */
	fprintf(out, "IT\t%d\n", c.calls);
	free(x);
	{ /* posterior state occupancy, em.c:69-74 */
		double sum = 0.0;
		for (int k = 0; k < N; ++k) sum += E[k] + E[N + k];
		for (int k = 0; k < N; ++k) m->post_sigma[k] = (E[k] + E[N + k]) / sum;
	}
	return c.calls;
}
