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
#include "psmc_host.h"

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

/* one exploratory sweep around x1 along every axis (kmin.c:48-66) */
static double explore(psmc_objective f, int n, double *x1, void *data, double fx1, double *dx, int *calls)
{
	for (int k = 0; k < n; ++k) {
		x1[k] += dx[k];
		double ft = f(n, x1, data); ++*calls;
		if (ft < fx1) fx1 = ft;
		else { /* try the opposite direction */
			dx[k] = 0.0 - dx[k];
			x1[k] += dx[k] + dx[k];
			ft = f(n, x1, data); ++*calls;
			if (ft < fx1) fx1 = ft;
			else x1[k] -= dx[k];
		}
	}
	return fx1;
}

double psmc_hooke_jeeves(psmc_objective f, int n, double *x, void *data, double r, double eps, int max_calls)
{
	double *x1 = (double *)calloc((size_t)n, sizeof(double)), *dx = (double *)calloc((size_t)n, sizeof(double));
	int calls = 0;
	for (int k = 0; k < n; ++k) { dx[k] = fabs(x[k]) * r; if (dx[k] == 0) dx[k] = r; }
	double radius = r, fx, fx1;
	fx1 = fx = f(n, x, data); ++calls;
	for (;;) {
		memcpy(x1, x, sizeof(double) * (size_t)n);
		fx1 = explore(f, n, x1, data, fx, dx, &calls);
		while (fx1 < fx) { /* pattern moves */
			for (int k = 0; k < n; ++k) {
				const double t = x[k];
				dx[k] = x1[k] > x[k] ? fabs(dx[k]) : 0.0 - fabs(dx[k]);
				x[k] = x1[k];
				x1[k] = x1[k] + x1[k] - t;
			}
			fx = fx1;
			if (calls >= max_calls) break;
			fx1 = f(n, x1, data); ++calls;
			fx1 = explore(f, n, x1, data, fx1, dx, &calls);
			if (fx1 >= fx) break;
			int k;
			for (k = 0; k < n; ++k)
				if (fabs(x1[k] - x[k]) > .5 * fabs(dx[k])) break;
			if (k == n) break;
		}
		if (radius >= eps) {
			if (calls >= max_calls) break;
			radius *= r;
			for (int k = 0; k < n; ++k) dx[k] *= r;
		} else break;
	}
	free(x1); free(dx);
	return fx1;
}

typedef struct { psmc_model *m; const double *A, *E; double Q0; int calls; } q_ctx;

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
	/* E-step on the device: em.c:33-55 */
	int rc = be->estep(be->self, m->a, m->e, m->a0, A, E, &LL, 0);
	if (rc) { free(A); free(E); return rc; }
	/* M-step: em.c:56-68 */
	q_ctx c;
	c.m = m; c.A = A; c.E = E; c.calls = 0;
	c.Q0 = psmc_Q0(N, A, E);
	m->lk = LL;
	double *x = (double *)calloc((size_t)m->n_params, sizeof(double));
	memcpy(x, m->params, sizeof(double) * (size_t)m->n_params);
	m->Q0 = psmc_Q(N, m->a, m->e, A, E, c.Q0);
	m->Q1 = -psmc_hooke_jeeves(neg_Q, m->n_params, x, &c, HJ_RADIUS, HJ_EPS, HJ_MAXCALL);
	fprintf(out, "IT\t%d\n", c.calls);
	free(x);
	{ /* posterior state occupancy, em.c:69-74 */
		double sum = 0.0;
		for (int k = 0; k < N; ++k) sum += E[k] + E[N + k];
		for (int k = 0; k < N; ++k) m->post_sigma[k] = (E[k] + E[N + k]) / sum;
	}
	free(A); free(E);
	return 0;
}
