/* ref_shim.c -- TEST INFRASTRUCTURE.  Flat-array entry points into the REAL
 * reference (lh3/psmc) so that python/ctypes can drive it.  This file is ours;
 * it is compiled together with the reference's own, unmodified sources where
 * they lie under $(REF) (default /root/reference) by oracle/Makefile, with the
 * output going only to oracle/_ref/ (git-ignored).  Nothing of the reference
 * is copied into this repository.
 *
 * Uses: (1) validating oracle/psmc_oracle.c bit-for-bit, (2) generating the
 * golden vectors under tests/golden/ (tests/golden/make_golden.py), (3) the
 * "reference" CPU baseline of bench.py when the prebuilt .so is present.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <stdint.h>
#include "psmc.h"   /* from $(REF) */
#include "khmm.h"
#include "kmin.h"

void psmc_read_seq(const char *fn, psmc_par_t *pp); /* cli.c:103 (not in psmc.h) */

static hmm_par_t *mk_par(int n, const double *a, const double *e, const double *a0)
{
	hmm_par_t *hp = hmm_new_par(2, n); /* sets e[2][*]=1 */
	for (int k = 0; k < n; ++k) {
		hp->a0[k] = a0[k];
		for (int l = 0; l < n; ++l) hp->a[k][l] = a[k * n + l];
		hp->e[0][k] = e[k]; hp->e[1][k] = e[n + k];
	}
	return hp;
}

static double underflow_chk(const hmm_par_t *hp, const hmm_data_t *hd)
{	/* the quantity khmm.c:237-238 computes and only prints */
	double tmp = 0.0;
	for (int l = 0; l < hp->n; ++l)
		tmp += hp->a0[l] * hd->b[1][l] * hp->e[(int)hd->seq[1]][l];
	return tmp;
}

/* forward + backward of one segment, dumping f (L+1 rows), b, s */
int ref_fwd_bwd(int n, const double *a, const double *e, const double *a0, int L,
                const uint8_t *seq, double *f, double *b, double *s, double *lk)
{
	hmm_par_t *hp = mk_par(n, a, e, a0);
	hmm_pre_backward(hp);
	hmm_data_t *hd = hmm_new_data(L, (const char*)seq, hp);
	hmm_forward(hp, hd);
	hmm_backward(hp, hd);
	for (int u = 0; u <= L; ++u) {
		if (f) memcpy(f + (size_t)u * n, hd->f[u], sizeof(double) * n);
		if (b) memcpy(b + (size_t)u * n, hd->b[u], sizeof(double) * n);
		if (s) s[u] = hd->s[u];
	}
	if (lk) *lk = hmm_lk(hd);
	hmm_delete_data(hd); hmm_delete_par(hp);
	return 0;
}

/* The E half of psmc_em (em.c:33-55) on caller-provided HMM parameters. */
int ref_estep(int n, const double *a, const double *e, const double *a0,
              int n_seg, const uint8_t *const *seq, const int32_t *L,
              double *A, double *E, double *A0, double *LL,
              double *per_seg_A, double *per_seg_E, double *per_seg_LL,
              double *per_seg_chk)
{
	hmm_par_t *hp = mk_par(n, a, e, a0);
	hmm_exp_t *sum = hmm_new_exp(hp);
	double ll = 0.0;
	hmm_pre_backward(hp);
	for (int i = 0; i < n_seg; ++i) {
		hmm_data_t *hd = hmm_new_data(L[i], (const char*)seq[i], hp);
		hmm_forward(hp, hd);
		hmm_backward(hp, hd);
		double l1 = hmm_lk(hd);
		ll += l1;
		hmm_exp_t *he = hmm_expect(hp, hd);
		hmm_add_expect(he, sum);
		if (per_seg_A)
			for (int k = 0; k < n; ++k)
				memcpy(per_seg_A + ((size_t)i * n + k) * n, he->A[k], sizeof(double) * n);
		if (per_seg_E)
			for (int b = 0; b < 3; ++b)
				memcpy(per_seg_E + ((size_t)i * 3 + b) * n, he->E[b], sizeof(double) * n);
		if (per_seg_LL) per_seg_LL[i] = l1;
		if (per_seg_chk) per_seg_chk[i] = underflow_chk(hp, hd);
		hmm_delete_exp(he); hmm_delete_data(hd);
	}
	for (int k = 0; k < n; ++k) memcpy(A + (size_t)k * n, sum->A[k], sizeof(double) * n);
	memcpy(E, sum->E[0], sizeof(double) * n);
	memcpy(E + n, sum->E[1], sizeof(double) * n);
	if (A0) memcpy(A0, sum->A0, sizeof(double) * n);
	*LL = ll;
	hmm_delete_exp(sum); hmm_delete_par(hp);
	return 0;
}

/* pattern -> par_map (cli.c:66-99).  Returns psmc's n (states-1). */
int ref_parse_pattern(const char *pattern, int *n_free, int *par_map /* >= n+1 */)
{
	int n, nf, *m = psmc_parse_pattern(pattern, &nf, &n);
	memcpy(par_map, m, sizeof(int) * (n + 1));
	*n_free = nf; free(m);
	return n;
}

static psmc_par_t *mk_pp(const char *pattern, double alpha, double dt0)
{
	psmc_par_t *pp = (psmc_par_t*)calloc(1, sizeof(psmc_par_t));
	pp->pattern = strdup(pattern);
	pp->par_map = psmc_parse_pattern(pattern, &pp->n_free, &pp->n);
	pp->alpha = alpha; pp->max_t = 15.0; pp->tr_ratio = 4.0; pp->dt0 = dt0;
	if (dt0 >= 0) pp->flag |= PSMC_F_DIVERG;
	pp->fpout = 0;
	return pp;
}
static void rm_pp(psmc_par_t *pp)
{
	for (int i = 0; i < pp->n_seqs; ++i) { free(pp->seqs[i].name); free(pp->seqs[i].seq); }
	free(pp->seqs); free(pp->par_map); free(pp->pattern); free(pp);
}
static psmc_data_t *mk_pd(psmc_par_t *pp, const double *params)
{	/* psmc_new_data (core.c:21-50) with inp_pa = params */
	int np = pp->n_free + PSMC_N_PARAMS + ((pp->flag & PSMC_F_DIVERG) ? 1 : 0);
	pp->inp_pa = (double*)malloc(sizeof(double) * (np + 1));
	memcpy(pp->inp_pa, params, sizeof(double) * np);
	psmc_data_t *pd = psmc_new_data(pp);
	free(pp->inp_pa); pp->inp_pa = 0;
	return pd;
}

/* psmc_update_hmm (core.c:61-133): population params -> HMM params.
 * e_out is 3*n_states (row 2 = 1.0).  t_out has n+2 entries. */
int ref_hmm_params(const char *pattern, const double *params, double alpha, double dt0,
                   double *t_out, double *a, double *e_out, double *a0, double *sigma,
                   double *C_pi, double *C_sigma)
{
	psmc_par_t *pp = mk_pp(pattern, alpha, dt0);
	psmc_data_t *pd = mk_pd(pp, params);
	int N = pp->n + 1;
	for (int k = 0; k < N; ++k) {
		memcpy(a + (size_t)k * N, pd->hp->a[k], sizeof(double) * N);
		a0[k] = pd->hp->a0[k]; sigma[k] = pd->sigma[k];
		e_out[k] = pd->hp->e[0][k]; e_out[N + k] = pd->hp->e[1][k]; e_out[2 * N + k] = pd->hp->e[2][k];
	}
	memcpy(t_out, pd->t, sizeof(double) * (pp->n + 2));
	*C_pi = pd->C_pi; *C_sigma = pd->C_sigma;
	psmc_delete_data(pd); rm_pp(pp);
	return N;
}

/* One whole psmc_em round (em.c:27-78) from given params on given segments.
 * Returns the IT count parsed back from the reference's own output line. */
int ref_em_round(const char *pattern, double *params_io, double alpha, double dt0,
                 int n_seg, const uint8_t *const *seq, const int32_t *L,
                 double *lk, double *Q0, double *Q1, double *post_sigma)
{
	psmc_par_t *pp = mk_pp(pattern, alpha, dt0);
	pp->n_seqs = n_seg;
	pp->seqs = (psmc_seq_t*)calloc(n_seg, sizeof(psmc_seq_t));
	for (int i = 0; i < n_seg; ++i) {
		pp->seqs[i].L = L[i];
		pp->seqs[i].seq = (char*)malloc(L[i]);
		memcpy(pp->seqs[i].seq, seq[i], L[i]);
		pp->seqs[i].name = strdup("x");
	}
	psmc_data_t *pd = mk_pd(pp, params_io);
	char *buf = 0; size_t sz = 0;
	pp->fpout = open_memstream(&buf, &sz);
	psmc_em(pp, pd);
	fclose(pp->fpout); pp->fpout = 0;
	int it = -1;
	if (buf) { sscanf(buf, "IT\t%d", &it); free(buf); }
	memcpy(params_io, pd->params, sizeof(double) * pd->n_params);
	*lk = pd->lk; *Q0 = pd->Q0; *Q1 = pd->Q1;
	memcpy(post_sigma, pd->post_sigma, sizeof(double) * (pp->n + 1));
	psmc_delete_data(pd); rm_pp(pp);
	return it;
}

/* psmc_resamp (aux.c:8-47) with a fixed srand48 seed: returns the number of
 * resampled segments and which original index each one is. */
int ref_resample(long seed, int n_seg, const int32_t *L, int32_t *picked, int max_out)
{
	psmc_par_t *pp = (psmc_par_t*)calloc(1, sizeof(psmc_par_t));
	pp->n_seqs = n_seg;
	pp->seqs = (psmc_seq_t*)calloc((n_seg + 0xff) & ~0xff, sizeof(psmc_seq_t));
	for (int i = 0; i < n_seg; ++i) {
		char nm[32]; snprintf(nm, sizeof nm, "%d", i);
		pp->seqs[i].name = strdup(nm);
		pp->seqs[i].L = L[i];
		pp->seqs[i].seq = (char*)calloc(L[i] ? L[i] : 1, 1);
	}
	srand48(seed);
	psmc_resamp(pp);
	int m = pp->n_seqs;
	for (int i = 0; i < m && i < max_out; ++i) picked[i] = atoi(pp->seqs[i].name);
	for (int i = 0; i < m; ++i) { free(pp->seqs[i].name); free(pp->seqs[i].seq); }
	free(pp->seqs); free(pp);
	return m;
}

/* psmc_read_seq (cli.c:103-138): decode a .psmcfa into 0/1/2 bytes.
 * Two-call protocol: first with seq_out=NULL to get counts/lengths. */
int ref_read_psmcfa(const char *fn, int max_seg, int32_t *L, int32_t *L_e, int32_t *n_e,
                    uint8_t *seq_out /* concatenated, may be NULL */, int64_t *sum_L, int *sum_n)
{
	psmc_par_t *pp = (psmc_par_t*)calloc(1, sizeof(psmc_par_t));
	psmc_read_seq(fn, pp);
	int m = pp->n_seqs; size_t off = 0;
	for (int i = 0; i < m && i < max_seg; ++i) {
		L[i] = pp->seqs[i].L; L_e[i] = pp->seqs[i].L_e; n_e[i] = pp->seqs[i].n_e;
		if (seq_out) memcpy(seq_out + off, pp->seqs[i].seq, pp->seqs[i].L);
		off += pp->seqs[i].L;
	}
	*sum_L = pp->sum_L; *sum_n = pp->sum_n;
	rm_pp(pp);
	return m;
}

/* kmin_hj on a convex test function, for a unit KAT of the minimiser */
static double quad(int n, double *x, void *data)
{
	double s = 0.0, *c = (double*)data;
	for (int i = 0; i < n; ++i) s += (i + 1) * (x[i] - c[i]) * (x[i] - c[i]) + 0.1 * fabs(x[i]);
	return s;
}
double ref_kmin_quad(int n, double *x_io, double *centre)
{
	return kmin_hj(quad, n, x_io, centre, KMIN_RADIUS, KMIN_EPS, KMIN_MAXCALL);
}
