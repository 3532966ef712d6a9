/* psmc_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see psmc_oracle.h).
 *
 * Plain-C restatement of the reference E-step in its exact IEEE-754 double
 * operation order (no FMA contraction: build with -ffp-contract=off; every sum
 * evaluated left to right in the same index order as the reference loops).
 * Citations are file:line in the lh3/psmc checkout.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "psmc_oracle.h"

/* khmm.c:194-206  hmm_pre_backward: ae[b][k][l] = e[b][l] * a[k][l] */
void orc_pre_backward(int n, const double *a, const double *e, double *ae)
{
	for (int b = 0; b < 3; ++b)
		for (int k = 0; k < n; ++k) {
			double *dst = ae + ((size_t)b * n + k) * n;
			for (int l = 0; l < n; ++l) dst[l] = e[b * n + l] * a[k * n + l];
		}
}

/* khmm.c:145-190  hmm_forward.  f row u is position u (1..L); s[0]=1. */
void orc_forward(int n, const double *a, const double *e, const double *a0,
                 int L, const uint8_t *seq, double *f, double *s)
{
	double *at = (double*)malloc(sizeof(double) * n * n);
	for (int k = 0; k < n; ++k)              /* khmm.c:162-166 transpose */
		for (int l = 0; l < n; ++l) at[k * n + l] = a[l * n + k];
	s[0] = 1.0;
	for (int k = 0; k < n; ++k) f[k] = 0.0; /* khmm.c:168-169 */
	{ /* khmm.c:171-174: position 1 */
		const double *e1 = e + (size_t)seq[0] * n;
		double *f1 = f + n, sum = 0.0;
		for (int k = 0; k < n; ++k) { f1[k] = a0[k] * e1[k]; sum += f1[k]; }
		for (int k = 0; k < n; ++k) f1[k] /= sum;
		s[1] = sum;
	}
	for (int u = 2; u <= L; ++u) { /* khmm.c:176-185 */
		double *fu = f + (size_t)u * n;
		const double *fp = fu - n, *eu = e + (size_t)seq[u - 1] * n;
		double sum = 0.0;
		for (int k = 0; k < n; ++k) {
			const double *col = at + (size_t)k * n;
			double tmp = 0.0;
			for (int l = 0; l < n; ++l) tmp += fp[l] * col[l];
			fu[k] = eu[k] * tmp;
			sum += fu[k];
		}
		for (int k = 0; k < n; ++k) fu[k] /= sum;
		s[u] = sum;
	}
	free(at);
}

/* khmm.c:210-241  hmm_backward.  Returns the underflow check value
 * sum_l a0[l]*b[1][l]*e[o_1][l] (khmm.c:237-238), which the reference compares
 * against 1 +- 1e-6 and reports on stderr. */
double orc_backward(int n, const double *ae, const double *e, const double *a0,
                    int L, const uint8_t *seq, const double *s, double *b)
{
	double *bL = b + (size_t)L * n;
	for (int k = 0; k < n; ++k) bL[k] = 1.0 / s[L]; /* khmm.c:226 */
	for (int u = L - 1; u >= 1; --u) {              /* khmm.c:228-235 */
		const double *bn = b + (size_t)(u + 1) * n;
		const double *blk = ae + (size_t)seq[u] * n * n; /* symbol of position u+1 */
		double *bu = b + (size_t)u * n;
		for (int k = 0; k < n; ++k) {
			const double *q = blk + (size_t)k * n;
			double tmp = 0.0;
			for (int l = 0; l < n; ++l) tmp += q[l] * bn[l];
			bu[k] = tmp / s[u];
		}
	}
	double chk = 0.0;
	const double *e1 = e + (size_t)seq[0] * n, *b1 = b + n;
	for (int l = 0; l < n; ++l) chk += a0[l] * b1[l] * e1[l];
	return chk;
}

/* khmm.c:245-260  hmm_lk: running product flushed through log() */
double orc_lk(int L, const double *s)
{
	double sum = 0.0, prod = 1.0;
	for (int u = 1; u <= L; ++u) {
		prod *= s[u];
		if (prod < ORC_TINY || prod >= 1.0 / ORC_TINY) {
			sum += log(prod);
			prod = 1.0;
		}
	}
	sum += log(prod);
	return sum;
}

/* khmm.c:297-324  hmm_expect.  A: n*n, E: 3*n, A0: n (may be NULL). */
void orc_expect(int n, const double *ae, const double *e, const double *a0,
                int L, const uint8_t *seq, const double *f, const double *b,
                const double *s, double *A, double *E, double *A0)
{
	for (int i = 0; i < n * n; ++i) A[i] = ORC_TINY; /* khmm.c:305-308 */
	for (int i = 0; i < 3 * n; ++i) E[i] = ORC_TINY;
	for (int u = 1; u < L; ++u) {                    /* khmm.c:310-319 */
		const double *fu = f + (size_t)u * n, *bu = b + (size_t)u * n, *bn = bu + n;
		const double ss = s[u];
		double *Ec = E + (size_t)seq[u - 1] * n;
		const double *blk = ae + (size_t)seq[u] * n * n;
		for (int k = 0; k < n; ++k) {
			const double *q = blk + (size_t)k * n;
			double *AA = A + (size_t)k * n, fuk = fu[k];
			for (int l = 0; l < n; ++l) AA[l] += fuk * q[l] * bn[l];
			Ec[k] += fuk * bu[k] * ss;
		}
	}
	if (A0) { /* khmm.c:321-322; A0 starts from calloc'ed zeros (khmm.c:66) */
		const double *e1 = e + (size_t)seq[0] * n, *b1 = b + n;
		for (int l = 0; l < n; ++l) A0[l] = 0.0 + a0[l] * e1[l] * b1[l];
	}
}

/* em.c:33-55,60 with hmm_add_expect (khmm.c:346-359) folded in. */
void orc_estep(int n, const double *a, const double *e, const double *a0,
               int n_seg, const uint8_t *const *seq, const int32_t *L,
               double *A, double *E, double *A0, double *LL,
               double *per_seg_A, double *per_seg_E, double *per_seg_LL,
               double *per_seg_chk)
{
	double *ae = (double*)malloc(sizeof(double) * 3 * n * n);
	double *hA = (double*)malloc(sizeof(double) * n * n);
	double *hE = (double*)malloc(sizeof(double) * 3 * n);
	double *hA0 = (double*)malloc(sizeof(double) * n);
	double ll = 0.0;
	memset(A, 0, sizeof(double) * n * n);  /* hmm_new_exp callocs, khmm.c:60-69 */
	memset(E, 0, sizeof(double) * 2 * n);
	if (A0) memset(A0, 0, sizeof(double) * n);
	orc_pre_backward(n, a, e, ae);          /* em.c:34 */
	for (int i = 0; i < n_seg; ++i) {       /* em.c:36 */
		const int Li = L[i];
		double *f = (double*)malloc(sizeof(double) * (size_t)(Li + 1) * n);
		double *b = (double*)malloc(sizeof(double) * (size_t)(Li + 1) * n);
		double *s = (double*)malloc(sizeof(double) * (size_t)(Li + 1));
		orc_forward(n, a, e, a0, Li, seq[i], f, s);                 /* em.c:46 */
		double chk = orc_backward(n, ae, e, a0, Li, seq[i], s, b);  /* em.c:47 */
		double l1 = orc_lk(Li, s);                                  /* em.c:48 */
		ll += l1;
		orc_expect(n, ae, e, a0, Li, seq[i], f, b, s, hA, hE, hA0); /* em.c:49 */
		for (int k = 0; k < n; ++k) {                               /* khmm.c:350-354 */
			if (A0) A0[k] += hA0[k];
			for (int l = 0; l < n; ++l) A[k * n + l] += hA[k * n + l];
		}
		for (int bb = 0; bb < 2; ++bb)                              /* khmm.c:355-358: b < m only */
			for (int l = 0; l < n; ++l) E[bb * n + l] += hE[bb * n + l];
		if (per_seg_A) memcpy(per_seg_A + (size_t)i * n * n, hA, sizeof(double) * n * n);
		if (per_seg_E) memcpy(per_seg_E + (size_t)i * 3 * n, hE, sizeof(double) * 3 * n);
		if (per_seg_LL) per_seg_LL[i] = l1;
		if (per_seg_chk) per_seg_chk[i] = chk;
		free(f); free(b); free(s);
	}
	*LL = ll;
	free(ae); free(hA); free(hE); free(hA0);
}

/* khmm.c:264-281  hmm_post_decode: argmax_k f*b*s, first maximum wins */
void orc_post_decode(int n, int L, const double *f, const double *b,
                     const double *s, int32_t *path, double *maxp)
{
	path[0] = 0; if (maxp) maxp[0] = 0.0;
	for (int u = 1; u <= L; ++u) {
		const double *fu = f + (size_t)u * n, *bu = b + (size_t)u * n, su = s[u];
		double best = -1.0; int arg = -1;
		for (int k = 0; k < n; ++k) {
			double p = fu[k] * bu[k] * su;
			if (best < p) { best = p; arg = k; }
		}
		path[u] = arg;
		if (maxp) maxp[u] = best;
	}
}

/* aux.c:183-200 (full decoding) with hmm_post_state khmm.c:285-292 */
void orc_post_full(int n, const double *a, const double *e, int L, const uint8_t *seq, const double *f,
                   const double *b, const double *s, double *post, double *recomb)
{
	for (int k = 1; k <= L; ++k) {
		double p;
		if (k < L) {                                               /* aux.c:189-193 */
			const double *fu = f + (size_t)k * n, *bu1 = b + (size_t)(k + 1) * n;
			const double *eu1 = e + (size_t)seq[k] * n;            /* hd->seq[k+1]: 1-indexed there, 0-indexed here */
			p = 0.0;
			for (int l = 0; l < n; ++l) p += fu[l] * a[(size_t)l * n + l] * bu1[l] * eu1[l];
			p = 1.0 - p;
		} else p = 0.0;
		if (recomb) recomb[k] = p;
		if (post) {                                                /* khmm.c:288-291 */
			const double ss = s[k], *fu = f + (size_t)k * n, *bu = b + (size_t)k * n;
			for (int l = 0; l < n; ++l) post[(size_t)k * n + l] = fu[l] * bu[l] * ss;
		}
	}
}

/* aux.c:202-219 */
void orc_post_counts(int n, int L, const double *f, const double *b, const double *s, const int32_t *cnt1,
                     int32_t l1, int32_t n_cnt, double *cnt)
{
	const int min_l = L < l1 ? L : l1;
	for (int k = 1; k <= min_l; ++k) {
		const double ss = s[k], *fu = f + (size_t)k * n, *bu = b + (size_t)k * n;
		for (int l = 0; l < n; ++l) {
			const double prob = fu[l] * bu[l] * ss;                /* hmm_post_state */
			for (int j = 0; j < n_cnt; ++j) cnt[(size_t)l * n_cnt + j] += prob * cnt1[(size_t)(k - 1) * n_cnt + j];
		}
	}
}

/* khmm.c:326-342  hmm_Q0 (m = 2 symbols) */
double orc_Q0(int n, const double *A, const double *E)
{
	double sum = 0.0;
	for (int k = 0; k < n; ++k) {
		double tmp = 0.0;
		for (int b = 0; b < 2; ++b) tmp += E[b * n + k];
		for (int b = 0; b < 2; ++b) sum += E[b * n + k] * log(E[b * n + k] / tmp);
	}
	for (int k = 0; k < n; ++k) {
		const double *Ak = A + (size_t)k * n;
		double tmp = 0.0;
		for (int l = 0; l < n; ++l) tmp += Ak[l];
		for (int l = 0; l < n; ++l) sum += Ak[l] * log(Ak[l] / tmp);
	}
	return sum;
}

/* khmm.c:363-382  hmm_Q */
double orc_Q(int n, const double *a, const double *e, const double *A,
             const double *E, double Q0)
{
	double sum = 0.0;
	for (int b = 0; b < 2; ++b)
		for (int k = 0; k < n; ++k) {
			if (e[b * n + k] <= 0.0) return -1e300;
			sum += E[b * n + k] * log(e[b * n + k]);
		}
	for (int k = 0; k < n; ++k)
		for (int l = 0; l < n; ++l) {
			if (a[k * n + l] <= 0.0) return -1e300;
			sum += A[k * n + l] * log(a[k * n + l]);
		}
	return sum - Q0;
}
