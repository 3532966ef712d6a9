/* psmc_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the PSMC E-step (scaled forward-backward + expected
 * counts) in the reference's exact floating-point operation order, on flat
 * row-major arrays.  Every function cites the reference file:line it follows
 * (paths relative to the lh3/psmc checkout).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * link or call this library, and only as the checker / reported CPU baseline.
 * The product path (psmc_amd/, libpsmc_hip.so, the psmc CLI) never does.
 *
 * Parity pin: validated bit-for-bit against the reference itself compiled from
 * its own sources (oracle/_ref, see oracle/Makefile) and against the golden
 * vectors under tests/golden/ that were dumped from that build
 * (tests/golden/make_golden.py).  The reference ships no tests of its own.
 *
 * Conventions (all arrays caller-owned, row-major, FP64):
 *   n      number of hidden states (psmc's n+1)
 *   a      n*n   a[k*n+l] = P(k -> l)                 (khmm.h:34)
 *   e      3*n   e[b*n+k], b=0 hom,1 het,2 missing(=1) (khmm.c:19-21)
 *   a0     n     initial distribution                 (khmm.h:36)
 *   ae     3*n*n ae[(b*n+k)*n+l] = e[b][l]*a[k][l]    (khmm.c:194-206)
 *   seq    L bytes in {0,1,2}, 0-indexed (psmc_seq_t.seq, psmc.h:22-26)
 *   f,b    (L+1)*n, row u = position u (1-indexed like khmm.c), row 0 unused
 *   s      L+1 scaling factors, s[0]=1
 */
#ifndef PSMC_ORACLE_H
#define PSMC_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define ORC_TINY 1e-25 /* HMM_TINY, khmm.h:28 */

void   orc_pre_backward(int n, const double *a, const double *e, double *ae);
void   orc_forward(int n, const double *a, const double *e, const double *a0,
                   int L, const uint8_t *seq, double *f, double *s);
double orc_backward(int n, const double *ae, const double *e, const double *a0,
                    int L, const uint8_t *seq, const double *s, double *b);
double orc_lk(int L, const double *s);
void   orc_expect(int n, const double *ae, const double *e, const double *a0,
                  int L, const uint8_t *seq, const double *f, const double *b,
                  const double *s, double *A, double *E /*3*n*/, double *A0);
/* One whole E-step over a batch of segments: em.c:33-55 + 60.
 * A (n*n), E (2*n: rows b=0,1 only, khmm.c:355), A0 (n, may be NULL) are the
 * summed statistics he_sum; per_seg_* (may be NULL) receive each segment's own
 * he (A: n*n, E: 3*n incl. the missing row), LL and underflow check value. */
void   orc_estep(int n, const double *a, const double *e, const double *a0,
                 int n_seg, const uint8_t *const *seq, const int32_t *L,
                 double *A, double *E, double *A0, double *LL,
                 double *per_seg_A, double *per_seg_E, double *per_seg_LL,
                 double *per_seg_chk);
/* Posterior decoding pieces used by aux.c:150-200 (row "next" f-3). */
void   orc_post_decode(int n, int L, const double *f, const double *b,
                       const double *s, int32_t *path /*L+1*/, double *maxp /*L+1*/);
/* -D branch of psmc_decode, aux.c:183-200: post (L+1)*n, row u = hmm_post_state (khmm.c:285-292);
 * recomb[u] = 1 - sum_l f[u][l]*a[l][l]*b[u+1][l]*e[seq[u+1]][l] for u < L, 0 at u = L (L+1 entries, [0] unused). */
void   orc_post_full(int n, const double *a, const double *e, int L, const uint8_t *seq, const double *f,
                     const double *b, const double *s, double *post, double *recomb);
/* -c branch of psmc_decode, aux.c:202-219: cnt[l*n_cnt+j] += post[u][l]*cnt1[(u-1)*n_cnt+j], u = 1..min(L, l1). */
void   orc_post_counts(int n, int L, const double *f, const double *b, const double *s, const int32_t *cnt1,
                       int32_t l1, int32_t n_cnt, double *cnt);
/* Q0 / Q as consumed by the M-step (khmm.c:326-342, 363-382); E is 2*n. */
double orc_Q0(int n, const double *A, const double *E);
double orc_Q(int n, const double *a, const double *e, const double *A,
             const double *E, double Q0);
#ifdef __cplusplus
}
#endif
#endif
