/* main.c -- the `psmc` executable: lh3/psmc's command line (main.c:6-34) with
 * the E-step bound to libpsmc_hip.so.  There is no CPU E-step in this binary:
 * without a visible AMD GPU it exits with an error.
 *   PSMC_HIP_MODE=exact (default: .psmc byte-identical to the reference) | fast
 *   PSMC_HIP_DEVICE=<index>                                                    */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "psmc_host.h"
#include "psmc_hip.h"

typedef struct { psmc_hip_ctx *ctx; int n_seg, n_states; double *chk; char msg[256]; } hip_be;

static int hb_load(void *self, int n_seg, const uint8_t *const *sym, const int32_t *L)
{
	hip_be *h = (hip_be *)self;
	h->n_seg = n_seg;
	h->chk = (double *)realloc(h->chk, sizeof(double) * (size_t)n_seg);
	return psmc_hip_load_segments(h->ctx, n_seg, sym, L);
}
static int hb_estep(void *self, const double *a, const double *e, const double *a0, double *A, double *E, double *LL,
                    double *chk)
{
	hip_be *h = (hip_be *)self;
	int rc = psmc_hip_estep(h->ctx, a, e, a0, A, E, 0, LL, h->chk);
	if (rc) return rc;
	for (int i = 0; i < h->n_seg; ++i) { /* the diagnostic of khmm.c:239-240 */
		if (h->chk[i] > 1.0 + 1e-6 || h->chk[i] < 1.0 - 1e-6) fprintf(stderr, "++ Underflow may have happened (%lg).\n", h->chk[i]);
		if (chk) chk[i] = h->chk[i];
	}
	return 0;
}
static int hb_estep_factored(void *self, const double *a, const double *e, const double *a0, double *sums, double *E, double *LL)
{
	hip_be *h = (hip_be *)self;
	int rc = psmc_hip_estep_factored(h->ctx, a, e, a0, sums, E, LL);
	if (rc != PSMC_HIP_ENOTSUP) return rc;
	/* a matrix without the two rank-1 triangles (e.g. -C): full counts, triangular sums on the host */
	const int n = h->n_states;
	double *A = (double *)calloc((size_t)n * n, sizeof(double));
	rc = psmc_hip_estep(h->ctx, a, e, a0, A, E, 0, LL, h->chk);
	if (rc == 0) {
		memset(sums, 0, sizeof(double) * 5 * (size_t)n);
		for (int k = 0; k < n; ++k)
			for (int l = 0; l < n; ++l) {
				const double v = A[(size_t)k * n + l];
				if (l < k) { sums[k] += v; sums[3 * n + l] += v; } else if (l > k) { sums[n + k] += v; sums[4 * n + l] += v; } else sums[2 * n + k] = v;
			}
	}
	free(A);
	return rc;
}
static int hb_tables(void *self, int seg, double *f, double *b, double *s) { return psmc_hip_get_tables(((hip_be *)self)->ctx, seg, f, b, s); }
static int hb_decode(void *self, int seg, int32_t *path, double *maxp) { return psmc_hip_decode(((hip_be *)self)->ctx, seg, path, maxp); }
static int hb_posterior(void *self, int seg, double *post, double *recomb) { return psmc_hip_posterior(((hip_be *)self)->ctx, seg, post, recomb); }
static int hb_post_counts(void *self, int seg, const int32_t *cnt1, int32_t l, int32_t n_cnt, double *cnt) { return psmc_hip_post_counts(((hip_be *)self)->ctx, seg, cnt1, l, n_cnt, cnt); }
static const char *hb_error(void *self) { return psmc_hip_last_error(((hip_be *)self)->ctx); }
static void hb_destroy(void *self) { hip_be *h = (hip_be *)self; psmc_hip_destroy(h->ctx); free(h->chk); }

int main(int argc, char *argv[])
{
	psmc_options o;
	psmc_options_default(&o);
	if (psmc_options_parse(&o, argc, argv)) { psmc_options_free(&o); return 1; }
	/* number of states is known only after the pattern (or -i file) is read: peek at it */
	psmc_pattern pat;
	int n_states = 0;
	if (o.param_file) {
		FILE *fp = fopen(o.param_file, "r"); char str[256];
		if (fp && fscanf(fp, "%255s", str) == 1 && psmc_pattern_parse(str, &pat) == 0) { n_states = pat.n_states; psmc_pattern_free(&pat); }
		if (fp) fclose(fp);
	} else if (psmc_pattern_parse(o.pattern_text ? o.pattern_text : "4+5*3+4", &pat) == 0) { n_states = pat.n_states; psmc_pattern_free(&pat); }
	if (n_states < 1) { fprintf(stderr, "psmc: malformed pattern\n"); return 1; }
	const char *mode_s = getenv("PSMC_HIP_MODE"), *dev_s = getenv("PSMC_HIP_DEVICE");
	int mode = (mode_s && strcmp(mode_s, "fast") == 0) ? PSMC_HIP_MODE_FAST : PSMC_HIP_MODE_EXACT;
	if ((o.decode || o.print_prob || o.cnt_file) && mode == PSMC_HIP_MODE_FAST) {
		fprintf(stderr, "psmc: decoding needs the exact forward/backward tables; using PSMC_HIP_MODE=exact\n");
		mode = PSMC_HIP_MODE_EXACT;
	}
	{ /* the O(N) objective goes with the fast E-step unless asked otherwise */
		const char *fm = getenv("PSMC_FAST_MSTEP");
		o.fast_mstep = fm ? atoi(fm) != 0 : (mode == PSMC_HIP_MODE_FAST);
	}
	hip_be h;
	memset(&h, 0, sizeof h);
	h.n_states = n_states;
	int rc = psmc_hip_create(&h.ctx, n_states, dev_s ? atoi(dev_s) : 0, mode);
	if (rc) {
		fprintf(stderr, "psmc: cannot start the MI355X E-step (%s); this build has no CPU path\n", psmc_hip_strerror(rc));
		psmc_options_free(&o);
		return 2;
	}
	/* the factored E-step goes with the O(N) objective: fast mode, n <= 64 (PSMC_FACTORED=0 keeps the full counts) */
	const char *fs = getenv("PSMC_FACTORED");
	const int use_factored = o.fast_mstep && mode == PSMC_HIP_MODE_FAST && n_states <= 64 && !(fs && atoi(fs) == 0);
	psmc_estep_backend be = {&h, hb_load, hb_estep, hb_tables, hb_decode, use_factored ? hb_estep_factored : 0, hb_error, hb_destroy, hb_posterior, hb_post_counts};
	int status = psmc_run(&o, &be);
	be.destroy(be.self);
	psmc_options_free(&o);
	return status;
}
