/* main.c -- the `psmc` executable: lh3/psmc's command line (main.c:6-34) with
 * the E-step bound to libpsmc_hip.so.  There is no CPU E-step in this binary:
 * without a visible AMD GPU it exits with an error.
 *   PSMC_HIP_MODE=exact (default: .psmc byte-identical to the reference) | fast
 *   PSMC_HIP_DEVICE=<index>          one GPU
 *   PSMC_HIP_DEVICES=<i>,<j>,...     the segments of every E-step sharded over these GPUs (psmc_hip_group_*: LPT
 *                                    partition, one RCCL all-reduce of the statistics per EM iteration in fast mode,
 *                                    ordered per-segment sum in exact mode -- the output does not depend on the list) */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "psmc_host.h"
#include "psmc_hip.h"

typedef struct { psmc_hip_ctx *ctx; psmc_hip_group *grp; int n_seg, n_states; double *chk; char msg[256]; } hip_be;

/* the per-segment readers go to the context that holds the segment's tables */
static int route(hip_be *h, int seg, psmc_hip_ctx **c, int *local)
{
	if (!h->grp) { *c = h->ctx; *local = seg; return 0; }
	return psmc_hip_group_route(h->grp, seg, c, local);
}

static int hb_load(void *self, int n_seg, const uint8_t *const *sym, const int32_t *L)
{
	hip_be *h = (hip_be *)self;
	h->n_seg = n_seg;
	h->chk = (double *)realloc(h->chk, sizeof(double) * (size_t)n_seg);
	return h->grp ? psmc_hip_group_load_segments(h->grp, n_seg, sym, L) : psmc_hip_load_segments(h->ctx, n_seg, sym, L);
}
static int hb_estep(void *self, const double *a, const double *e, const double *a0, double *A, double *E, double *LL,
                    double *chk)
{
	hip_be *h = (hip_be *)self;
	int rc = h->grp ? psmc_hip_group_estep(h->grp, a, e, a0, A, E, 0, LL, h->chk) : psmc_hip_estep(h->ctx, a, e, a0, A, E, 0, LL, h->chk);
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
	int rc = h->grp ? psmc_hip_group_estep_factored(h->grp, a, e, a0, sums, E, LL) : psmc_hip_estep_factored(h->ctx, a, e, a0, sums, E, LL);
	if (rc != PSMC_HIP_ENOTSUP) return rc;
	/* a matrix without the two rank-1 triangles (e.g. -C): full counts, triangular sums on the host */
	const int n = h->n_states;
	double *A = (double *)calloc((size_t)n * n, sizeof(double));
	rc = h->grp ? psmc_hip_group_estep(h->grp, a, e, a0, A, E, 0, LL, h->chk) : psmc_hip_estep(h->ctx, a, e, a0, A, E, 0, LL, h->chk);
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
static int hb_tables(void *self, int seg, double *f, double *b, double *s)
{
	psmc_hip_ctx *c; int l; int rc = route((hip_be *)self, seg, &c, &l);
	return rc ? rc : psmc_hip_get_tables(c, l, f, b, s);
}
static int hb_decode(void *self, int seg, int32_t *path, double *maxp)
{
	psmc_hip_ctx *c; int l; int rc = route((hip_be *)self, seg, &c, &l);
	return rc ? rc : psmc_hip_decode(c, l, path, maxp);
}
static int hb_posterior(void *self, int seg, double *post, double *recomb)
{
	psmc_hip_ctx *c; int l; int rc = route((hip_be *)self, seg, &c, &l);
	return rc ? rc : psmc_hip_posterior(c, l, post, recomb);
}
static int hb_post_counts(void *self, int seg, const int32_t *cnt1, int32_t l1, int32_t n_cnt, double *cnt)
{
	psmc_hip_ctx *c; int l; int rc = route((hip_be *)self, seg, &c, &l);
	return rc ? rc : psmc_hip_post_counts(c, l, cnt1, l1, n_cnt, cnt);
}
static const char *hb_error(void *self) { hip_be *h = (hip_be *)self; return h->grp ? psmc_hip_group_last_error(h->grp) : psmc_hip_last_error(h->ctx); }
static void hb_destroy(void *self) { hip_be *h = (hip_be *)self; if (h->grp) psmc_hip_group_destroy(h->grp); else psmc_hip_destroy(h->ctx); free(h->chk); }

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
	if (n_states > 128) { /* the reference has no such limit (khmm.c:10-23 allocates for any n); this build's kernels keep one or two states per lane */
		fprintf(stderr, "psmc: the pattern gives %d hidden states; this MI355X build supports at most 128 (e.g. -p \"64*2\"). "
		        "Use a coarser pattern, or the reference binary for this run.\n", n_states);
		psmc_options_free(&o);
		return 2;
	}
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
	int rc;
	const char *devs = getenv("PSMC_HIP_DEVICES");
	if (devs && strchr(devs, ',')) { /* several devices: shard every E-step */
		int list[64], n_list = 0;
		char *dup = strdup(devs);
		for (char *t = strtok(dup, ","); t && n_list < 64; t = strtok(0, ",")) list[n_list++] = atoi(t);
		free(dup);
		rc = psmc_hip_group_create(&h.grp, n_states, n_list, list, mode);
		const char *rc_s = getenv("PSMC_HIP_RCCL");
		if (rc == 0 && rc_s) rc = psmc_hip_group_set_option(h.grp, "rccl", atof(rc_s));
	} else rc = psmc_hip_create(&h.ctx, n_states, devs && *devs ? atoi(devs) : (dev_s ? atoi(dev_s) : 0), mode);
	if (rc) {
		fprintf(stderr, "psmc: cannot start the MI355X E-step (%s); this build has no CPU path\n", psmc_hip_strerror(rc));
		psmc_options_free(&o);
		return 2;
	}
	/* the factored E-step goes with the O(N) objective: fast mode (PSMC_FACTORED=0 keeps the full counts) */
	const char *fs = getenv("PSMC_FACTORED");
	const int use_factored = o.fast_mstep && mode == PSMC_HIP_MODE_FAST && n_states <= 128 && !(fs && atoi(fs) == 0);
	psmc_estep_backend be = {&h, hb_load, hb_estep, hb_tables, hb_decode, use_factored ? hb_estep_factored : 0, hb_error, hb_destroy, hb_posterior, hb_post_counts};
	int status = psmc_run(&o, &be);
	be.destroy(be.self);
	psmc_options_free(&o);
	return status;
}
