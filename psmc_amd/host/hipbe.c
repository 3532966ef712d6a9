/* hipbe.c -- psmc_estep_backend over libpsmc_hip.so: one context, or a group of devices (hipbe.h). */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "hipbe.h"

typedef struct {
	psmc_hip_ctx *ctx; psmc_hip_group *grp; int n_seg, n_states; double *chk; char msg[256];
	/* what a fast-mode run needs to repeat ONE E-step in exact mode when its tile boundaries do not converge (PSMC_HIP_ECONVERGE):
	 * how it was created, the segments (borrowed from the caller's psmc_input, alive for the whole run), and the exact twin once made */
	int mode, device; char *devs;
	const uint8_t **sym; int32_t *L;
	psmc_hip_ctx *x_ctx; psmc_hip_group *x_grp; int n_fallbacks;
} hip_be;

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
	h->sym = (const uint8_t **)realloc((void *)h->sym, sizeof(void *) * (size_t)n_seg);
	h->L = (int32_t *)realloc(h->L, sizeof(int32_t) * (size_t)n_seg);
	for (int i = 0; i < n_seg; ++i) { h->sym[i] = sym[i]; h->L[i] = L[i]; }
	if (h->x_ctx) { psmc_hip_destroy(h->x_ctx); h->x_ctx = 0; }
	if (h->x_grp) { psmc_hip_group_destroy(h->x_grp); h->x_grp = 0; }
	return h->grp ? psmc_hip_group_load_segments(h->grp, n_seg, sym, L) : psmc_hip_load_segments(h->ctx, n_seg, sym, L);
}
/* PSMC_HIP_ECONVERGE from a fast E-step (the verify / repair rounds of the speculative tiles ran out: an input whose chain forgets
 * more slowly than anything the plan allows for) used to end the run.  The statistics are a well-defined quantity all the same:
 * this ONE E-step is repeated by the exact kernels -- a twin context / group in exact mode over the same segments, made on first
 * need -- with a note on stderr; the run goes on in fast mode.  (khmm.c:145-324 has no such failure mode.) */
static int exact_once(hip_be *h, const double *a, const double *e, const double *a0, double *A, double *E, double *LL)
{
	int rc = 0;
	fprintf(stderr, "psmc: fast E-step did not converge (%s); repeating this E-step with the exact kernels\n",
	        h->grp ? psmc_hip_group_last_error(h->grp) : psmc_hip_last_error(h->ctx));
	if (!h->x_ctx && !h->x_grp) {
		if (h->grp) {
			int list[64], n_list = 0;
			char *dup = strdup(h->devs ? h->devs : "0");
			for (char *t = strtok(dup, ","); t && n_list < 64; t = strtok(0, ",")) list[n_list++] = atoi(t);
			free(dup);
			rc = psmc_hip_group_create(&h->x_grp, h->n_states, n_list, list, PSMC_HIP_MODE_EXACT);
			if (rc == 0) rc = psmc_hip_group_load_segments(h->x_grp, h->n_seg, h->sym, h->L);
		} else {
			rc = psmc_hip_create(&h->x_ctx, h->n_states, h->device, PSMC_HIP_MODE_EXACT);
			if (rc == 0) rc = psmc_hip_load_segments(h->x_ctx, h->n_seg, h->sym, h->L);
		}
		if (rc) { fprintf(stderr, "psmc: cannot set up the exact E-step (%s)\n", psmc_hip_strerror(rc)); return rc; }
	}
	++h->n_fallbacks;
	return h->x_grp ? psmc_hip_group_estep(h->x_grp, a, e, a0, A, E, 0, LL, h->chk) : psmc_hip_estep(h->x_ctx, a, e, a0, A, E, 0, LL, h->chk);
}

static void tri_sums(int n, const double *A, double *sums)
{
	memset(sums, 0, sizeof(double) * 5 * (size_t)n);
	for (int k = 0; k < n; ++k)
		for (int l = 0; l < n; ++l) {
			const double v = A[(size_t)k * n + l];
			if (l < k) { sums[k] += v; sums[3 * n + l] += v; } else if (l > k) { sums[n + k] += v; sums[4 * n + l] += v; } else sums[2 * n + k] = v;
		}
}

static int hb_estep(void *self, const double *a, const double *e, const double *a0, double *A, double *E, double *LL,
                    double *chk)
{
	hip_be *h = (hip_be *)self;
	int rc = h->grp ? psmc_hip_group_estep(h->grp, a, e, a0, A, E, 0, LL, h->chk) : psmc_hip_estep(h->ctx, a, e, a0, A, E, 0, LL, h->chk);
	if (rc == PSMC_HIP_ECONVERGE && h->mode == PSMC_HIP_MODE_FAST) rc = exact_once(h, a, e, a0, A, E, LL);
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
	if (rc != PSMC_HIP_ENOTSUP && rc != PSMC_HIP_ECONVERGE) return rc;
	/* a matrix without the two rank-1 triangles (e.g. -C): full counts, triangular sums on the host; tile boundaries that
	 * did not converge: the exact kernels once (exact_once), sums of their A */
	const int n = h->n_states;
	double *A = (double *)calloc((size_t)n * n, sizeof(double));
	if (rc == PSMC_HIP_ECONVERGE) rc = exact_once(h, a, e, a0, A, E, LL);
	else rc = hb_estep(self, a, e, a0, A, E, LL, 0);
	if (rc == 0) tri_sums(n, A, sums);
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
static void hb_destroy(void *self)
{
	hip_be *h = (hip_be *)self;
	if (h->x_ctx) psmc_hip_destroy(h->x_ctx);
	if (h->x_grp) psmc_hip_group_destroy(h->x_grp);
	if (h->grp) psmc_hip_group_destroy(h->grp); else psmc_hip_destroy(h->ctx);
	free(h->chk); free((void *)h->sym); free(h->L); free(h->devs); free(h);
}


int psmc_hipbe_create(psmc_estep_backend *be, int n_states, int mode, int use_factored, const char *devs, int device)
{
	hip_be *h = (hip_be *)calloc(1, sizeof(hip_be));
	int rc;
	memset(be, 0, sizeof *be);
	if (!h) return PSMC_HIP_ENOMEM;
	h->n_states = n_states; h->mode = mode;
	h->device = devs && *devs && !strchr(devs, ',') ? atoi(devs) : device;
	h->devs = devs ? strdup(devs) : 0;
	if (devs && strchr(devs, ',')) { /* several devices: shard every E-step */
		int list[64], n_list = 0;
		char *dup = strdup(devs);
		for (char *t = strtok(dup, ","); t && n_list < 64; t = strtok(0, ",")) list[n_list++] = atoi(t);
		free(dup);
		rc = psmc_hip_group_create(&h->grp, n_states, n_list, list, mode);
		const char *rc_s = getenv("PSMC_HIP_RCCL");
		if (rc == 0 && rc_s) rc = psmc_hip_group_set_option(h->grp, "rccl", atof(rc_s));
		if (rc == 0) {
			/* first contact with the devices, before a segment is loaded: every listed device answers, the exchange the
			 * E-steps will use comes up and adds correctly in stream order -- a failure names its step here, not in the first E-step */
			int sc[4] = {0, 0, 0, 0};
			static const char *path[] = {"single shard", "RCCL all-reduce", "host sum of the shards' vectors", "exact mode: ordered per-segment sum, nothing to exchange"};
			rc = psmc_hip_group_selfcheck(h->grp, sc);
			if (rc) fprintf(stderr, "psmc: devices %s: self-check failed: %s\n", devs, psmc_hip_group_last_error(h->grp));
			else {
				if (sc[3]) fprintf(stderr, "psmc: devices %s: RCCL wanted but not usable (%s); the host adds the shards' vectors\n", devs, psmc_hip_group_last_error(h->grp));
				if (getenv("PSMC_TIMING")) fprintf(stderr, "[timing] devices %s: self-check ok, %d shards, exchange: %s%s\n", devs, sc[0], path[sc[1] >= 0 && sc[1] <= 3 ? sc[1] : 0], sc[2] ? " (communicator up)" : "");
			}
		}
	} else rc = psmc_hip_create(&h->ctx, n_states, devs && *devs ? atoi(devs) : device, mode);
	if (rc) { if (h->grp) psmc_hip_group_destroy(h->grp); free(h->devs); free(h); return rc; }
	be->self = h; be->load = hb_load; be->estep = hb_estep; be->tables = hb_tables; be->decode = hb_decode;
	be->estep_factored = use_factored ? hb_estep_factored : 0; be->error = hb_error; be->destroy = hb_destroy;
	be->posterior = hb_posterior; be->post_counts = hb_post_counts;
	return 0;
}

psmc_hip_ctx *psmc_hipbe_ctx(psmc_estep_backend *be) { return be && be->self ? ((hip_be *)be->self)->ctx : 0; }
