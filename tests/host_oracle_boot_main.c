/* host_oracle_boot_main.c -- TEST INFRASTRUCTURE.  The bootstrap driver (psmc_amd/host/boot.c) with the CPU oracle
 * injected as batch E-step backend (two pretend devices), so that its replicate bookkeeping -- seeds, resampling
 * draws, batch packing, M-steps on threads, per-replicate output streams -- can be checked on a machine without a GPU
 * against single `psmc -b` runs of the oracle-backed binary (tests/host_oracle_main.c).  Never shipped. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "psmc_host.h"
#include "psmc_oracle.h"
/* the single-run oracle backend (orc_be, ob_*) for the --main run: the other test main, its main() renamed away */
#define main host_oracle_single_main
#include "host_oracle_main.c"
#undef main

typedef struct { int n, n_seg; const uint8_t **sym; int32_t *L; } orc_bb;

static int obb_load(void *self, int dev, int n_seg, const uint8_t *const *sym, const int32_t *L)
{
	orc_bb *o = (orc_bb *)self;
	if (dev != 0) return 0; /* both pretend devices read the same host copy */
	o->n_seg = n_seg;
	o->sym = (const uint8_t **)malloc(sizeof(void *) * n_seg);
	o->L = (int32_t *)malloc(sizeof(int32_t) * n_seg);
	for (int i = 0; i < n_seg; ++i) { o->sym[i] = sym[i]; o->L[i] = L[i]; }
	return 0;
}
static int obb_estep_batch(void *self, int dev, int n_rep, const double *a, const double *e, const double *a0,
                          const int32_t *sel_off, const int32_t *sel_idx, double *A, double *sums, double *E, double *LL,
                          void (*done)(void *user, int n_done, const int32_t *pos), void *user)
{
	orc_bb *o = (orc_bb *)self;
	const int n = o->n;
	(void)dev;
	for (int r = 0; r < n_rep; ++r) {
		const int ns = sel_off[r + 1] - sel_off[r];
		const uint8_t **sq = (const uint8_t **)malloc(sizeof(void *) * ns);
		int32_t *ln = (int32_t *)malloc(sizeof(int32_t) * ns);
		for (int i = 0; i < ns; ++i) { sq[i] = o->sym[sel_idx[sel_off[r] + i]]; ln[i] = o->L[sel_idx[sel_off[r] + i]]; }
		double *e3 = (double *)malloc(sizeof(double) * 3 * n), *Ar = (double *)calloc((size_t)n * n, sizeof(double));
		memcpy(e3, e + (size_t)r * 2 * n, sizeof(double) * 2 * n);
		for (int k = 0; k < n; ++k) e3[2 * n + k] = 1.0; /* khmm.c:21 */
		orc_estep(n, a + (size_t)r * n * n, e3, a0 + (size_t)r * n, ns, sq, ln, Ar, E + (size_t)r * 2 * n, 0, LL + r, 0, 0, 0, 0);
		if (A) memcpy(A + (size_t)r * n * n, Ar, sizeof(double) * n * n);
		if (sums) {
			double *q = sums + (size_t)r * 5 * n;
			memset(q, 0, sizeof(double) * 5 * n);
			for (int k = 0; k < n; ++k)
				for (int l = 0; l < n; ++l) {
					const double v = Ar[(size_t)k * n + l];
					if (l < k) { q[k] += v; q[3 * n + l] += v; } else if (l > k) { q[n + k] += v; q[4 * n + l] += v; } else q[2 * n + k] = v;
				}
		}
		free(sq); free(ln); free(e3); free(Ar);
		{ const int32_t rr = r; done(user, 1, &rr); } /* final: its M-step may start */
	}
	return 0;
}
static const char *obb_error(void *self, int dev) { return "oracle batch backend"; }
static void obb_destroy(void *self)
{
	orc_bb *o = (orc_bb *)self;
	free((void *)o->sym); free(o->L); o->sym = 0; o->L = 0;
}

int main(int argc, char **argv)
{	/* host_oracle_boot R SEED PATTERN [--main OUT IN] <psmc options> input */
	if (argc < 5) return 1;
	const int n_rep = atoi(argv[1]);
	const long seed0 = atol(argv[2]);
	const char *pattern = argv[3], *main_out = 0, *main_in = 0;
	if (!strcmp(argv[4], "--main") && argc >= 8) { main_out = argv[5]; main_in = argv[6]; argv += 3; argc -= 3; }
	psmc_options o, om;
	psmc_options_default(&o); psmc_options_default(&om);
	argv[3] = argv[0];
	if (psmc_options_parse(&o, argc - 3, argv + 3)) return 1;
	if (main_out && psmc_options_parse(&om, argc - 3, argv + 3)) return 1;
	o.bootstrap = 1;
	o.fast_mstep = getenv("PSMC_FAST_MSTEP") && atoi(getenv("PSMC_FAST_MSTEP")) != 0;
	psmc_pattern pat;
	if (psmc_pattern_parse(o.pattern_text ? o.pattern_text : "4+5*3+4", &pat)) return 1;
	orc_bb ob; memset(&ob, 0, sizeof ob);
	ob.n = pat.n_states;
	psmc_batch_backend bb = {&ob, 2, obb_load, obb_estep_batch, obb_error, obb_destroy, o.fast_mstep, 0, 0};
	/* --main: the un-resampled run on its own input beside the replicates (boot_main.c does the same with the HIP backend) */
	orc_be om_be; memset(&om_be, 0, sizeof om_be);
	psmc_estep_backend be_main = {&om_be, ob_load, ob_estep, ob_tables, 0, 0, ob_error, ob_destroy, 0, 0};
	psmc_run_state *main_run = 0;
	if (main_out) {
		om.bootstrap = 0; om.fast_mstep = o.fast_mstep;
		free(om.in_file); om.in_file = strdup(main_in);
		free(om.out_file); om.out_file = strdup(main_out);
		om_be.n = pat.n_states;
		om_be.a = (double *)malloc(sizeof(double) * om_be.n * om_be.n); om_be.e = (double *)malloc(sizeof(double) * 3 * om_be.n); om_be.a0 = (double *)malloc(sizeof(double) * om_be.n);
		if (o.fast_mstep && getenv("PSMC_FACTORED") && atoi(getenv("PSMC_FACTORED")) != 0) be_main.estep_factored = ob_estep_factored;
		main_run = psmc_run_begin(&om, &be_main);
		if (!main_run) return 1;
	}
	const int status = psmc_boot_run(&o, n_rep, seed0, pattern, &bb, main_run);
	if (main_out) be_main.destroy(be_main.self);
	bb.destroy(bb.self);
	psmc_pattern_free(&pat);
	psmc_options_free(&o);
	return status;
}
