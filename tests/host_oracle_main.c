/* host_oracle_main.c -- TEST INFRASTRUCTURE.  The host driver (psmc_amd/host)
 * with the CPU oracle injected as E-step backend, so that the host logic
 * (command line, reader, model, M-step, writer, decoding output) can be checked
 * byte for byte against the reference's golden .psmc files on a machine without
 * a GPU.  Built by tests/test_host_cli.py; never shipped: the product binary
 * (psmc_amd/host/psmc) has no such backend. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "psmc_host.h"
#include "psmc_oracle.h"

typedef struct { int n, n_seg; const uint8_t **sym; int32_t *L; double *a, *e, *a0; } orc_be;

static int ob_load(void *self, int n_seg, const uint8_t *const *sym, const int32_t *L)
{
	orc_be *o = (orc_be *)self;
	o->n_seg = n_seg;
	o->sym = (const uint8_t **)malloc(sizeof(void *) * n_seg);
	o->L = (int32_t *)malloc(sizeof(int32_t) * n_seg);
	for (int i = 0; i < n_seg; ++i) { o->sym[i] = sym[i]; o->L[i] = L[i]; }
	return 0;
}
static int ob_estep(void *self, const double *a, const double *e, const double *a0, double *A, double *E, double *LL, double *chk)
{
	orc_be *o = (orc_be *)self;
	const int n = o->n;
	memcpy(o->a, a, sizeof(double) * n * n); memcpy(o->e, e, sizeof(double) * 3 * n); memcpy(o->a0, a0, sizeof(double) * n);
	double *c = (double *)malloc(sizeof(double) * o->n_seg);
	orc_estep(n, a, e, a0, o->n_seg, o->sym, o->L, A, E, 0, LL, 0, 0, 0, c);
	for (int i = 0; i < o->n_seg; ++i)
		if (c[i] > 1.0 + 1e-6 || c[i] < 1.0 - 1e-6) fprintf(stderr, "++ Underflow may have happened (%lg).\n", c[i]);
	if (chk) memcpy(chk, c, sizeof(double) * o->n_seg);
	free(c);
	return 0;
}
/* the factored E-step of the HIP backend, restated with the oracle: triangular sums of its A */
static int ob_estep_factored(void *self, const double *a, const double *e, const double *a0, double *sums, double *E, double *LL)
{
	orc_be *o = (orc_be *)self;
	const int n = o->n;
	double *A = (double *)calloc((size_t)n * n, sizeof(double));
	int rc = ob_estep(self, a, e, a0, A, E, LL, 0);
	memset(sums, 0, sizeof(double) * 5 * (size_t)n);
	for (int k = 0; k < n; ++k)
		for (int l = 0; l < n; ++l) {
			const double v = A[(size_t)k * n + l];
			if (l < k) { sums[k] += v; sums[3 * n + l] += v; } else if (l > k) { sums[n + k] += v; sums[4 * n + l] += v; } else sums[2 * n + k] = v;
		}
	free(A);
	return rc;
}
static int ob_tables(void *self, int seg, double *f, double *b, double *s)
{
	orc_be *o = (orc_be *)self;
	const int n = o->n, L = o->L[seg];
	double *ae = (double *)malloc(sizeof(double) * 3 * n * n);
	double *ff = (double *)malloc(sizeof(double) * (size_t)(L + 1) * n), *bb = (double *)malloc(sizeof(double) * (size_t)(L + 1) * n);
	double *ss = (double *)malloc(sizeof(double) * (size_t)(L + 1));
	orc_pre_backward(n, o->a, o->e, ae);
	orc_forward(n, o->a, o->e, o->a0, L, o->sym[seg], ff, ss);
	orc_backward(n, ae, o->e, o->a0, L, o->sym[seg], ss, bb);
	if (f) memcpy(f, ff + n, sizeof(double) * (size_t)L * n);
	if (b) memcpy(b, bb + n, sizeof(double) * (size_t)L * n);
	if (s) memcpy(s, ss + 1, sizeof(double) * (size_t)L);
	free(ae); free(ff); free(bb); free(ss);
	return 0;
}
static const char *ob_error(void *self) { return "oracle backend"; }
static void ob_destroy(void *self)
{
	orc_be *o = (orc_be *)self;
	free((void *)o->sym); free(o->L); free(o->a); free(o->e); free(o->a0);
	o->sym = 0; o->L = 0; o->a = o->e = o->a0 = 0;
}

int main(int argc, char **argv)
{
	psmc_options o;
	psmc_options_default(&o);
	if (psmc_options_parse(&o, argc, argv)) return 1;
	o.fast_mstep = getenv("PSMC_FAST_MSTEP") && atoi(getenv("PSMC_FAST_MSTEP")) != 0; /* O(N) objective (what PSMC_HIP_MODE=fast uses) */
	psmc_pattern pat;
	char str[256] = "4+5*3+4";
	if (o.param_file) { FILE *fp = fopen(o.param_file, "r"); if (!fp || fscanf(fp, "%255s", str) != 1) return 1; fclose(fp); }
	else if (o.pattern_text) snprintf(str, sizeof str, "%s", o.pattern_text);
	if (psmc_pattern_parse(str, &pat)) return 1;
	orc_be ob; memset(&ob, 0, sizeof ob);
	ob.n = pat.n_states;
	ob.a = (double *)malloc(sizeof(double) * ob.n * ob.n); ob.e = (double *)malloc(sizeof(double) * 3 * ob.n); ob.a0 = (double *)malloc(sizeof(double) * ob.n);
	const int fac = o.fast_mstep && getenv("PSMC_FACTORED") && atoi(getenv("PSMC_FACTORED")) != 0;
	psmc_estep_backend be = {&ob, ob_load, ob_estep, ob_tables, 0, fac ? ob_estep_factored : 0, ob_error, ob_destroy, 0, 0};
	const int status = psmc_run(&o, &be);
	be.destroy(be.self);
	psmc_pattern_free(&pat);
	psmc_options_free(&o);
	return status;
}
