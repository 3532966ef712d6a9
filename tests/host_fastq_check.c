/* host_fastq_check.c -- TEST INFRASTRUCTURE.  Extreme Hooke-Jeeves trial points through the scalar and the
 * SIMD log-factor routines of the fast M-step (psmc_amd/host/model.c, fastq.c): both must reject the same
 * points (return 0 -> objective +1e300, like hmm_Q's -HMM_INF, khmm.c:369-377) and agree where they accept.
 * fastq.c is built with -ffast-math, so its rejections must not depend on NaN-aware comparisons.
 * Every fourth point is an ordinary one (lambdas in 0.2 .. 3.2): there the two must agree to rounding.  At the
 * corrupted points both evaluate ill-conditioned expressions (alpha_k ~ 1e-300 ...), so only the accept / reject
 * decision and finiteness are compared.
 * Prints "n_points n_accept n_mismatch max_rel_diff_ordinary"; exit status 0 when n_mismatch == 0. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "psmc_host.h"

static unsigned long long rs = 88172645463325252ull;
static double rnd(void) { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (rs >> 11) * (1.0 / 9007199254740992.0); }

int main(int argc, char **argv)
{
	const char *ptxt = argc > 1 ? argv[1] : "4+25*2+4+6";
	psmc_pattern pat;
	if (psmc_pattern_parse(ptxt, &pat)) return 2;
	psmc_model *m = psmc_model_new(&pat, ptxt, 0.1, 0);
	const int N = pat.n_states, np = m->n_params;
	double *o1 = calloc(7 * N, 8), *o2 = calloc(7 * N, 8);
	const double special[] = {0.0, 1e-320, 1e-300, 1e-200, 1e-30, 1e-8, 1e-3, 1.0, 50.0, 1e8, 1e200, 1e308};
	int n_pts = 0, n_acc = 0, n_mis = 0;
	double maxrel = 0.0;
	for (int it = 0; it < 20000; ++it) {
		m->params[0] = 0.05; m->params[1] = 0.0125; m->params[2] = 15.0;
		for (int k = 3; k < np; ++k) m->params[k] = 0.2 + 3.0 * rnd();
		const int ordinary = it % 4 == 0;
		if (!ordinary) { /* corrupt one to three parameters with special values */
			const int nc = 1 + (int)(3 * rnd());
			for (int c = 0; c < nc; ++c) m->params[(int)(np * rnd())] = special[(int)(12 * rnd())];
		}
		const int r1 = psmc_model_logfactors(m, o1), r2 = psmc_model_logfactors_simd(m, o2);
		++n_pts;
		if (r1 != r2) { ++n_mis; if (n_mis < 5) { fprintf(stderr, "mismatch at it=%d: scalar %d simd %d; params", it, r1, r2); for (int k = 0; k < np; ++k) fprintf(stderr, " %g", m->params[k]); fprintf(stderr, "\n"); } continue; }
		if (!r1) continue;
		++n_acc;
		for (int i = 0; i < 7 * N; ++i) {
			if (!isfinite(o2[i])) { ++n_mis; break; }
			const double d = fabs(o1[i] - o2[i]) / (fabs(o1[i]) + 1e-3);
			if (ordinary && d > maxrel) { maxrel = d; if (getenv("FASTQ_CHECK_VERBOSE") && d > 1e-9) { fprintf(stderr, "it=%d i=%d (%s[%d]) scalar %.17g simd %.17g; params", it, i, (const char *[]){"lFL","lFU","lD","lqa","lc","le0","le1"}[i / N], i % N, o1[i], o2[i]); for (int k = 0; k < np; ++k) fprintf(stderr, " %g", m->params[k]); fprintf(stderr, "\n"); } }
		}
	}
	printf("%d %d %d %.3g\n", n_pts, n_acc, n_mis, maxrel);
	return n_mis ? 1 : 0;
}
