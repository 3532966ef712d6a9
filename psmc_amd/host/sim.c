/* sim.c -- seeded synthetic observation streams for benchmarks: a hidden state
 * path from (a0, a), a het/hom emission per bin from e, and short runs of
 * missing bins -- what hmm_simulate (khmm.c:386-423) does for the reference's
 * -S, with our own splitmix64/xoshiro RNG so that it is reproducible anywhere. */
#include <stdlib.h>
#include "psmc_host.h"

static uint64_t splitmix(uint64_t *s) { uint64_t z = (*s += 0x9e3779b97f4a7c15ULL); z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL; z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL; return z ^ (z >> 31); }
static double unif(uint64_t *s) { return (double)(splitmix(s) >> 11) * (1.0 / 9007199254740992.0); }

void psmc_simulate_segment(int n, const double *a, const double *e, const double *a0, int32_t L, uint64_t seed,
                           double miss_rate, uint8_t *out)
{
	uint64_t st = seed * 0x2545f4914f6cdd1dULL + 1;
	double *cum = (double *)malloc(sizeof(double) * (size_t)n * n);
	for (int k = 0; k < n; ++k) { double y = 0.0; for (int l = 0; l < n; ++l) { y += a[(size_t)k * n + l]; cum[(size_t)k * n + l] = y; } }
	int k = 0; { double x = unif(&st), y = 0.0; for (k = 0; k < n - 1; ++k) { y += a0[k]; if (y >= x) break; } }
	int32_t miss_left = 0;
	for (int32_t i = 0; i < L; ++i) {
		const double x = unif(&st), *c = cum + (size_t)k * n;
		int lo = 0, hi = n - 1; /* first l with cum[l] >= x */
		if (c[k] >= x && (k == 0 || c[k - 1] < x)) lo = hi = k; /* staying put is by far the common case */
		while (lo < hi) { const int mid = (lo + hi) >> 1; if (c[mid] >= x) hi = mid; else lo = mid + 1; }
		k = lo;
		uint8_t v = unif(&st) < e[(size_t)n + k] ? 1 : 0;
		if (miss_left > 0) { v = 2; --miss_left; }
		else if (miss_rate > 0 && unif(&st) < miss_rate) { miss_left = 10 + (int32_t)(unif(&st) * 81); v = 2; --miss_left; }
		out[i] = v;
	}
	free(cum);
}
