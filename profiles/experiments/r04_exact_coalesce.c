/* does the EXACT forward vector (khmm.c:177-186 operation order: products summed in l order, s = sum in
 * k order, true division), started `W` bins early from the initial distribution a0, become BITWISE equal to the true trajectory
 * -- and after how many bins?  If it did, exact mode could tile a segment like fast mode (speculate, verify bitwise, repair).
 * Own code (a diagnostic, not product, not the oracle).  Input: a binary file written by exact_coalesce.py:
 *   int32 n, L, n_rounds, n_starts, W, H;  uint8 seq[L];  per round: double a[n*n], e[3*n], a0[n];  int32 starts[n_starts]
 * Output per round: how many starts coalesce within H bins after the tile start, and the quantiles of the coalescence bin. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>

static void step(int n, const double *a /* a[l*n+k] */, const double *e, const double *f1, double *f, double *s)
{
	double tmp[256];
	for (int k = 0; k < n; ++k) tmp[k] = 0.0;
	for (int l = 0; l < n; ++l) { /* per k the same order of additions as `for l: tmp += fu1[l]*aa[l]` */
		const double fl = f1[l], *al = a + (size_t)l * n;
		for (int k = 0; k < n; ++k) tmp[k] += fl * al[k];
	}
	double sum = 0.0;
	for (int k = 0; k < n; ++k) sum += (f[k] = e[k] * tmp[k]);
	for (int k = 0; k < n; ++k) f[k] /= sum;
	*s = sum;
}
static int cmp_int(const void *x, const void *y) { return *(const int *)x - *(const int *)y; }

int main(int argc, char **argv)
{
	if (argc < 2) return 2;
	FILE *fp = fopen(argv[1], "rb");
	if (!fp) return 2;
	int32_t hd[6];
	if (fread(hd, 4, 6, fp) != 6) return 2;
	const int n = hd[0], L = hd[1], R = hd[2], NS = hd[3], W = hd[4], H = hd[5];
	uint8_t *seq = malloc(L);
	if (fread(seq, 1, L, fp) != (size_t)L) return 2;
	double *par = malloc(sizeof(double) * R * (n * n + 4 * n));
	if (fread(par, 8, (size_t)R * (n * n + 4 * n), fp) != (size_t)R * (n * n + 4 * n)) return 2;
	int32_t *starts = malloc(4 * NS);
	if (fread(starts, 4, NS, fp) != (size_t)NS) return 2;
	double *F = malloc(sizeof(double) * (size_t)L * n), s;
	int *when = malloc(sizeof(int) * NS);
	printf("# n=%d L=%d rounds=%d starts=%d warm-up=%d horizon=%d\n", n, L, R, NS, W, H);
	printf("# round coalesced_by_tile_start coalesced_within_horizon median p90 max_coalesced  ulp_components_at_horizon(min/median/max over the others)\n");
	long tot_ok = 0, tot = 0;
	for (int r = 0; r < R; ++r) {
		const double *a = par + (size_t)r * (n * n + 4 * n), *e = a + n * n, *a0 = e + 3 * n;
		/* true trajectory, khmm.c:171-186 */
		double sum = 0.0;
		for (int k = 0; k < n; ++k) sum += (F[k] = a0[k] * e[seq[0] * n + k]);
		for (int k = 0; k < n; ++k) F[k] /= sum;
		for (int u = 1; u < L; ++u) step(n, a, e + seq[u] * n, F + (size_t)(u - 1) * n, F + (size_t)u * n, &s);
		int n_at0 = 0, n_ok = 0, nd[4096], ndn = 0;
		for (int i = 0; i < NS; ++i) {
			const int t0 = starts[i], u0 = t0 - W; /* the speculation starts at bin u0 as if it were a segment start */
			double x[256], y[256];
			sum = 0.0;
			for (int k = 0; k < n; ++k) sum += (x[k] = a0[k] * e[seq[u0] * n + k]);
			for (int k = 0; k < n; ++k) x[k] /= sum;
			when[i] = -1;
			int u, differ = n;
			for (u = u0 + 1; u < t0 + H && u < L; ++u) {
				step(n, a, e + seq[u] * n, x, y, &s);
				memcpy(x, y, sizeof(double) * n);
				differ = 0;
				for (int k = 0; k < n; ++k) differ += memcmp(&x[k], &F[(size_t)u * n + k], 8) != 0;
				if (differ == 0) { when[i] = u - t0; break; } /* deterministic: equal once, equal for ever */
			}
			if (when[i] >= -W && differ == 0) { ++n_ok; if (when[i] <= 0) ++n_at0; } else nd[ndn++] = differ;
		}
		int ok[4096], m = 0;
		for (int i = 0; i < NS; ++i) if (when[i] != -1 || 0) { if (when[i] > -W - 1 && when[i] != -1) ok[m++] = when[i]; }
		qsort(ok, m, sizeof(int), cmp_int); qsort(nd, ndn, sizeof(int), cmp_int);
		printf("%2d  %3d  %3d  %6d %6d %6d   %d/%d/%d\n", r + 1, n_at0, n_ok, m ? ok[m / 2] : -1, m ? ok[(int)(0.9 * (m - 1))] : -1, m ? ok[m - 1] : -1,
		       ndn ? nd[0] : 0, ndn ? nd[ndn / 2] : 0, ndn ? nd[ndn - 1] : 0);
		fflush(stdout);
		tot_ok += n_ok; tot += NS;
	}
	printf("# total: %ld of %ld starts coalesce bitwise within the horizon (%.1f %%)\n", tot_ok, tot, 100.0 * tot_ok / tot);
	return 0;
}
