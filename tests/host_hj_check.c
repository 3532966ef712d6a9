/* host_hj_check.c -- TEST INFRASTRUCTURE.  psmc_hooke_jeeves (psmc_amd/host/mstep.c) on the objective oracle/ref_shim.c gives the
 * reference's kmin_hj (ref_kmin_quad): the same C expression, so the two searches can be compared to the last bit of every
 * coordinate (tests/test_host_cli.py).  Built as a shared object beside libpsmc_host.so. */
#include <math.h>
#include "psmc_host.h"

static double quad(int n, double *x, void *data)
{
	double s = 0.0, *c = (double *)data;
	for (int i = 0; i < n; ++i) s += (i + 1) * (x[i] - c[i]) * (x[i] - c[i]) + 0.1 * fabs(x[i]);
	return s;
}

double mine_kmin_quad(int n, double *x_io, double *centre, int max_calls)
{
	return psmc_hooke_jeeves(quad, n, x_io, centre, 0.5, 1e-7, max_calls); /* kmin.h:4-6: KMIN_RADIUS, KMIN_EPS */
}
