// selftest.hip -- on-device check of the cross-lane primitives the E-step
// kernels are built from.  The expected values come from plain LDS indexing, so
// a wrong assumption about a DPP / permlane / MFMA lane mapping shows up as a
// bit in the returned mask instead of as silently wrong statistics.
#include <hip/hip_runtime.h>
#include "wave_prims.h"
#include "struct_prims.h"
#include "psmc_hip_internal.h"

namespace psmc {

typedef double d4s_t __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(64) void k_selftest(unsigned *flags)
{
	__shared__ double sh[64];
	__shared__ double D[16][16];
	const int lane = threadIdx.x;
	const double x = 1.0 / (3.0 + lane) + 0.001 * lane * lane; // distinct, non-trivial mantissas
	sh[lane] = x;
	__syncthreads();
	unsigned bad = 0;
	double r0[4], r1[4];
	rep_rows_bperm(x, r0);
	rep_rows_swap(x, r1);
	for (int j = 0; j < 4; ++j) {
		const double want = sh[16 * j + (lane & 15)];
		if (r0[j] != want) bad |= 1u;
		if (r1[j] != want) bad |= 2u;
	}
	// from here on use LDS-built replicated registers so later checks are independent
	double r[4];
	for (int j = 0; j < 4; ++j) r[j] = sh[16 * j + (lane & 15)];
	if (bcast16<0>(x) != sh[(lane & ~15) + 0]) bad |= 4u;
	if (bcast16<5>(x) != sh[(lane & ~15) + 5]) bad |= 4u;
	if (bcast16<15>(x) != sh[(lane & ~15) + 15]) bad |= 4u;
	{
		double acc = 0.5;
		const double m = 2.0 + lane;
		dpp_guard(r);
		fmac_bcast<7>(acc, r[2], m); // fma(x[39], m, 0.5)
		if (acc != __builtin_fma(sh[39], m, 0.5)) bad |= 8u;
		double acc2 = 0.25;
		fmac_bcast<0>(acc2, r[0], m);
		fmac_bcast<15>(acc2, r[3], m);
		if (acc2 != __builtin_fma(sh[63], m, __builtin_fma(sh[0], m, 0.25))) bad |= 8u;
	}
	{
		double seq = 0.0;
		for (int k = 0; k < 64; ++k) seq = seq + sh[k];
		if (seq_sum_rep(r) != seq) bad |= 32u;
		const double t = wave_sum_rep(r);
		if (fabs(t - seq) > 1e-13 * fabs(seq)) bad |= 16u;
	}
	{ // exact dot against a serial loop
		double m[64];
		for (int l = 0; l < 64; ++l) m[l] = 1.0 / (1.0 + l + 0.37 * lane);
		double want = 0.0;
		for (int l = 0; l < 64; ++l) want = want + sh[l] * m[l];
		if (xdot64(r, m) != want) bad |= 64u;
		double wf = fdot64(r, m);
		if (fabs(wf - want) > 1e-13 * fabs(want)) bad |= 128u;
	}
	{ // natural-layout matvec and wave sum
		double Mreg[64];
		// M[l][k] = 1/(1 + l + 0.37 k): load the lane's 64 entries exactly like load_nat_matrix would
		const int rr = lane >> 4, mm = lane & 15;
		for (int j = 0; j < 4; ++j)
			for (int N = 0; N < 16; ++N) Mreg[16 * j + N] = 1.0 / (1.0 + (16 * rr + N) + 0.37 * (16 * j + mm));
		double want = 0.0;
		for (int l = 0; l < 64; ++l) want = __builtin_fma(sh[l], 1.0 / (1.0 + l + 0.37 * lane), want);
		const double got = matvec64_nat(x, Mreg);
		if (fabs(got - want) > 1e-13 * fabs(want)) bad |= 512u;
		double seq = 0.0;
		for (int k = 0; k < 64; ++k) seq = seq + sh[k];
		if (fabs(wave_sum_nat(x) - seq) > 1e-13 * fabs(seq)) bad |= 1024u;
	}
	{ // f64 MFMA 16x16x4 operand / result lane mapping
		const int t = lane >> 4, i = lane & 15;
		const double A = 0.5 * i + 3.0 * t + 1.0;   // A[i][t]
		const double B = 7.0 * i - 1.0 * t + 0.25;  // B[t][j=i]
		d4s_t acc = {0.0, 0.0, 0.0, 0.0};
		acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A, B, acc, 0, 0, 0);
		for (int q = 0; q < 4; ++q) D[t + 4 * q][i] = acc[q];
		__syncthreads();
		for (int e = lane; e < 256; e += 64) {
			const int ri = e >> 4, cj = e & 15;
			double want = 0.0;
			for (int tt = 0; tt < 4; ++tt) want += (0.5 * ri + 3.0 * tt + 1.0) * (7.0 * cj - 1.0 * tt + 0.25);
			if (fabs(D[ri][cj] - want) > 1e-9) bad |= 256u;
		}
	}
	{ // row-local exclusive scans (structured sweeps): lane m of a row sums the row's lanes above / below it
		const double es = row_excl_suffix(x), ep = row_excl_prefix(x);
		double ws = 0.0, wp = 0.0;
		for (int j = (lane & 15) + 1; j < 16; ++j) ws += sh[(lane & ~15) + j];
		for (int j = 0; j < (lane & 15); ++j) wp += sh[(lane & ~15) + j];
		if (fabs(es - ws) > 1e-13 * (fabs(ws) + 1e-300) || fabs(ep - wp) > 1e-13 * (fabs(wp) + 1e-300)) bad |= 2048u;
		double rs = 0.0;
		for (int j = 0; j < 16; ++j) rs += sh[(lane & ~15) + j];
		if (fabs(row_sum16(x) - rs) > 1e-13 * rs) bad |= 2048u;
	}
	{ // structured step == dense product with a[k][l] = P_k qa_l (l<k), R_k c_l (l>k), dd on the diagonal (+P.qa+R.c)
		__shared__ double sP[64], sR[64], sq[64], sc[64], sd[64], sx[4][64];
		sP[lane] = 0.01 + 0.003 * lane; sR[lane] = 0.02 / (1.0 + lane); sq[lane] = 0.5 + 0.01 * lane;
		sc[lane] = 1.0 / (3.0 + 0.2 * lane); sd[lane] = 0.9 - 0.002 * lane;
		for (int r = 0; r < 4; ++r) sx[r][lane] = 1.0 / (1.0 + ((lane * 7 + r * 13) % 64)) + 1e-3 * r; // 4 different vectors, one per row
		__syncthreads();
		const int row = lane >> 4, k0 = 4 * (lane & 15);
		StructPar c;
		double xv[4];
		for (int i = 0; i < 4; ++i) {
			c.mS[i] = sP[k0 + i]; c.wS[i] = sq[k0 + i]; c.mP[i] = sR[k0 + i]; c.wP[i] = sc[k0 + i]; c.dd[i] = sd[k0 + i];
			xv[i] = sx[row][k0 + i];
		}
		struct_step(c, xv); // forward form: y_j = sum_k x_k a[k][j]
		for (int i = 0; i < 4; ++i) {
			const int j = k0 + i;
			double want = 0.0;
			for (int k = 0; k < 64; ++k) {
				const double akj = k > j ? sP[k] * sq[j] : (k < j ? sR[k] * sc[j] : sd[j] + sP[j] * sq[j] + sR[j] * sc[j]);
				want += sx[row][k] * akj;
			}
			if (fabs(xv[i] - want) > 1e-13 * fabs(want)) bad |= 4096u;
		}
	}
	{ // one-state-per-lane structured step (fused backward + counts kernel) against the same dense product
		__shared__ double tP[64], tR[64], tq[64], tc[64], td[64];
		tP[lane] = 0.01 + 0.003 * lane; tR[lane] = 0.02 / (1.0 + lane); tq[lane] = 0.5 + 0.01 * lane;
		tc[lane] = 1.0 / (3.0 + 0.2 * lane); td[lane] = 0.9 - 0.002 * lane;
		__syncthreads();
		const WaveScanMasks wm = wave_scan_masks(lane);
		StructPar1 c1; c1.mS = tP[lane]; c1.wS = tq[lane]; c1.mP = tR[lane]; c1.wP = tc[lane]; c1.dd = td[lane];
		const double got = struct_step1(c1, x, wm);
		double want = 0.0;
		for (int k = 0; k < 64; ++k) {
			const double akj = k > lane ? tP[k] * tq[lane] : (k < lane ? tR[k] * tc[lane] : td[lane] + tP[lane] * tq[lane] + tR[lane] * tc[lane]);
			want += sh[k] * akj;
		}
		if (fabs(got - want) > 1e-13 * fabs(want)) bad |= 8192u;
		double pre = 0.0;
		for (int k = 0; k <= lane; ++k) pre += sh[k];
		if (fabs(wave_prefix_incl(x, wm) - pre) > 1e-13 * pre || fabs(wave_prefix_incl_bc(x) - pre) > 1e-13 * pre) bad |= 16384u;
	}
	if (bad) atomicOr(flags, bad);
}

int run_selftest(hipStream_t stream, unsigned *d_flags)
{
	hipLaunchKernelGGL(k_selftest, dim3(4), dim3(64), 0, stream, d_flags);
	return (int)hipGetLastError();
}

} // namespace psmc
