// estep_wide.hip -- EXACT mode beyond 128 hidden states (129 .. 1024; `psmc -p "100*2"` gives 200).
//
// The reference allocates for any n (khmm.c:10-23) and cli.c:66-99 accepts any pattern.  The kernels of estep_exact.hip
// keep one or two states per lane and the matrix in registers / LDS; here the state dimension is TILED over the waves
// of a work-group instead: S = n rounded up to 64, one work-group of S/64 waves per segment, thread t owns state k = t.
//   forward   f[u][k] = e[o_u][k] * (sum_l f[u-1][l] * a[l][k]) / s[u]       khmm.c:176-185, l = 0..n-1 IN ORDER
//   backward  b[u][k] = (sum_l (e[o_{u+1}][l] * a[k][l]) * b[u+1][l]) / s[u]   khmm.c:228-235, e*a rounded first (khmm.c:203)
// The previous row lives in LDS (double buffered: one barrier per position for the exchange), every lane walks l in the
// reference's order reading x_l as an LDS broadcast and its matrix element from memory (`a` row-major for the forward
// sweep, its transpose for the backward one: consecutive lanes read consecutive addresses; the S x S matrix, 0.5 MB at
// S = 256, stays in the L2 of the XCD).  The normaliser s[u] = sum_k g_k is the reference's left-to-right sum: every
// lane adds the n values out of LDS itself -- the same n adds in the same order in all of them, hence the same bits --
// so nothing has to be broadcast back.  No FMA (the library is built with -ffp-contract=off; tests/test_abi.py audits
// this file's assembly), true division, every sum in index order: bit-identical to khmm.c, checked against the oracle
// and against goldens of the reference itself at 150 and 200 states (tests/test_gpu_wide.py).
// This path is about ACCEPTING the input, not about speed: a position costs 13 us at n = 200 (7.5 forward + 5.9 backward: the n matrix
// elements of a lane come from the L2, eight per round trip; fetching 32 a block ahead by hand was SLOWER, 10.4 + 11.1 us -- round 5,
// profiles/r05_wide_timing.json), i.e. 1.4e6 bins/s over 20 segments, 60 x one host core of the reference at that size.
#include <hip/hip_runtime.h>
#include "psmc_hip_internal.h"

namespace psmc {

#define PSMC_TINY_W 1e-25 /* HMM_TINY khmm.h:28 */

// strict left-to-right sum of v[0..n): ((0 + v0) + v1) + ...  (khmm.c:177-181 `sum += ...`)
__device__ __forceinline__ double ordered_sum_lds(const double *v, int n)
{
	double s = 0.0;
	int k = 0;
	for (; k + 8 <= n; k += 8) { // the reads are independent (issued together), the adds are the chain
		const double v0 = v[k], v1 = v[k + 1], v2 = v[k + 2], v3 = v[k + 3], v4 = v[k + 4], v5 = v[k + 5], v6 = v[k + 6], v7 = v[k + 7];
		s += v0; s += v1; s += v2; s += v3; s += v4; s += v5; s += v6; s += v7;
	}
	for (; k < n; ++k) s += v[k];
	return s;
}

// acc = sum_{l<n} x[l] * m[l * S]   (x in LDS, m = this lane's column, row stride S), left to right from 0.0
__device__ __forceinline__ double ordered_dot_col(const double *x, const double *__restrict__ m, int n, int S)
{
	double acc = 0.0;
	int l = 0;
	for (; l + 8 <= n; l += 8) {
		double mv[8], xv[8];
#pragma unroll
		for (int j = 0; j < 8; ++j) { mv[j] = m[(int64_t)(l + j) * S]; xv[j] = x[l + j]; }
#pragma unroll
		for (int j = 0; j < 8; ++j) acc += xv[j] * mv[j]; // tmp += fu1[l] * aa[l]
	}
	for (; l < n; ++l) acc += x[l] * m[(int64_t)l * S];
	return acc;
}

// the same with the emission factor: acc += (e[l] * m[l * S]) * x[l], the product e*a rounded first (hmm_pre_backward, khmm.c:203)
__device__ __forceinline__ double ordered_dot_col_e(const double *x, const double *ev, const double *__restrict__ m, int n, int S)
{
	double acc = 0.0;
	int l = 0;
	for (; l + 8 <= n; l += 8) {
		double mv[8], xv[8], ee[8];
#pragma unroll
		for (int j = 0; j < 8; ++j) { mv[j] = m[(int64_t)(l + j) * S]; xv[j] = x[l + j]; ee[j] = ev[l + j]; }
#pragma unroll
		for (int j = 0; j < 8; ++j) { const double q = ee[j] * mv[j]; acc += q * xv[j]; } // tmp += q[l] * bu1[l]
	}
	for (; l < n; ++l) { const double q = ev[l] * m[(int64_t)l * S]; acc += q * x[l]; }
	return acc;
}

// ---------------------------------------------------------------- forward (khmm.c:145-190)
// grid = entries, block = S threads.  LDS: xs[2][S] | gs[S] | es[3][S]
__global__ __launch_bounds__(1024) void k_fwd_wide(const double *__restrict__ a, const double *__restrict__ e, const double *__restrict__ a0,
                                                   const uint8_t *__restrict__ obs, const int64_t *__restrict__ seg_off,
                                                   const int32_t *__restrict__ seg_len, const ExWork wl, int n, int S,
                                                   double *__restrict__ f, double *__restrict__ s)
{
	extern __shared__ double lds_w[];
	double *xs = lds_w, *gs = lds_w + 2 * S, *es = lds_w + 3 * S;
	const int k = threadIdx.x;
	const int seg = wl.seg[blockIdx.x];
	if (seg < 0) return; // padding entry of a batch (block-uniform)
	const int64_t off = seg_off[seg], toff = wl.tab ? wl.tab[blockIdx.x] : off;
	{ const int64_t po = wl.par ? wl.par[blockIdx.x] * wl.par_stride : 0; a += po; e += po; a0 += po; }
	const int L = seg_len[seg];
	const uint8_t *o = obs + off;
	double *fo = f + toff * S, *so = s + toff;
	for (int i = k; i < 3 * S; i += S) es[i] = e[i];
	__syncthreads();
	const double *col = a + k; // at[k][l] = a[l][k] (khmm.c:162-166)
	int cur = 0;
	for (int u = 0; u < L; ++u) { // index u = position u + 1
		const int sym = o[u];
		double g;
		if (u == 0) g = k < n ? a0[k] * es[sym * S + k] : 0.0;                         // khmm.c:171-172
		else { const double tmp = ordered_dot_col(xs + cur * S, col, n, S); g = es[sym * S + k] * tmp; } // khmm.c:179-180
		gs[k] = g;
		__syncthreads();
		const double sum = ordered_sum_lds(gs, n);
		const double x = g / sum;                                                       // khmm.c:173, 182
		fo[(int64_t)u * S + k] = x;
		xs[(cur ^ 1) * S + k] = x;
		if (k == 0) so[u] = sum;
		cur ^= 1;
		__syncthreads(); // the next position reads xs[cur] of every state and rewrites gs
	}
}

// ---------------------------------------------------------------- backward (khmm.c:210-241) + the underflow check value
// LDS: bs[2][S] | ts[S] | es[3][S]
__global__ __launch_bounds__(1024) void k_bwd_wide(const double *__restrict__ aT, const double *__restrict__ e, const double *__restrict__ a0,
                                                   const uint8_t *__restrict__ obs, const int64_t *__restrict__ seg_off,
                                                   const int32_t *__restrict__ seg_len, const ExWork wl, int n, int S,
                                                   const double *__restrict__ s, double *__restrict__ b, double *__restrict__ chk)
{
	extern __shared__ double lds_w[];
	double *bs = lds_w, *ts = lds_w + 2 * S, *es = lds_w + 3 * S;
	const int k = threadIdx.x;
	const int seg = wl.seg[blockIdx.x];
	if (seg < 0) return;
	const int64_t off = seg_off[seg], toff = wl.tab ? wl.tab[blockIdx.x] : off;
	{ const int64_t po = wl.par ? wl.par[blockIdx.x] * wl.par_stride : 0; aT += po; e += po; a0 += po; }
	const int L = seg_len[seg];
	const uint8_t *o = obs + off;
	const double *so = s + toff;
	double *bo = b + toff * S;
	for (int i = k; i < 3 * S; i += S) es[i] = e[i];
	const double *row = aT + k; // aT[l * S + k] = a[k][l]
	double x = 1.0 / so[L - 1];  // b[L][k] = 1/s[L] (khmm.c:226)
	bo[(int64_t)(L - 1) * S + k] = x;
	int cur = 0;
	bs[k] = x;
	__syncthreads();
	for (int u = L - 2; u >= 0; --u) { // index u = position u + 1; uses b[u+1], obs[u+1], s[u]
		const int sym = o[u + 1];
		const double tmp = ordered_dot_col_e(bs + cur * S, es + sym * S, row, n, S); // khmm.c:231-232
		x = tmp / so[u];                                                              // khmm.c:233
		bo[(int64_t)u * S + k] = x;
		bs[(cur ^ 1) * S + k] = x;
		cur ^= 1;
		__syncthreads();
	}
	// khmm.c:237-238: sum_l a0[l] * b[1][l] * e[o_1][l], products left to right, the sum in state order
	ts[k] = k < n ? a0[k] * x * es[o[0] * S + k] : 0.0;
	__syncthreads();
	if (k == 0) chk[blockIdx.x] = ordered_sum_lds(ts, n);
}

// ---------------------------------------------------------------- expect (khmm.c:297-324)
// grid = (entries, (S/4) * H + H), H = S / 64, one wave each: the first (S/4) * H blocks accumulate rows 4g .. 4g+3 of A for
// the 64 columns of one column block (lane = column) in position order; the last H accumulate E and A0 of 64 states.
// Same arithmetic as k_expect_exact's generic path, with the stride at run time.
__global__ __launch_bounds__(64) void k_expect_wide(const double *__restrict__ a, const double *__restrict__ e, const double *__restrict__ a0,
                                                    const uint8_t *__restrict__ obs, const int64_t *__restrict__ seg_off,
                                                    const int32_t *__restrict__ seg_len, const ExWork wl, int S,
                                                    const double *__restrict__ f, const double *__restrict__ b, const double *__restrict__ s,
                                                    double *__restrict__ segA, double *__restrict__ segE, double *__restrict__ segA0)
{
	const int H = S / 64, NA = (S / 4) * H;
	const int lane = threadIdx.x;
	const int seg = wl.seg[blockIdx.x];
	if (seg < 0) return;
	const int64_t off = seg_off[seg], toff = wl.tab ? wl.tab[blockIdx.x] : off;
	{ const int64_t po = wl.par ? wl.par[blockIdx.x] * wl.par_stride : 0; a += po; e += po; a0 += po; }
	const int L = seg_len[seg];
	const uint8_t *o = obs + off;
	const double *fo = f + toff * S, *bo = b + toff * S, *so = s + toff;
	constexpr int BLK = 8;
	if ((int)blockIdx.y < NA) {
		const int k0 = ((int)blockIdx.y / H) * 4;
		const int col = lane + 64 * ((int)blockIdx.y % H);
		double q[3][4]; // ae[sym][k0+j][l] = e[sym][l] * a[k0+j][l], one rounding (khmm.c:194-206)
#pragma unroll
		for (int sy = 0; sy < 3; ++sy)
#pragma unroll
			for (int j = 0; j < 4; ++j) q[sy][j] = e[sy * S + col] * a[(int64_t)(k0 + j) * S + col];
		double acc[4] = {PSMC_TINY_W, PSMC_TINY_W, PSMC_TINY_W, PSMC_TINY_W}; // khmm.c:305-306
		const int nn = L - 1; // u = 1..L-1, index i = u-1: f[i][k], b[i+1][l], obs[i+1]
		for (int i0 = 0; i0 < nn; i0 += BLK) {
			const int nb = min(BLK, nn - i0);
			double fv[BLK][4], bv[BLK]; int sy[BLK];
#pragma unroll
			for (int t = 0; t < BLK; ++t) {
				const int i = min(i0 + t, nn - 1);
				bv[t] = bo[(int64_t)(i + 1) * S + col]; sy[t] = o[i + 1];
#pragma unroll
				for (int j = 0; j < 4; ++j) fv[t][j] = fo[(int64_t)i * S + k0 + j];
			}
#pragma unroll
			for (int t = 0; t < BLK; ++t) {
				if (t < nb) {
					const int sm = sy[t];
#pragma unroll
					for (int j = 0; j < 4; ++j) {
						const double qq = sm == 0 ? q[0][j] : (sm == 1 ? q[1][j] : q[2][j]);
						acc[j] += fv[t][j] * qq * bv[t]; // khmm.c:316: AA[l] += fuk * q[l] * bu1[l]
					}
				}
			}
		}
		double *out = segA + (int64_t)blockIdx.x * ((int64_t)S * S);
#pragma unroll
		for (int j = 0; j < 4; ++j) out[(int64_t)(k0 + j) * S + col] = acc[j];
	} else {
		const int col = lane + 64 * ((int)blockIdx.y - NA);
		double E0 = PSMC_TINY_W, E1 = PSMC_TINY_W, E2 = PSMC_TINY_W; // khmm.c:307-308
		const int nn = L - 1; // u = 1..L-1, index i = u-1: f[i], b[i], s[i], obs[i]
		for (int i0 = 0; i0 < nn; i0 += BLK) {
			const int nb = min(BLK, nn - i0);
			double fv[BLK], bv[BLK], sv[BLK]; int sy[BLK];
#pragma unroll
			for (int t = 0; t < BLK; ++t) {
				const int i = min(i0 + t, nn - 1);
				fv[t] = fo[(int64_t)i * S + col]; bv[t] = bo[(int64_t)i * S + col]; sv[t] = so[i]; sy[t] = o[i];
			}
#pragma unroll
			for (int t = 0; t < BLK; ++t) {
				if (t < nb) {
					const double v = fv[t] * bv[t] * sv[t]; // khmm.c:317: Ec[k] += fuk * bu[k] * ss
					if (sy[t] == 0) E0 += v; else if (sy[t] == 1) E1 += v; else E2 += v;
				}
			}
		}
		double *oe = segE + (int64_t)blockIdx.x * (3 * S);
		oe[col] = E0; oe[S + col] = E1; oe[2 * S + col] = E2;
		const int sym1 = o[0]; // khmm.c:321-322: A0[l] += a0[l]*e[o_1][l]*b[1][l], A0 starts at 0
		segA0[(int64_t)blockIdx.x * S + col] = 0.0 + a0[col] * e[sym1 * S + col] * bo[col];
	}
}

int launch_exact_wide(const EstepLaunch &p)
{
	const int S = p.ns, n = p.n_states;
	if (p.n_work <= 0) return 0;
	(void)hipGetLastError();
	const ExWork wl = {p.d_work, p.d_work_par, p.d_work_tab, nullptr, p.n_work, p.par_stride};
	const size_t lds = sizeof(double) * 6 * (size_t)S;
	const int H = S / 64;
	if (p.ev[0]) hipEventRecord(p.ev[0], p.stream);
	hipLaunchKernelGGL(k_fwd_wide, dim3(p.n_work), dim3(S), lds, p.stream, p.d_a, p.d_e, p.d_a0, p.d_obs, p.d_seg_off, p.d_seg_len, wl, n, S, p.d_f, p.d_s);
	if (p.ev[1]) hipEventRecord(p.ev[1], p.stream);
	hipLaunchKernelGGL(k_bwd_wide, dim3(p.n_work), dim3(S), lds, p.stream, p.d_aeT, p.d_e, p.d_a0, p.d_obs, p.d_seg_off, p.d_seg_len, wl, n, S, p.d_s, p.d_b, p.d_chk);
	if (p.ev[2]) hipEventRecord(p.ev[2], p.stream);
	hipLaunchKernelGGL(k_expect_wide, dim3(p.n_work, (S / 4) * H + H), dim3(64), 0, p.stream, p.d_a, p.d_e, p.d_a0, p.d_obs, p.d_seg_off, p.d_seg_len, wl, S,
	                   p.d_f, p.d_b, p.d_s, p.d_segA, p.d_segE, p.d_segA0);
	if (p.ev[3]) hipEventRecord(p.ev[3], p.stream);
	if (p.ev[4]) hipEventRecord(p.ev[4], p.stream);
	return (int)hipGetLastError();
}

// ---------------------------------------------------------------- decoding on the resident tables (aux.c:150-231), any S
// hmm_post_decode (khmm.c:264-281): first maximum wins.  One wave per 64 positions; lane holds states lane, lane + 64, ...
__global__ __launch_bounds__(64) void k_post_decode_wide(const double *__restrict__ f, const double *__restrict__ b, const double *__restrict__ s,
                                                         int64_t off, int L, int n, int S, int32_t *__restrict__ path, double *__restrict__ maxp)
{
	const int lane = threadIdx.x;
	const int u0 = blockIdx.x * 64, u1 = min(L, u0 + 64);
	for (int u = u0; u < u1; ++u) {
		const int64_t g = off + u;
		const double ss = s[g];
		double v = -1.0; int k = lane;
		for (int kk = lane; kk < n; kk += 64) { // ascending: a later state replaces an earlier one only when strictly larger
			const double v2 = f[g * S + kk] * b[g * S + kk] * ss;
			if (v2 > v) { v = v2; k = kk; }
		}
#pragma unroll
		for (int m = 32; m >= 1; m >>= 1) {
			const double ov = __shfl_xor(v, m, 64);
			const int ok = __shfl_xor(k, m, 64);
			if (ov > v || (ov == v && ok < k)) { v = ov; k = ok; }
		}
		if (lane == 0) { path[u] = k; maxp[u] = v; }
	}
}

// aux.c:183-200: post[u][l] = f*b*s (hmm_post_state), recomb[u] = 1 - sum_l f[u][l]*a[l][l]*b[u+1][l]*e[o_{u+1}][l] in state order
// block = S threads (thread = state), one block per 16 positions
__global__ __launch_bounds__(1024) void k_post_full_wide(const double *__restrict__ a, const double *__restrict__ e, const uint8_t *__restrict__ obs,
                                                         const double *__restrict__ f, const double *__restrict__ b, const double *__restrict__ s,
                                                         int64_t off, int L, int n, int S, double *__restrict__ post, double *__restrict__ recomb)
{
	extern __shared__ double lds_w[];
	const int k = threadIdx.x;
	const int u0 = blockIdx.x * 16, u1 = min(L, u0 + 16);
	const double dg = k < n ? a[(int64_t)k * S + k] : 0.0;
	for (int u = u0; u < u1; ++u) {
		const int64_t g = off + u;
		const double ss = s[g], fu = f[g * S + k];
		if (post && k < n) post[(int64_t)u * n + k] = fu * b[g * S + k] * ss;
		if (!recomb) continue; // (uniform)
		double t = 0.0;
		if (u < L - 1 && k < n) t = fu * dg * b[(g + 1) * S + k] * e[obs[g + 1] * S + k]; // fu[l] * a[l][l] * bu1[l] * eu1[l]
		lds_w[k] = t;
		__syncthreads();
		if (k == 0) recomb[u] = u < L - 1 ? 1.0 - ordered_sum_lds(lds_w, n) : 0.0;
		__syncthreads();
	}
}

// aux.c:202-219: cnt[l][j] += post[u][l] * cnt1[u][j] for u = 1..min_l in position order.  grid = (count columns, S / 64), lane = state
__global__ __launch_bounds__(64) void k_post_counts_wide(const double *__restrict__ f, const double *__restrict__ b, const double *__restrict__ s,
                                                         int64_t off, int min_l, const int32_t *__restrict__ cnt1, int n_cnt, int n, int S,
                                                         double *__restrict__ cnt)
{
	const int k = threadIdx.x + 64 * blockIdx.y, j = blockIdx.x;
	if (k >= n) return;
	double acc = cnt[(int64_t)k * n_cnt + j];
	for (int u = 0; u < min_l; ++u) {
		const int64_t r = (off + u) * S + k;
		acc += f[r] * b[r] * s[off + u] * (double)cnt1[(int64_t)u * n_cnt + j]; // cnt += prob[l] * cnt1
	}
	cnt[(int64_t)k * n_cnt + j] = acc;
}

int launch_post_decode_wide(hipStream_t st, const double *f, const double *b, const double *s, int64_t off, int L, int n, int S, int32_t *path, double *maxp)
{
	hipLaunchKernelGGL(k_post_decode_wide, dim3((L + 63) / 64), dim3(64), 0, st, f, b, s, off, L, n, S, path, maxp);
	return (int)hipGetLastError();
}
int launch_post_full_wide(hipStream_t st, const double *a, const double *e, const uint8_t *obs, const double *f, const double *b, const double *s,
                          int64_t off, int L, int n, int S, double *post, double *recomb)
{
	hipLaunchKernelGGL(k_post_full_wide, dim3((L + 15) / 16), dim3(S), sizeof(double) * S, st, a, e, obs, f, b, s, off, L, n, S, post, recomb);
	return (int)hipGetLastError();
}
int launch_post_counts_wide(hipStream_t st, const double *f, const double *b, const double *s, int64_t off, int min_l, const int32_t *cnt1,
                            int n_cnt, int n, int S, double *cnt)
{
	if (min_l <= 0 || n_cnt <= 0) return 0;
	hipLaunchKernelGGL(k_post_counts_wide, dim3(n_cnt, S / 64), dim3(64), 0, st, f, b, s, off, min_l, cnt1, n_cnt, n, S, cnt);
	return (int)hipGetLastError();
}

} // namespace psmc
