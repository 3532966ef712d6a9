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
// These general kernels are about ACCEPTING the input, not about speed: a position costs 13 us at n = 200 (7.5 forward + 5.9 backward: the n matrix
// elements of a lane come from the L2, eight per round trip; fetching 32 a block ahead by hand was SLOWER, 10.4 + 11.1 us -- round 5,
// profiles/r05_wide_timing.json).  Round 6: 129 .. 224 states keep the matrix in REGISTERS instead (k_fwd_wide2 / k_bwd_wide2 below): 5.9 us per
// position at n = 200, 4.9 at 149 (profiles/r06_wide_timing.json) -- 3.1e6 bins/s over 20 segments, 130 x one host core of the reference.
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

// ---------------------------------------------------------------- 129 .. 224 states: the matrix in REGISTERS (round 6)
// k_fwd_wide / k_bwd_wide fetch every lane's n matrix elements from the L2 at every position: 7.5 + 5.9 us per position at n = 200, nearly
// all of it round trips (profiles/r05_wide_timing.json).  A compute unit's register file holds 64 K doubles -- an n x n matrix up to n = 224
// with room to work in -- so here a work-group of 2 S threads (two waves per SIMD, 256 registers each) keeps it there: thread (q, k) owns
// state k's column (forward; row, backward) for l in [q C, q C + C), C = 96 / 104 / 112: C matrix elements in registers for the whole
// sweep, zero beyond n.  The reference's left-to-right sum over l (khmm.c:180, 232) is walked in two STAGES: the threads of stage 0 add their
// C products onto 0.0 in order and hand the partial sum to stage 1 through LDS, which ends with the full sum -- the same additions in the
// same order as one thread doing all n (a zero product appended to a part leaves its sum as it is), hence the same bits.  One barrier per
// stage.  The operands that are the same for every state (the previous row, the emission factors) come out of LDS as broadcast reads, several
// in flight in a ring of registers: left to itself the compiler keeps ONE read in flight and waits for it before every pair of additions.
// The normaliser of the forward sweep is the reference's sum in state order, added by every wave of the last stage for itself out of LDS.
typedef double d2w_t __attribute__((ext_vector_type(2)));
constexpr int WRING = 6; // 16-byte LDS reads in flight
template <int C>
__device__ __forceinline__ double stage_dot(double acc, const double *xv, const double (&m)[C])
{
	static_assert(C % 2 == 0 && C / 2 > WRING, "pairs");
	const d2w_t *x2 = reinterpret_cast<const d2w_t *>(xv);
	d2w_t ring[WRING];
#pragma unroll
	for (int i = 0; i < WRING; ++i) ring[i] = x2[i];
#pragma unroll
	for (int i = 0; i < C / 2; ++i) {
		const d2w_t v = ring[i % WRING];
		if (i + WRING < C / 2) ring[i % WRING] = x2[i + WRING];
		acc += v.x * m[2 * i];                                      // tmp += fu1[l] * aa[l], l in order (khmm.c:180)
		acc += v.y * m[2 * i + 1];
	}
	return acc;
}
template <int C>
__device__ __forceinline__ double stage_dot_e(double acc, const double *bv, const double *ev, const double (&m)[C])
{
	constexpr int R = WRING / 2;
	const d2w_t *b2 = reinterpret_cast<const d2w_t *>(bv), *e2 = reinterpret_cast<const d2w_t *>(ev);
	d2w_t rb[R], re[R];
#pragma unroll
	for (int i = 0; i < R; ++i) { rb[i] = b2[i]; re[i] = e2[i]; }
#pragma unroll
	for (int i = 0; i < C / 2; ++i) {
		const d2w_t vb = rb[i % R], ve = re[i % R];
		if (i + R < C / 2) { rb[i % R] = b2[i + R]; re[i % R] = e2[i + R]; }
		const double q0 = ve.x * m[2 * i]; acc += q0 * vb.x;      // tmp += q[l] * bu1[l], q = e * a rounded first (khmm.c:203, 231-232)
		const double q1 = ve.y * m[2 * i + 1]; acc += q1 * vb.y;
	}
	return acc;
}
// strict left-to-right sum of v[0..n) out of LDS, WRING reads in flight.  v[n ..] must be +0.0 up to the next multiple of 2 WRING and readable one
// ring turn beyond that: the sum runs over whole turns of the ring (adding +0.0 to a non-negative sum leaves its bits as they are), so that no
// read needs a bound
__device__ __forceinline__ double ordered_sum_ring(const double *v, int n)
{
	const d2w_t *v2 = reinterpret_cast<const d2w_t *>(v);
	const int turns = (n + 2 * WRING - 1) / (2 * WRING);
	double s = 0.0;
	d2w_t ring[WRING];
#pragma unroll
	for (int i = 0; i < WRING; ++i) ring[i] = v2[i];
	for (int tr = 0; tr < turns; ++tr) { // (a whole turn of the ring per trip: the registers keep their places)
		v2 += WRING;
#pragma unroll
		for (int r = 0; r < WRING; ++r) {
			const d2w_t t = ring[r];
			ring[r] = v2[r];
			s += t.x; s += t.y;
		}
	}
	return s;
}

template <int C>
__global__ __launch_bounds__(512) void k_fwd_wide2(const double *__restrict__ a, const double *__restrict__ e, const double *__restrict__ a0,
                                                   const uint8_t *__restrict__ obs, const int64_t *__restrict__ seg_off,
                                                   const int32_t *__restrict__ seg_len, const ExWork wl, int n, int S,
                                                   double *__restrict__ f, double *__restrict__ s)
{
	extern __shared__ double lds_w[];
	double *xs = lds_w, *gs = lds_w + 2 * S, *es = lds_w + 3 * S, *hand = lds_w + 6 * S; // xs[2][S] | gs[S] | es[3][S] | hand[S]
	const int t = threadIdx.x, q = t / S, k = t - q * S;
	const int seg = wl.seg[blockIdx.x];
	if (seg < 0) return; // padding entry of a batch (block-uniform)
	const int64_t off = seg_off[seg], toff = wl.tab ? wl.tab[blockIdx.x] : off;
	{ const int64_t po = wl.par ? wl.par[blockIdx.x] * wl.par_stride : 0; a += po; e += po; a0 += po; }
	const int L = seg_len[seg];
	const uint8_t *o = obs + off;
	double *fo = f + toff * S, *so = s + toff;
	for (int i = t; i < 3 * S; i += 2 * S) es[i] = e[i];
	for (int i = t; i < 3 * S; i += 2 * S) xs[i] = 0.0; // (a part may reach past n: those products are 0 * 0; gs beyond n: +0.0 for ordered_sum_ring)
	double m[C]; // at[k][l] = a[l][k] (khmm.c:162-166), l = q C + j
#pragma unroll
	for (int j = 0; j < C; ++j) { const int l = q * C + j; m[j] = (l < n && l < S) ? a[(int64_t)l * S + k] : 0.0; }
	__syncthreads();
	int cur = 0;
	double g = 0.0;
	int sym = o[0];
	for (int u = 0; u < L; ++u) { // index u = position u + 1
		const int sym_next = o[min(u + 1, L - 1)];
		if (u == 0) {
			if (q == 1) { g = k < n ? a0[k] * es[sym * S + k] : 0.0; gs[k] = g; }                    // khmm.c:171-172
			__syncthreads();
		} else {
			const double *xv = xs + cur * S + q * C;
			if (q == 0) hand[k] = stage_dot<C>(0.0, xv, m);
			__syncthreads();
			if (q == 1) { g = es[sym * S + k] * stage_dot<C>(hand[k], xv, m); gs[k] = g; }            // khmm.c:179-180
			__syncthreads();
		}
		if (q == 1) { // every wave of the last stage: the same n additions in the same order, the same bits
			const double sum = ordered_sum_ring(gs, n);                                                 // khmm.c:181 (sum in state order)
			const double x = g / sum;                                                                   // khmm.c:173, 182
			fo[(int64_t)u * S + k] = x; xs[(cur ^ 1) * S + k] = x;
			if (k == 0) so[u] = sum;
		}
		cur ^= 1; sym = sym_next;
		__syncthreads();
	}
}

template <int C>
__global__ __launch_bounds__(512) void k_bwd_wide2(const double *__restrict__ aT, const double *__restrict__ e, const double *__restrict__ a0,
                                                   const uint8_t *__restrict__ obs, const int64_t *__restrict__ seg_off,
                                                   const int32_t *__restrict__ seg_len, const ExWork wl, int n, int S,
                                                   const double *__restrict__ s, double *__restrict__ b, double *__restrict__ chk)
{
	extern __shared__ double lds_w[];
	double *bs = lds_w, *es = lds_w + 3 * S, *hand = lds_w + 6 * S; // bs[2][S] | (unused [S]) | es[3][S] | hand[S]
	const int t = threadIdx.x, q = t / S, k = t - q * S;
	const int seg = wl.seg[blockIdx.x];
	if (seg < 0) return;
	const int64_t off = seg_off[seg], toff = wl.tab ? wl.tab[blockIdx.x] : off;
	{ const int64_t po = wl.par ? wl.par[blockIdx.x] * wl.par_stride : 0; aT += po; e += po; a0 += po; }
	const int L = seg_len[seg];
	const uint8_t *o = obs + off;
	const double *so = s + toff;
	double *bo = b + toff * S;
	for (int i = t; i < 3 * S; i += 2 * S) es[i] = e[i];
	for (int i = t; i < 2 * S; i += 2 * S) bs[i] = 0.0;
	double m[C]; // aT[l * S + k] = a[k][l], l = q C + j
#pragma unroll
	for (int j = 0; j < C; ++j) { const int l = q * C + j; m[j] = (l < n && l < S) ? aT[(int64_t)l * S + k] : 0.0; }
	__syncthreads();
	double x = 1.0 / so[L - 1];  // b[L][k] = 1/s[L] (khmm.c:226)
	if (q == 1) { bo[(int64_t)(L - 1) * S + k] = x; bs[k] = x; }
	int cur = 0;
	double s_u = L >= 2 ? so[L - 2] : 1.0;
	int sym = L >= 2 ? o[L - 1] : 0;
	__syncthreads();
	for (int u = L - 2; u >= 0; --u) { // index u = position u + 1; uses b[u+1], obs[u+1], s[u]
		const double s_next = so[max(u - 1, 0)]; // (a round trip to memory: asked for a position ahead)
		const int sym_next = o[max(u, 1)];
		const double *bv = bs + cur * S + q * C, *ev = es + sym * S + q * C;
		if (q == 0) hand[k] = stage_dot_e<C>(0.0, bv, ev, m);
		__syncthreads();
		if (q == 1) { x = stage_dot_e<C>(hand[k], bv, ev, m) / s_u; bo[(int64_t)u * S + k] = x; bs[(cur ^ 1) * S + k] = x; } // khmm.c:233
		__syncthreads();
		cur ^= 1; s_u = s_next; sym = sym_next;
	}
	// khmm.c:237-238: sum_l a0[l] * b[1][l] * e[o_1][l], products left to right, the sum in state order
	if (q == 1) hand[k] = k < n ? a0[k] * x * es[o[0] * S + k] : 0.0;
	__syncthreads();
	if (t == 0) chk[blockIdx.x] = ordered_sum_lds(hand, n);
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
	const size_t lds2 = sizeof(double) * 7 * (size_t)S;
	static const bool no2 = getenv("PSMC_HIP_WIDE_L2") != nullptr; // (A/B, and the tests of the general kernels at sizes the register kernels take)
	const int reg2 = no2 ? 0 : (S == 192 ? 96 : (S == 256 && n <= 208 ? 104 : (S == 256 && n <= 224 ? 112 : 0))); // matrix in registers: 129 .. 224 states
#define PSMC_W2(C) do { \
		hipLaunchKernelGGL(k_fwd_wide2<C>, dim3(p.n_work), dim3(2 * S), lds2, p.stream, p.d_a, p.d_e, p.d_a0, p.d_obs, p.d_seg_off, p.d_seg_len, wl, n, S, p.d_f, p.d_s); \
		if (p.ev[1]) hipEventRecord(p.ev[1], p.stream); \
		hipLaunchKernelGGL(k_bwd_wide2<C>, dim3(p.n_work), dim3(2 * S), lds2, p.stream, p.d_aeT, p.d_e, p.d_a0, p.d_obs, p.d_seg_off, p.d_seg_len, wl, n, S, p.d_s, p.d_b, p.d_chk); } while (0)
	if (reg2 == 96) PSMC_W2(96); else if (reg2 == 104) PSMC_W2(104); else if (reg2 == 112) PSMC_W2(112);
	else {
		hipLaunchKernelGGL(k_fwd_wide, dim3(p.n_work), dim3(S), lds, p.stream, p.d_a, p.d_e, p.d_a0, p.d_obs, p.d_seg_off, p.d_seg_len, wl, n, S, p.d_f, p.d_s);
		if (p.ev[1]) hipEventRecord(p.ev[1], p.stream);
		hipLaunchKernelGGL(k_bwd_wide, dim3(p.n_work), dim3(S), lds, p.stream, p.d_aeT, p.d_e, p.d_a0, p.d_obs, p.d_seg_off, p.d_seg_len, wl, n, S, p.d_s, p.d_b, p.d_chk);
	}
#undef PSMC_W2
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
		if (k == 0) { const double sm = u < L - 1 ? ordered_sum_lds(lds_w, n) : 1.0; recomb[u] = sm != sm ? sm : 1.0 - sm; } // (NaN: sign kept, as estep_exact.hip k_post_full)
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
