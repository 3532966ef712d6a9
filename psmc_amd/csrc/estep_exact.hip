// estep_exact.hip -- EXACT mode of the PSMC E-step for gfx950.
//
// Reproduces khmm.c's hmm_forward / hmm_backward / hmm_expect (lh3/psmc
// khmm.c:145-190, 210-241, 297-324) BIT FOR BIT: strict IEEE-754 double, no FMA
// contraction (file built with -ffp-contract=off; the one fused instruction in it,
// v_fmac_f64_dpp s, x, 1.0 of the ordered sums, rounds x + s once: the IEEE sum --
// wave_prims.h add_bcast16, audited by tests/test_abi.py), every sum in
// the reference's index order, true division.  That forbids tree reductions,
// atomics and any split of a segment along the sequence, so the parallelism is
// one wavefront per segment (lane = hidden state k) for the two sweeps, and
// (segment x 4-row group) workgroups for the ordered accumulation of the
// expected counts.  Data layout (bin g = seg_off + position-1):
//   obs[g] u8 {0,1,2} | f[g*64+k], b[g*64+k], s[g] fp64 | aeT[b][l*64+k] = e[b][l]*a[k][l]
// States are padded to 64 with zero probabilities (adds of +0.0 are exact).
#include <hip/hip_runtime.h>
#include "wave_prims.h"
#include "psmc_hip_internal.h"

namespace psmc {

#define PSMC_TINY 1e-25 /* HMM_TINY khmm.h:28 */

__device__ __forceinline__ double pick_e(int sym, double e0, double e1) {
	return sym == 0 ? e0 : (sym == 1 ? e1 : 1.0); // e[2][*] = 1 (khmm.c:21)
}

// ---------------------------------------------------------------- forward
// khmm.c:145-190.  One wave per selected segment.
// STORE_F = false: only the scale factors s[] are written (the batch's table-free forward pass: k_expect_exact_rf below recomputes f)
template <int REP, bool STORE_F = true>
__global__ __launch_bounds__(64) void k_fwd_exact(const double *__restrict__ a, const double *__restrict__ e,
                                                    const double *__restrict__ a0, const uint8_t *__restrict__ obs,
                                                    const int64_t *__restrict__ seg_off, const int32_t *__restrict__ seg_len,
                                                    const ExWork wl, double *__restrict__ f,
                                                    double *__restrict__ s)
{
	const int lane = threadIdx.x;
	const int seg = wl.seg[blockIdx.x];
	if (seg < 0) return; // padding entry of a batch
	const int64_t off = seg_off[seg], toff = wl.tab ? wl.tab[blockIdx.x] : off;
	{ const int64_t po = wl.par ? wl.par[blockIdx.x] * wl.par_stride : 0; a += po; e += po; a0 += po; }
	const int L = seg_len[seg];
	double col[64]; // at[k][l] = a[l][k] (khmm.c:162-166)
#pragma unroll
	for (int l = 0; l < 64; ++l) col[l] = a[l * 64 + lane];
	const double e0 = e[lane], e1 = e[64 + lane];
	const uint8_t *o = obs + off;
	double *fo = f + toff * 64, *so = s + toff;

	int symv = o[lane]; // obs is padded by >= 64 bytes at the end
	double x;
	{ // position 1 (khmm.c:171-174)
		const int sym = __builtin_amdgcn_readlane(symv, 0);
		const double g = a0[lane] * pick_e(sym, e0, e1);
		double sum;
		{ double r[4]; rep_rows<REP>(g, r); sum = seq_sum_rep(r); }
		x = g / sum;
		if (STORE_F) fo[lane] = x;
		if (lane == 0) so[0] = sum;
	}
	for (int base = 0; base < L; base += 64) { // positions base+1 .. base+64
		const int nb = min(64, L - base);
		const int symn = (base + 64 < L) ? (int)o[base + 64 + lane] : 2;
		for (int i = (base == 0 ? 1 : 0); i < nb; ++i) { // khmm.c:176-185
			const int sym = __builtin_amdgcn_readlane(symv, i);
			double tmp, sum;
			{ double r[4]; rep_rows<REP>(x, r); tmp = xdot64(r, col); }
			const double g = pick_e(sym, e0, e1) * tmp;
			{ double q[4]; rep_rows<REP>(g, q); sum = seq_sum_rep(q); }
			x = g / sum;
			if (STORE_F) fo[(int64_t)(base + i) * 64 + lane] = x;
			if (lane == 0) so[base + i] = sum;
		}
		symv = symn;
	}
}

// ---------------------------------------------------------------- backward
// khmm.c:210-241.  4 waves per block, one segment each.  The hom-symbol row of
// ae lives in VGPRs; the (rare) het and missing rows are read from LDS.
template <int REP>
__global__ __launch_bounds__(256) void k_bwd_exact(const double *__restrict__ aeT, const double *__restrict__ e,
                                                     const double *__restrict__ a0, const uint8_t *__restrict__ obs,
                                                     const int64_t *__restrict__ seg_off, const int32_t *__restrict__ seg_len,
                                                     const ExWork wl,
                                                     const double *__restrict__ s, double *__restrict__ b,
                                                     double *__restrict__ chk)
{
	__shared__ double lds_ae[2 * 4096]; // [0]: het (b=1), [1]: missing (b=2); [l*64+k]
	// the four sweeps of a block share the LDS copy: a batch keeps the items of one parameter set block-aligned
	{ const int64_t po = wl.par ? wl.par[blockIdx.x * 4] * wl.par_stride : 0; aeT += po; e += po; a0 += po; }
	for (int i = threadIdx.x; i < 2 * 4096; i += 256) lds_ae[i] = aeT[4096 + i];
	__syncthreads();
	const int lane = threadIdx.x & 63;
	const int w = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); // wave-uniform: segment, length and the loop run on the scalar unit
	if (w >= wl.n) return;
	const int seg = wl.seg[w];
	if (seg < 0) return; // padding entry of a batch
	const int64_t off = seg_off[seg], toff = wl.tab ? wl.tab[w] : off;
	const int L = seg_len[seg];
	double row0[64];
#pragma unroll
	for (int l = 0; l < 64; ++l) row0[l] = aeT[l * 64 + lane];
	const uint8_t *o = obs + off;
	const double *so = s + (wl.tab_s ? wl.tab_s[w] : toff);
	double *bo = b + toff * 64;

	// b[L][k] = 1/s[L] (khmm.c:226)
	double x = 1.0 / so[L - 1];
	bo[(int64_t)(L - 1) * 64 + lane] = x;
	// positions u = L-1 .. 1; index of u is u-1.  Walk blocks of 64 indices downwards.
	for (int top = L - 2; top >= 0; top -= 64) { // indices top, top-1, ..., max(top-63,0)
		const int nb = min(64, top + 1);
		const int lo = top - 63; // lane i holds index lo+i (may be < 0 -> clamp)
		const int idx = max(lo + lane, 0);
		const double sv = so[idx];
		const int symv = o[idx + 1]; // symbol of position u+1 (index u)
		for (int i = 63; i >= 64 - nb; --i) {
			const int sym = __builtin_amdgcn_readlane(symv, i);
			const double su = readlane_f64(sv, i);
			double r[4];
			rep_rows<REP>(x, r);
			dpp_guard(r);
			double tmp;
			if (sym == 0) {
				tmp = xdot64_guarded(r, row0);
			} else {
				const double *t = lds_ae + (sym - 1) * 4096 + lane;
				double acc = 0.0;
#pragma unroll
				for (int j = 0; j < 4; ++j) { // same order as xdot64, 16 LDS operands at a time
					double m[16];
#pragma unroll
					for (int l = 0; l < 16; ++l) m[l] = t[(16 * j + l) * 64];
					PSMC_XDOT16(r[j], m, 0)
				}
				tmp = acc;
			}
			x = tmp / su;
			bo[(int64_t)(lo + i) * 64 + lane] = x;
		}
	}
	{ // underflow check value (khmm.c:237-238): sum_l a0[l]*b[1][l]*e[o_1][l]
		const int sym = o[0];
		const double t = a0[lane] * x * e[sym * 64 + lane];
		double c;
		{ double r[4]; rep_rows<REP>(t, r); c = seq_sum_rep(r); }
		if (lane == 0) chk[w] = c;
	}
}

// ---------------------------------------------------------------- expect
// One position of hmm_expect for four rows k0..k0+3 of A in the lane's column: acc[j] += f[u][k0+j] * q[sym][j] * b[u+1][l]
// (khmm.c:316, products left to right, then the sum).  `ft` holds sixteen f values -- four positions x four rows -- in the lanes
// m = 4 (position) + j of EVERY 16-lane row, so each operand is one v_mov_b64_dpp instead of two v_readlane.  The whole position,
// the three-way choice of the emission row included, is ONE asm block: a wave alone on its SIMD pays four cycles for every
// instruction of any kind (wave_prims.h xdot16), and hipcc's lowering of the choice costs more than the sixteen instructions
// it selects (flag registers for a uniform branch, a copy of the four sums per position).  Homozygous path: 19 instructions.
#define PSMC_EXBODY(Q0, Q1, Q2, Q3, N0, N1, N2, N3)                                                       \
	"v_mov_b64_dpp %4, %9 row_newbcast:" #N0 " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                \
	"v_mov_b64_dpp %5, %9 row_newbcast:" #N1 " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                \
	"v_mov_b64_dpp %6, %9 row_newbcast:" #N2 " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                \
	"v_mov_b64_dpp %7, %9 row_newbcast:" #N3 " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                \
	"v_mul_f64 %4, %4, " Q0 "\n\tv_mul_f64 %5, %5, " Q1 "\n\tv_mul_f64 %6, %6, " Q2 "\n\tv_mul_f64 %7, %7, " Q3 "\n\t" \
	"v_mul_f64 %4, %4, %10\n\tv_mul_f64 %5, %5, %10\n\tv_mul_f64 %6, %6, %10\n\tv_mul_f64 %7, %7, %10\n\t"        \
	"v_add_f64 %0, %0, %4\n\tv_add_f64 %1, %1, %5\n\tv_add_f64 %2, %2, %6\n\tv_add_f64 %3, %3, %7\n\t"
#define PSMC_EXPOS(N0, N1, N2, N3)                                                                          \
	asm("v_readlane_b32 %8, %11, %12\n\t"                                                                   \
	    "s_cmp_eq_u32 %8, 0\n\ts_cbranch_scc1 .Lex0_%=\n\t"                                                  \
	    "s_cmp_eq_u32 %8, 1\n\ts_cbranch_scc1 .Lex1_%=\n\t"                                                  \
	    PSMC_EXBODY("%21", "%22", "%23", "%24", N0, N1, N2, N3) "s_branch .Lexe_%=\n"                           \
	    ".Lex1_%=:\n\t" PSMC_EXBODY("%17", "%18", "%19", "%20", N0, N1, N2, N3) "s_branch .Lexe_%=\n"          \
	    ".Lex0_%=:\n\t" PSMC_EXBODY("%13", "%14", "%15", "%16", N0, N1, N2, N3)                                 \
	    ".Lexe_%=:"                                                                                             \
	    : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&s"(st) \
	    : "v"(ft), "v"(bl), "v"(sv), "n"(T), "v"(q[0][0]), "v"(q[0][1]), "v"(q[0][2]), "v"(q[0][3]),    \
	      "v"(q[1][0]), "v"(q[1][1]), "v"(q[1][2]), "v"(q[1][3]), "v"(q[2][0]), "v"(q[2][1]), "v"(q[2][2]), "v"(q[2][3]) \
	    : "scc")
template <int T>
__device__ __forceinline__ void expect_pos4(double (&acc)[4], double ft, const double (&q)[3][4], double bl, int sv)
{
	double t0, t1, t2, t3; int st;
	if constexpr ((T & 3) == 0) { PSMC_EXPOS(0, 1, 2, 3); }
	else if constexpr ((T & 3) == 1) { PSMC_EXPOS(4, 5, 6, 7); }
	else if constexpr ((T & 3) == 2) { PSMC_EXPOS(8, 9, 10, 11); }
	else { PSMC_EXPOS(12, 13, 14, 15); }
}

// One position of the emission counts, lane = state: Ec[o_u][k] += f[u][k] * b[u][k] * s[u] (khmm.c:317); `ssv` holds sixteen
// scale factors in the lanes 0..15 of every row, `sv` the sixteen symbols.  Same reasoning as expect_pos4: 8 instructions.
template <int T>
__device__ __forceinline__ void expect_posE(double &E0, double &E1, double &E2, double fu, double bu, double ssv, int sv)
{
	double t, u; int st;
	asm("v_readlane_b32 %5, %9, %10\n\t"
	    "v_mov_b64_dpp %3, %8 row_newbcast:%10 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
	    "v_mul_f64 %4, %6, %7\n\t"
	    "v_mul_f64 %4, %4, %3\n\t"
	    "s_cmp_eq_u32 %5, 0\n\ts_cbranch_scc1 .Lee0_%=\n\t"
	    "s_cmp_eq_u32 %5, 1\n\ts_cbranch_scc1 .Lee1_%=\n\t"
	    "v_add_f64 %2, %2, %4\n\ts_branch .Leee_%=\n"
	    ".Lee1_%=:\n\tv_add_f64 %1, %1, %4\n\ts_branch .Leee_%=\n"
	    ".Lee0_%=:\n\tv_add_f64 %0, %0, %4\n"
	    ".Leee_%=:"
	    : "+v"(E0), "+v"(E1), "+v"(E2), "=&v"(t), "=&v"(u), "=&s"(st)
	    : "v"(fu), "v"(bu), "v"(ssv), "v"(sv), "n"(T)
	    : "scc");
}

// khmm.c:297-324.  S = padded number of states (64 or 128), H = S/64 column halves.
// grid = (n_work, (S/4)*H + H): the first (S/4)*H blocks accumulate rows 4g..4g+3 of
// A for one segment (lane = column lane+64*half) in position order; the last H
// accumulate E and A0 (lane = state lane+64*half).  Per-segment results (he of
// em.c:49) go to segA/segE/segA0; the host adds them in input order
// (hmm_add_expect, khmm.c:346-359).  For S == 64 `aeT` is the precomputed
// e[b][l]*a[k][l]; for S == 128 the same single-rounding product is formed here
// from `a` (row-major) and `e`.
template <int S>
__global__ __launch_bounds__(64) void k_expect_exact(const double *__restrict__ a, const double *__restrict__ aeT,
                                                       const double *__restrict__ e,
                                                       const double *__restrict__ a0, const uint8_t *__restrict__ obs,
                                                       const int64_t *__restrict__ seg_off,
                                                       const int32_t *__restrict__ seg_len, const ExWork wl,
                                                       const double *__restrict__ f, const double *__restrict__ b,
                                                       const double *__restrict__ s, double *__restrict__ segA,
                                                       double *__restrict__ segE, double *__restrict__ segA0)
{
	constexpr int H = S / 64, NA = (S / 4) * H;
	const int lane = threadIdx.x;
	const int seg = wl.seg[blockIdx.x];
	if (seg < 0) return; // padding entry of a batch
	const int64_t off = seg_off[seg], toff = wl.tab ? wl.tab[blockIdx.x] : off;
	{ const int64_t po = wl.par ? wl.par[blockIdx.x] * wl.par_stride : 0; a += po; aeT += po; e += po; a0 += po; }
	const int L = seg_len[seg];
	const uint8_t *o = obs + off;
	const double *fo = f + toff * S, *bo = b + toff * S, *so = s + toff;
	constexpr int BLK = 16;
	if (blockIdx.y < NA) {
		const int k0 = (blockIdx.y / H) * 4;
		const int col = lane + 64 * (blockIdx.y % H);
		double q[3][4]; // ae[sym][k0+j][l], l = col
#pragma unroll
		for (int sy = 0; sy < 3; ++sy)
#pragma unroll
			for (int j = 0; j < 4; ++j) {
				if constexpr (S == 64) q[sy][j] = aeT[sy * 4096 + lane * 64 + k0 + j];
				else q[sy][j] = e[sy * S + col] * a[(k0 + j) * S + col]; // khmm.c:194-206
			}
		double acc[4] = {PSMC_TINY, PSMC_TINY, PSMC_TINY, PSMC_TINY}; // khmm.c:305-306
		// u = 1..L-1 ; index i = u-1 = 0..L-2 ; uses f[i][k], b[i+1][l], obs[i+1]
		const int n = L - 1;
		const int fl_i = lane >> 2, fl_j = lane & 3; // f tile: lane -> (position i, row j)
		double ft = 0.0, bn[BLK]; int sv = 2;
		auto load_blk = [&](int i0, double &ft_, double (&bn_)[BLK], int &sv_) {
			const int ii = min(i0 + fl_i, max(n - 1, 0));
			ft_ = fo[(int64_t)ii * S + k0 + fl_j];
			sv_ = o[min(i0 + min(lane, BLK - 1) + 1, L - 1)];
#pragma unroll
			for (int t = 0; t < BLK; ++t) bn_[t] = bo[(int64_t)min(i0 + t + 1, L - 1) * S + col];
		};
		int i0 = 0;
		if constexpr (S == 64) {
			// full blocks of 32 positions, two register buffers filled alternately one block ahead (~1.3 us of work per block against
			// the latency of the loads); no clamps, no per-position address arithmetic.  What is left goes through the generic loop below.
			constexpr int B2 = 32;
			if (n >= B2) {
				const int fm = lane & 15;
				const double *fq = fo + (int64_t)(fm >> 2) * 64 + k0 + (fm & 3); // chunk c of the block at i: fq[(i + 4c) * 64]
				const double *bq = bo + 64 + col;                                 // b[i+1][l] = bq[i * 64]
				double fA[8], bA[B2], fB[8], bB[B2]; int sA = 2, sB = 2;
				typedef const double __attribute__((address_space(1))) *gptr_t;
				auto load = [&](int i, double (&f_)[8], double (&b_)[B2], int &s_) {
					const int64_t d = (int64_t)i * 64;
#pragma unroll
					for (int g = 0; g < 4; ++g) { // one address per eight rows, the rest in the instruction's offset field
						const double *bg_ = bq + d + 8 * g * 64, *fg_ = fq + d + 8 * g * 64;
						asm volatile("" : "+v"(bg_), "+v"(fg_));
						const gptr_t bg = (gptr_t)bg_, fg = (gptr_t)fg_;
#pragma unroll
						for (int t = 0; t < 8; ++t) b_[8 * g + t] = bg[t * 64];
						f_[2 * g] = fg[0]; f_[2 * g + 1] = fg[4 * 64];
					}
					{
						typedef const uint8_t __attribute__((address_space(1))) *gbptr_t;
						const uint8_t *yp_ = o + i + 1 + (lane & 31);
						asm volatile("" : "+v"(yp_)); // (pinned like the rest: see the E part below)
						s_ = *(gbptr_t)yp_;
					}
					asm volatile("" ::: "memory");  // all of a block's loads ahead of the other block's positions: they may not sink
					__builtin_amdgcn_sched_barrier(0); // below something that might write memory, and the scheduler may not move them either
				};
#define PSMC_EXP1(T, F, B, SV) expect_pos4<T>(acc, F[(T) >> 2], q, B[T], SV);
#define PSMC_EXP8(T, F, B, SV) PSMC_EXP1(T, F, B, SV) PSMC_EXP1(T + 1, F, B, SV) PSMC_EXP1(T + 2, F, B, SV) PSMC_EXP1(T + 3, F, B, SV) \
	PSMC_EXP1(T + 4, F, B, SV) PSMC_EXP1(T + 5, F, B, SV) PSMC_EXP1(T + 6, F, B, SV) PSMC_EXP1(T + 7, F, B, SV)
#define PSMC_EXP32(F, B, SV) PSMC_EXP8(0, F, B, SV) PSMC_EXP8(8, F, B, SV) PSMC_EXP8(16, F, B, SV) PSMC_EXP8(24, F, B, SV)
				load(0, fA, bA, sA);
				for (;;) { // the block in hand is full; the next one is fetched unconditionally (the last fetch repeats the block in
				           // hand): a conditional fetch makes hipcc wait for every outstanding load before the first position
					load(i0 + 2 * B2 <= n ? i0 + B2 : i0, fB, bB, sB);
					PSMC_EXP32(fA, bA, sA)
					i0 += B2;
					if (i0 + B2 > n) break;
					load(i0 + 2 * B2 <= n ? i0 + B2 : i0, fA, bA, sA);
					PSMC_EXP32(fB, bB, sB)
					i0 += B2;
					if (i0 + B2 > n) break;
				}
#undef PSMC_EXP32
#undef PSMC_EXP8
#undef PSMC_EXP1
			}
		}
		if (n > i0) load_blk(i0, ft, bn, sv);
		for (; i0 < n; i0 += BLK) {
			double ft2 = 0.0, bn2[BLK]; int sv2 = 2;
			if (i0 + BLK < n) load_blk(i0 + BLK, ft2, bn2, sv2);
			const int nb = min(BLK, n - i0);
#pragma unroll
			for (int t = 0; t < BLK; ++t) {
				if (t < nb) {
					const int sym = __builtin_amdgcn_readlane(sv, t);
					const double f0 = readlane_f64(ft, 4 * t + 0), f1 = readlane_f64(ft, 4 * t + 1);
					const double f2 = readlane_f64(ft, 4 * t + 2), f3 = readlane_f64(ft, 4 * t + 3);
					double q0, q1, q2, q3;
					if (sym == 0) { q0 = q[0][0]; q1 = q[0][1]; q2 = q[0][2]; q3 = q[0][3]; }
					else if (sym == 1) { q0 = q[1][0]; q1 = q[1][1]; q2 = q[1][2]; q3 = q[1][3]; }
					else { q0 = q[2][0]; q1 = q[2][1]; q2 = q[2][2]; q3 = q[2][3]; }
					const double bl = bn[t];
					acc[0] += f0 * q0 * bl; // khmm.c:316: AA[l] += fuk * q[l] * bu1[l]
					acc[1] += f1 * q1 * bl;
					acc[2] += f2 * q2 * bl;
					acc[3] += f3 * q3 * bl;
				}
			}
			ft = ft2; sv = sv2;
#pragma unroll
			for (int t = 0; t < BLK; ++t) bn[t] = bn2[t];
		}
		double *out = segA + (int64_t)blockIdx.x * (S * S);
#pragma unroll
		for (int j = 0; j < 4; ++j) out[(k0 + j) * S + col] = acc[j];
	} else {
		const int col = lane + 64 * (blockIdx.y - NA);
		double E0 = PSMC_TINY, E1 = PSMC_TINY, E2 = PSMC_TINY; // khmm.c:307-308
		const int n = L - 1; // u = 1..L-1, index i = u-1: f[i], b[i], s[i], obs[i]
		double fu[BLK], bu[BLK], sv = 0.0; int symv = 2;
		auto load_blk = [&](int i0, double (&fu_)[BLK], double (&bu_)[BLK], double &sv_, int &symv_) {
			const int ii = min(i0 + min(lane, BLK - 1), L - 1);
			sv_ = so[ii]; symv_ = o[ii];
#pragma unroll
			for (int t = 0; t < BLK; ++t) {
				const int64_t r = (int64_t)min(i0 + t, L - 1) * S + col;
				fu_[t] = fo[r]; bu_[t] = bo[r];
			}
		};
		int i0 = 0;
		if constexpr (S == 64) { // full blocks of 16 positions, as in the A part above
			if (n >= BLK) {
				typedef const double __attribute__((address_space(1))) *gptr_t;
				double fA[BLK], bA[BLK], fB[BLK], bB[BLK], ssA = 0.0, ssB = 0.0; int yA = 2, yB = 2;
				auto load = [&](int i, double (&f_)[BLK], double (&b_)[BLK], double &ss_, int &y_) {
					const int64_t d = (int64_t)i * 64 + col;
#pragma unroll
					for (int g = 0; g < 2; ++g) {
						const double *fg_ = fo + d + 8 * g * 64, *bg_ = bo + d + 8 * g * 64;
						asm volatile("" : "+v"(fg_), "+v"(bg_));
						const gptr_t fg = (gptr_t)fg_, bg = (gptr_t)bg_;
#pragma unroll
						for (int t = 0; t < 8; ++t) { f_[8 * g + t] = fg[t * 64]; b_[8 * g + t] = bg[t * 64]; }
					}
					{ // (through pinned pointers as well: loads from a const __restrict__ argument are invariant to hipcc, which sinks them
					  //  past the memory clobber into the block that uses them -- behind the other block's sixteen positions)
						typedef const uint8_t __attribute__((address_space(1))) *gbptr_t;
						const double *sp_ = so + i + (lane & 15); const uint8_t *yp_ = o + i + (lane & 15);
						asm volatile("" : "+v"(sp_), "+v"(yp_));
						ss_ = *(gptr_t)sp_; y_ = *(gbptr_t)yp_;
					}
					asm volatile("" ::: "memory");
					__builtin_amdgcn_sched_barrier(0);
				};
#define PSMC_EE1(T, F, B, SS, Y) expect_posE<T>(E0, E1, E2, F[T], B[T], SS, Y);
#define PSMC_EE16(F, B, SS, Y)                                                                                              \
	PSMC_EE1(0, F, B, SS, Y) PSMC_EE1(1, F, B, SS, Y) PSMC_EE1(2, F, B, SS, Y) PSMC_EE1(3, F, B, SS, Y) PSMC_EE1(4, F, B, SS, Y)    \
	PSMC_EE1(5, F, B, SS, Y) PSMC_EE1(6, F, B, SS, Y) PSMC_EE1(7, F, B, SS, Y) PSMC_EE1(8, F, B, SS, Y) PSMC_EE1(9, F, B, SS, Y)    \
	PSMC_EE1(10, F, B, SS, Y) PSMC_EE1(11, F, B, SS, Y) PSMC_EE1(12, F, B, SS, Y) PSMC_EE1(13, F, B, SS, Y)                       \
	PSMC_EE1(14, F, B, SS, Y) PSMC_EE1(15, F, B, SS, Y)
				load(0, fA, bA, ssA, yA);
				for (;;) {
					load(i0 + 2 * BLK <= n ? i0 + BLK : i0, fB, bB, ssB, yB);
					PSMC_EE16(fA, bA, ssA, yA)
					i0 += BLK;
					if (i0 + BLK > n) break;
					load(i0 + 2 * BLK <= n ? i0 + BLK : i0, fA, bA, ssA, yA);
					PSMC_EE16(fB, bB, ssB, yB)
					i0 += BLK;
					if (i0 + BLK > n) break;
				}
#undef PSMC_EE16
#undef PSMC_EE1
			}
		}
		if (n > i0) load_blk(i0, fu, bu, sv, symv);
		for (; i0 < n; i0 += BLK) {
			double fu2[BLK], bu2[BLK], sv2 = 0.0; int symv2 = 2;
			if (i0 + BLK < n) load_blk(i0 + BLK, fu2, bu2, sv2, symv2);
			const int nb = min(BLK, n - i0);
#pragma unroll
			for (int t = 0; t < BLK; ++t) {
				if (t < nb) {
					const int sym = __builtin_amdgcn_readlane(symv, t);
					const double ss = readlane_f64(sv, t);
					const double v = fu[t] * bu[t] * ss; // khmm.c:317: Ec[k] += fuk * bu[k] * ss
					if (sym == 0) E0 += v; else if (sym == 1) E1 += v; else E2 += v;
				}
			}
			sv = sv2; symv = symv2;
#pragma unroll
			for (int t = 0; t < BLK; ++t) { fu[t] = fu2[t]; bu[t] = bu2[t]; }
		}
		double *oe = segE + (int64_t)blockIdx.x * (3 * S);
		oe[col] = E0; oe[S + col] = E1; oe[2 * S + col] = E2;
		const int sym1 = o[0]; // khmm.c:321-322: A0[l] += a0[l]*e[o_1][l]*b[1][l], A0 starts at 0
		segA0[(int64_t)blockIdx.x * S + col] = 0.0 + a0[col] * e[sym1 * S + col] * bo[col];
	}
}

// ---------------------------------------------------------------- expect with the forward sweep recomputed (exact batch, 64 states)
// VERDICT r3 item 3(a).  A batch group is as many replicates as fit the tables, and f + b + s cost 1032 bytes per bin: 12-14
// replicates of a 30 M-bin genome in 250 GB, i.e. ~550 waves in the latency-bound sweeps on a device with 1024 SIMDs.  hmm_expect
// adds A[k][l] += f[u][k] ae b[u+1][l] for u ASCENDING (khmm.c:310-319), the order in which the forward sweep produces f -- so
// the third pass can recompute f instead of reading it back: the recursion is deterministic, the same instructions in the same
// order give the same bits, and the f table (half of the memory) is never written.  One work-group per (replicate, segment):
//   wave 0  the forward recursion of k_fwd_exact, verbatim, its vector going into a two-half LDS ring of 16 positions each;
//   wave 1  rows k = 0..31 of A (lane = column l) + E and A0;   wave 2  rows k = 32..63 of A.
// The consumers work on the half the producer filled in the previous phase: one barrier per 16 positions.  They are far below
// the producer's 0.8 us per position (3 instructions per cell and position against ~400 per position), so the pass runs at the
// forward sweep's latency: group time (fwd s[] only) + bwd + this = 0.83 + 0.53 + 0.83 us per bin of the longest segment instead
// of 0.83 + 0.53 + 0.45, for twice the replicates per group.  q = e[o][l] a[k][l] is formed here with one rounding, as
// hmm_pre_backward (khmm.c:194-206) and the host's aeT do.
template <int REP>
__global__ __launch_bounds__(192, 2) void k_expect_exact_rf(const double *__restrict__ a, const double *__restrict__ e,
                                                          const double *__restrict__ a0, const uint8_t *__restrict__ obs,
                                                          const int64_t *__restrict__ seg_off, const int32_t *__restrict__ seg_len,
                                                          const ExWork wl, const double *__restrict__ b, const double *__restrict__ s,
                                                          double *__restrict__ segA, double *__restrict__ segE, double *__restrict__ segA0)
{
	constexpr int RB = 16; // positions per ring half
	__shared__ double ring[2][RB][64];
	const int lane = threadIdx.x & 63;
	const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	const int seg = wl.seg[blockIdx.x];
	if (seg < 0) return; // padding entry of a batch (block-uniform)
	const int64_t off = seg_off[seg], toff = wl.tab ? wl.tab[blockIdx.x] : off;
	{ const int64_t po = wl.par ? wl.par[blockIdx.x] * wl.par_stride : 0; a += po; e += po; a0 += po; }
	const int L = seg_len[seg];
	const int n = L - 1;                 // u = 1 .. L-1: index i = u - 1 = 0 .. n-1 uses f[i], b[i+1], obs[i+1] (A) and f[i], b[i], s[i], obs[i] (E)
	const int P = (n + RB - 1) / RB;     // phases: every wave passes exactly P barriers
	const uint8_t *o = obs + off;
	const double e0 = e[lane], e1 = e[64 + lane];
	if (w == 0) { // ---------------- producer: k_fwd_exact, position by position (khmm.c:171-185)
		double col[64];
#pragma unroll
		for (int l = 0; l < 64; ++l) col[l] = a[l * 64 + lane];
		int symv = o[lane];
		double x = 0.0;
		for (int base = 0; base < n; base += 64) { // positions base+1 .. base+64 = indices base .. base+63
			const int nb = min(64, n - base);
			const int symn = (base + 64 < L) ? (int)o[base + 64 + lane] : 2;
			for (int i = 0; i < nb; ++i) {
				const int sym = __builtin_amdgcn_readlane(symv, i);
				double g, sum;
				if (base + i == 0) g = a0[lane] * pick_e(sym, e0, e1);
				else { double r[4]; rep_rows<REP>(x, r); const double tmp = xdot64(r, col); g = pick_e(sym, e0, e1) * tmp; }
				{ double q[4]; rep_rows<REP>(g, q); sum = seq_sum_rep(q); }
				x = g / sum;
				const int idx = base + i;
				ring[(idx / RB) & 1][idx % RB][lane] = x;
				if (idx % RB == RB - 1 || idx == n - 1) __syncthreads(); // the half is complete: hand it to the consumers
			}
			symv = symn;
		}
		return;
	}
	// ---------------- consumers
	const int cw = w - 1, k0 = 32 * cw, colc = lane; // rows k0 .. k0+31 of A, lane = column l
	const double *bo = b + toff * 64, *so = s + (wl.tab_s ? wl.tab_s[blockIdx.x] : toff);
	// q0 = ae[0][k][l] = e[0][l] * a[k][l] with its one rounding (khmm.c:194-206), kept for the homozygous symbol -- nearly every
	// position; the other two symbols form the product per position (e[2][l] = 1.0: the product IS a[k][l])
	// (rows of `a` re-read from the cache there: 32 more registers would spill)
	double q0[32], acc[32];
#pragma unroll
	for (int j = 0; j < 32; ++j) { q0[j] = e0 * a[(k0 + j) * 64 + colc]; acc[j] = PSMC_TINY; } // khmm.c:305-306
	double E0 = PSMC_TINY, E1 = PSMC_TINY, E2 = PSMC_TINY; // khmm.c:307-308 (wave 1 only)
	double bprev = n > 0 ? bo[colc] : 0.0;                  // b[i] of the block's first position (E uses b[i], A uses b[i+1])
	double bn[RB]; int sv = 2, sve = 2; double ssv = 0.0;
	auto load_blk = [&](int i0, double (&bn_)[RB], int &sv_, int &sve_, double &ss_) {
#pragma unroll
		for (int t = 0; t < RB; ++t) bn_[t] = bo[(int64_t)min(i0 + t + 1, L - 1) * 64 + colc];
		const int ii = min(i0 + min(lane, RB - 1), L - 1);
		sv_ = o[min(ii + 1, L - 1)];   // symbol of position u + 1 (A)
		sve_ = o[ii];                  // symbol of position u (E)
		ss_ = so[ii];
	};
	for (int ph = 0; ph < P; ++ph) {
		const int i0 = ph * RB, nb = min(RB, n - i0);
		// (no double buffer for the b rows: two work-groups per compute unit need the wave below 256 registers, and the consumers have
		// ~14 us per phase for ~5 us of work -- the load latency is paid in their slack, not on the producer's path)
		load_blk(i0, bn, sv, sve, ssv);
		__syncthreads(); // the producer has filled half ph & 1
		const double (*hf)[64] = ring[ph & 1];
#pragma unroll
		for (int t = 0; t < RB; ++t) {
			if (t < nb) {
				const int sym = __builtin_amdgcn_readlane(sv, t);
				const double bl = bn[t];
				if (sym == 0) {
#pragma unroll
					for (int j = 0; j < 32; ++j) acc[j] += hf[t][k0 + j] * q0[j] * bl; // khmm.c:316
				} else {
					const double es = sym == 1 ? e1 : 1.0;    // e[o_{u+1}][l]
					typedef const double __attribute__((address_space(1))) *gptr_t;
					const double *ar_ = a + k0 * 64 + colc;
					asm volatile("" : "+v"(ar_));             // loop invariant, but NOT to be hoisted into registers
					const gptr_t ar = (gptr_t)ar_;
#pragma unroll
					for (int j = 0; j < 32; ++j) {
						const double q = es * ar[j * 64];     // ae[o][k][l]: one rounding (khmm.c:194-206)
						acc[j] += hf[t][k0 + j] * q * bl;
					}
				}
				if (cw == 0) { // khmm.c:317: Ec[k] += f[u][k] * b[u][k] * s[u]
					const int syme = __builtin_amdgcn_readlane(sve, t);
					const double v = hf[t][colc] * bprev * readlane_f64(ssv, t);
					if (syme == 0) E0 += v; else if (syme == 1) E1 += v; else E2 += v;
				}
				bprev = bl;
			}
		}
	}
	double *out = segA + (int64_t)blockIdx.x * 4096;
#pragma unroll
	for (int j = 0; j < 32; ++j) out[(k0 + j) * 64 + colc] = acc[j];
	if (cw == 0) {
		double *oe = segE + (int64_t)blockIdx.x * 192;
		oe[colc] = E0; oe[64 + colc] = E1; oe[128 + colc] = E2;
		const int sym1 = o[0]; // khmm.c:321-322
		segA0[(int64_t)blockIdx.x * 64 + colc] = 0.0 + a0[colc] * e[sym1 * 64 + colc] * bo[colc];
	}
}

// Two entries per work-group (round 4, after the instruction trimming above).  With one entry per group of three waves a compute
// unit holds two entries -- eight wave slots at 256 registers -- and the pass runs at 512 entries x one producer step per ~1 us
// with the vector pipes half idle: a consumer wave needs ~110 vector instructions per position, the producer ~280.  Here a group
// is TWO producers (entries 2b, 2b+1 of the list: same replicate, same parameters -- the batch pads every replicate to four
// entries) and two consumers that serve both, entry after entry out of a ring per entry: wave 2 rows 0..31 of A, wave 3 rows
// 32..63.  The emission counts and A0 moved to the producers (lane = state there, and the wave has f[u][k] and s[u] in hand).
// Four entries per compute unit, every SIMD a producer and a consumer: the pass becomes bound by the vector pipe.
// Same instructions in the same order per entry as k_expect_exact_rf: bit-identical.
template <int REP>
__global__ __launch_bounds__(256, 2) void k_expect_exact_rf2(const double *__restrict__ a, const double *__restrict__ e,
                                                           const double *__restrict__ a0, const uint8_t *__restrict__ obs,
                                                           const int64_t *__restrict__ seg_off, const int32_t *__restrict__ seg_len,
                                                           const ExWork wl, const double *__restrict__ b, const double *__restrict__ s,
                                                           double *__restrict__ segA, double *__restrict__ segE, double *__restrict__ segA0,
                                                           int *__restrict__ cu_mask)
{
	constexpr int RB = 16, CB = 4; // positions per ring half; b rows a consumer fetches at a time
	__shared__ double ring[2][2][RB][64]; // [entry][half][position][state]
	__shared__ int roles[8];
	const int lane = threadIdx.x & 63;
	const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	// Which wave does what.  The four waves of a group sit on the four SIMDs of a compute unit, and so do those of the second resident
	// group: with fixed roles wave i of both lands on SIMD i -- two producers (2 x ~280 vector instructions per position) on two of
	// the SIMDs, two consumers on the others (measured: 1.6 us per position where the mixed placement needs ~1).  So a wave takes
	// its role from the SIMD it runs on and a bit t the group claims in a word of its compute unit (the first resident group gets
	// t = 0, the second t = 1): producer iff (SIMD + t) is even, entry / row half = SIMD >> 1.  Every SIMD then holds one producer
	// and one consumer.  Without the word, or when the waves do not sit on four different SIMDs: roles by wave number.
	int w = wv, held = 0, key = 0;
	if (cu_mask) {
		unsigned hid, xcc;
		asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hid));  // [5:4] SIMD, [11:8] CU, [12] SH, [15:13] SE
		asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); // [3:0] XCC
		const int simd = (int)((hid >> 4) & 3u);
		if (lane == 0) roles[wv] = simd;
		if (threadIdx.x == 0) {
			key = (int)(((xcc & 15u) << 8) | ((hid >> 8) & 0xffu));
			int t;
			if (!(atomicOr(&cu_mask[key], 1) & 1)) { t = 0; held = 1; }
			else if (!(atomicOr(&cu_mask[key], 2) & 2)) { t = 1; held = 2; }
			else t = (int)(blockIdx.x & 1); // (more groups on the unit than the two this kernel's registers allow: not expected)
			roles[4] = t;
		}
		__syncthreads();
		const int seen = (1 << roles[0]) | (1 << roles[1]) | (1 << roles[2]) | (1 << roles[3]);
		if (seen == 15) w = ((simd + roles[4]) & 1) == 0 ? (simd >> 1) : 2 + (simd >> 1);
	}
	w = __builtin_amdgcn_readfirstlane(w);
	auto release = [&]() { if (held) atomicAnd(&cu_mask[key], ~held); }; // (thread 0 only holds one)
	const int ent = 2 * (int)blockIdx.x;
	{ const int64_t po = wl.par ? wl.par[ent] * wl.par_stride : 0; a += po; e += po; a0 += po; }
	// segment of the two entries (-1: padding / past the list) and their positions u = 1 .. L-1 (index i = u-1 = 0 .. n-1)
	const int sg0 = wl.seg[ent], sg1 = ent + 1 < wl.n ? wl.seg[ent + 1] : -1;
	const int n0 = sg0 >= 0 ? seg_len[sg0] - 1 : -1, n1 = sg1 >= 0 ? seg_len[sg1] - 1 : -1;
	const int P = (max(max(n0, n1), 0) + RB - 1) / RB; // phases: every wave passes exactly P barriers
	const double e0 = e[lane], e1 = e[64 + lane];
	if (w < 2) { // ---------------- producer of entry w: k_fwd_exact, position by position (khmm.c:171-185), + E and A0 of the entry
		const int sg = w == 0 ? sg0 : sg1, n = w == 0 ? n0 : n1, L = n + 1;
		const int64_t off = sg >= 0 ? seg_off[sg] : 0, toff = sg >= 0 && wl.tab ? wl.tab[ent + w] : off;
		const uint8_t *o = obs + off;
		const double *bo = b + toff * 64 + lane, *so = s + (sg >= 0 && wl.tab_s ? wl.tab_s[ent + w] : toff);
		double col[64];
#pragma unroll
		for (int l = 0; l < 64; ++l) col[l] = a[l * 64 + lane];
		double x = 0.0, E0 = PSMC_TINY, E1 = PSMC_TINY, E2 = PSMC_TINY; // khmm.c:307-308
		int symn = n > 0 ? (int)o[lane & 15] : 2; // symbols of the positions of phase 0 (obs is padded by >= 64 bytes)
		for (int ph = 0; ph < P; ++ph) {
			const int i0 = ph * RB, nb = min(RB, n - i0);
			const int symv = symn;
			double bq[RB], ssv = 0.0; // b[u][k] and s[u] of this phase's positions: fetched now, used after the last of them
			if (nb > 0) {
#pragma unroll
				for (int t = 0; t < RB; ++t) bq[t] = bo[(int64_t)min(i0 + t, L - 1) * 64];
				ssv = so[min(i0 + (lane & 15), L - 1)];
				if (i0 + RB < n) symn = (int)o[i0 + RB + (lane & 15)];
			}
			for (int t = 0; t < nb; ++t) {
				const int sym = __builtin_amdgcn_readlane(symv, t);
				double g, sum;
				if (i0 + t == 0) g = a0[lane] * pick_e(sym, e0, e1);
				else { double r[4]; rep_rows<REP>(x, r); const double tmp = xdot64(r, col); g = pick_e(sym, e0, e1) * tmp; }
				{ double q[4]; rep_rows<REP>(g, q); sum = seq_sum_rep(q); }
				x = g / sum;
				ring[w][ph & 1][t][lane] = x;
			}
			if (nb > 0) { // khmm.c:317: Ec[o_u][k] += f[u][k] * b[u][k] * s[u], u ascending
#define PSMC_PE(T) if (T < nb) expect_posE<T>(E0, E1, E2, ring[w][ph & 1][T][lane], bq[T], ssv, symv);
				PSMC_PE(0) PSMC_PE(1) PSMC_PE(2) PSMC_PE(3) PSMC_PE(4) PSMC_PE(5) PSMC_PE(6) PSMC_PE(7)
				PSMC_PE(8) PSMC_PE(9) PSMC_PE(10) PSMC_PE(11) PSMC_PE(12) PSMC_PE(13) PSMC_PE(14) PSMC_PE(15)
#undef PSMC_PE
			}
			__syncthreads(); // the halves of this phase are complete: hand them to the consumers
		}
		if (sg >= 0) {
			double *oe = segE + (int64_t)(ent + w) * 192;
			oe[lane] = E0; oe[64 + lane] = E1; oe[128 + lane] = E2;
			const int sym1 = o[0]; // khmm.c:321-322
			segA0[(int64_t)(ent + w) * 64 + lane] = 0.0 + a0[lane] * e[sym1 * 64 + lane] * bo[0];
		}
		release();
		return;
	}
	// ---------------- consumers: rows k0 .. k0+31 of A of BOTH entries, lane = column l
	// 64 sums + 32 products e*a of the homozygous symbol fill the register file, and a choice between two instruction sequences that
	// both update the sums makes hipcc copy all of them at every position.  So four rows of a position are ONE asm block: the LDS
	// reads of the forward vector, the choice of the emission row (the other two symbols read e*a -- the host's aeT, the same
	// single-rounding product -- from memory) and the twelve instructions of khmm.c:316 (products left to right, then the sum).
	const int cw = w - 2, k0 = 32 * cw, colc = lane;
	double q0[32], acc[2][32];
#pragma unroll
	for (int j = 0; j < 32; ++j) { q0[j] = e0 * a[(k0 + j) * 64 + colc]; acc[0][j] = PSMC_TINY; acc[1][j] = PSMC_TINY; } // khmm.c:194-206, 305-306
	const double *ap1 = a + 2 * 4096 + colc * 64 + k0, *ap2 = ap1 + 4096; // aeT[1], aeT[2]: [l * 64 + k] (api.hip fill_params)
	asm volatile("" : "+v"(ap1), "+v"(ap2));
	// One position = eight asm blocks of four rows.  Operands: 0-3 sums | 4-7 f of this block's rows (then the products) | 8-11 f of the
	// NEXT block's rows | 12-15 e*a of a rare symbol | 16-19 e*a of symbol 0 | 20 b[u+1][l] | 21 LDS byte address of f[u][k0] (wave-uniform:
	// broadcast reads) | 22 / 23 this lane's rows of aeT[1] / aeT[2] | 24 symbol | 25 byte offset of the block's rows.  The LDS reads of
	// block k+1 are issued before block k computes (a read costs ~100 cycles, and eight exposed ones per position made the consumers the
	// slowest waves of the group): registers 8-11 leave a block with their loads IN FLIGHT and enter the next one as its 4-7 -- only
	// asm statements stand between (tests/test_abi.py checks the compiled code for that), and each waits (lgkmcnt counts down in issue
	// order for LDS; foreign scalar loads in the count only make the wait longer) before it touches them.  The blocks read LDS the
	// compiler knows nothing about: they are `asm volatile` with a "memory" clobber, so that none of them can be moved across the
	// barrier that orders the producers' writes against these reads, duplicated, or dropped.
#define PSMC_C4_TAIL                                                                                                                  \
	      "s_cmp_eq_u32 %24, 0\n\ts_cbranch_scc0 .Lc4s_%=\n\t"                                                                            \
	      "v_mul_f64 %4, %4, %16\n\tv_mul_f64 %5, %5, %17\n\tv_mul_f64 %6, %6, %18\n\tv_mul_f64 %7, %7, %19\n"                                \
	      ".Lc4j_%=:\n\t"                                                                                                                \
	      "v_mul_f64 %4, %4, %20\n\tv_mul_f64 %5, %5, %20\n\tv_mul_f64 %6, %6, %20\n\tv_mul_f64 %7, %7, %20\n\t"                            \
	      "v_add_f64 %0, %0, %4\n\tv_add_f64 %1, %1, %5\n\tv_add_f64 %2, %2, %6\n\tv_add_f64 %3, %3, %7\n\t"                                \
	      "s_branch .Lc4e_%=\n"                                                                                                          \
	      ".Lc4s_%=:\n\t"                                                                                                                \
	      "s_cmp_eq_u32 %24, 1\n\ts_cbranch_scc0 .Lc4t_%=\n\t"                                                                            \
	      "global_load_dwordx2 %12, %22, off offset:%25\n\tglobal_load_dwordx2 %13, %22, off offset:%25+8\n\t"                             \
	      "global_load_dwordx2 %14, %22, off offset:%25+16\n\tglobal_load_dwordx2 %15, %22, off offset:%25+24\n\t"                         \
	      "s_branch .Lc4w_%=\n"                                                                                                          \
	      ".Lc4t_%=:\n\t"                                                                                                                \
	      "global_load_dwordx2 %12, %23, off offset:%25\n\tglobal_load_dwordx2 %13, %23, off offset:%25+8\n\t"                             \
	      "global_load_dwordx2 %14, %23, off offset:%25+16\n\tglobal_load_dwordx2 %15, %23, off offset:%25+24\n"                           \
	      ".Lc4w_%=:\n\t"                                                                                                                \
	      "s_waitcnt vmcnt(0) lgkmcnt(0)\n\t"                                                                                             \
	      "v_mul_f64 %4, %4, %12\n\tv_mul_f64 %5, %5, %13\n\tv_mul_f64 %6, %6, %14\n\tv_mul_f64 %7, %7, %15\n\t"                            \
	      "s_branch .Lc4j_%=\n"                                                                                                          \
	      ".Lc4e_%=:"                                                                                                                      \

#define PSMC_C4_IN(A, O, BL, AD, SYM)                                                                                                 \
	        "=&v"(g0), "=&v"(g1), "=&v"(g2), "=&v"(g3)                                                                                     \
	      : "v"(q0[O]), "v"(q0[O + 1]), "v"(q0[O + 2]), "v"(q0[O + 3]), "v"(BL), "v"(AD), "v"(ap1), "v"(ap2), "s"(SYM), "n"(8 * (O))       \
	      : "scc", "memory")
	// first block of a position: reads its own rows and the next block's
#define PSMC_C4F(A, BL, AD, SYM, C0, C1, C2, C3, N0, N1, N2, N3)                                                                     \
	{ double g0, g1, g2, g3;                                                                                                           \
	  asm volatile("s_waitcnt lgkmcnt(0)\n\t"                                                                                           \
	      "ds_read_b64 %4, %21\n\tds_read_b64 %5, %21 offset:8\n\tds_read_b64 %6, %21 offset:16\n\tds_read_b64 %7, %21 offset:24\n\t"       \
	      "ds_read_b64 %8, %21 offset:32\n\tds_read_b64 %9, %21 offset:40\n\tds_read_b64 %10, %21 offset:48\n\tds_read_b64 %11, %21 offset:56\n\t" \
	      "s_waitcnt lgkmcnt(4)\n\t"                                                                                                    \
	      PSMC_C4_TAIL                                                                                                                  \
	      : "+v"(A[0]), "+v"(A[1]), "+v"(A[2]), "+v"(A[3]), "=&v"(C0), "=&v"(C1), "=&v"(C2), "=&v"(C3),                                    \
	        "=&v"(N0), "=&v"(N1), "=&v"(N2), "=&v"(N3),                                                                                    \
	        PSMC_C4_IN(A, 0, BL, AD, SYM); }
	// blocks 1..6: rows O..O+3 arrive in C0..C3; reads the rows of block O/4 + 1 (LDS offset NO = 8 * (O + 4)) into N0..N3
#define PSMC_C4M(A, O, NO, BL, AD, SYM, C0, C1, C2, C3, N0, N1, N2, N3)                                                              \
	{ double g0, g1, g2, g3;                                                                                                           \
	  asm volatile("ds_read_b64 %8, %21 offset:" #NO "\n\tds_read_b64 %9, %21 offset:" #NO "+8\n\t"                                                 \
	      "ds_read_b64 %10, %21 offset:" #NO "+16\n\tds_read_b64 %11, %21 offset:" #NO "+24\n\t"                                           \
	      "s_waitcnt lgkmcnt(4)\n\t"                                                                                                    \
	      PSMC_C4_TAIL                                                                                                                  \
	      : "+v"(A[O]), "+v"(A[O + 1]), "+v"(A[O + 2]), "+v"(A[O + 3]), "+v"(C0), "+v"(C1), "+v"(C2), "+v"(C3),                            \
	        "=&v"(N0), "=&v"(N1), "=&v"(N2), "=&v"(N3),                                                                                    \
	        PSMC_C4_IN(A, O, BL, AD, SYM); }
	// last block: nothing to read ahead (operands 8-11 are placeholders)
#define PSMC_C4L(A, O, BL, AD, SYM, C0, C1, C2, C3)                                                                                  \
	{ double g0, g1, g2, g3, z0, z1, z2, z3;                                                                                           \
	  asm volatile("s_waitcnt lgkmcnt(0)\n\t"                                                                                                    \
	      PSMC_C4_TAIL                                                                                                                  \
	      : "+v"(A[O]), "+v"(A[O + 1]), "+v"(A[O + 2]), "+v"(A[O + 3]), "+v"(C0), "+v"(C1), "+v"(C2), "+v"(C3),                            \
	        "=&v"(z0), "=&v"(z1), "=&v"(z2), "=&v"(z3),                                                                                    \
	        PSMC_C4_IN(A, O, BL, AD, SYM); }
#define PSMC_CPOS(A, BL, AD, SYM)                                                                                                     \
	{ double fa0, fa1, fa2, fa3, fb0, fb1, fb2, fb3;                                                                                   \
	  PSMC_C4F(A, BL, AD, SYM, fa0, fa1, fa2, fa3, fb0, fb1, fb2, fb3)                                                                 \
	  PSMC_C4M(A, 4, 64, BL, AD, SYM, fb0, fb1, fb2, fb3, fa0, fa1, fa2, fa3)                                                          \
	  PSMC_C4M(A, 8, 96, BL, AD, SYM, fa0, fa1, fa2, fa3, fb0, fb1, fb2, fb3)                                                          \
	  PSMC_C4M(A, 12, 128, BL, AD, SYM, fb0, fb1, fb2, fb3, fa0, fa1, fa2, fa3)                                                        \
	  PSMC_C4M(A, 16, 160, BL, AD, SYM, fa0, fa1, fa2, fa3, fb0, fb1, fb2, fb3)                                                        \
	  PSMC_C4M(A, 20, 192, BL, AD, SYM, fb0, fb1, fb2, fb3, fa0, fa1, fa2, fa3)                                                        \
	  PSMC_C4M(A, 24, 224, BL, AD, SYM, fa0, fa1, fa2, fa3, fb0, fb1, fb2, fb3)                                                        \
	  PSMC_C4L(A, 28, BL, AD, SYM, fb0, fb1, fb2, fb3) }
	typedef const double __attribute__((address_space(3))) *lptr_t;
	for (int ph = 0; ph < P; ++ph) {
		__syncthreads(); // the producers have filled half ph & 1
		const int i0 = ph * RB;
#pragma unroll
		for (int q = 0; q < 2; ++q) {
			const int n = q == 0 ? n0 : n1, nb = min(RB, n - i0);
			if (nb > 0) { // (wave-uniform)
				const int sg = q == 0 ? sg0 : sg1, L = n + 1;
				const int64_t off = seg_off[sg], toff = wl.tab ? wl.tab[ent + q] : off;
				const double *bo = b + toff * 64 + colc;
				const int sv = obs[off + min(i0 + min(lane, RB - 1) + 1, L - 1)]; // symbols of positions u + 1
				const unsigned hbase = (unsigned)(uintptr_t)(lptr_t)&ring[q][ph & 1][0][k0];
#pragma unroll
				for (int c = 0; c < RB / CB; ++c) {
					double bn[CB];
#pragma unroll
					for (int t = 0; t < CB; ++t) bn[t] = bo[(int64_t)min(i0 + CB * c + t + 1, L - 1) * 64];
#pragma unroll
					for (int tt = 0; tt < CB; ++tt) {
						const int t = CB * c + tt;
						if (t < nb) {
							const int sym = __builtin_amdgcn_readlane(sv, t);
							const double bl = bn[tt];
							const unsigned ad = hbase + (unsigned)t * 512u;
							PSMC_CPOS(acc[q], bl, ad, sym)
						}
					}
				}
			}
		}
	}
#undef PSMC_CPOS
#undef PSMC_C4L
#undef PSMC_C4M
#undef PSMC_C4F
#undef PSMC_C4_IN
#undef PSMC_C4_TAIL
#pragma unroll
	for (int q = 0; q < 2; ++q) {
		if ((q == 0 ? sg0 : sg1) >= 0) {
			double *out = segA + (int64_t)(ent + q) * 4096;
#pragma unroll
			for (int j = 0; j < 32; ++j) out[(k0 + j) * 64 + colc] = acc[q][j];
		}
	}
	release();
}

// ---------------------------------------------------------------- posterior decoding
// hmm_post_decode (khmm.c:264-281) on the tables of one segment: path[u] = argmax_k
// f[u][k]*b[u][k]*s[u] with the FIRST maximum winning (the reference compares with `<`),
// maxp[u] = that posterior.  One wave per 64 positions, lane = state (and state+64).
template <int S>
__global__ __launch_bounds__(64) void k_post_decode(const double *__restrict__ f, const double *__restrict__ b,
                                                      const double *__restrict__ s, int64_t off, int L, int n,
                                                      int32_t *__restrict__ path, double *__restrict__ maxp)
{
	const int lane = threadIdx.x;
	const int u0 = blockIdx.x * 64, u1 = min(L, u0 + 64);
	for (int u = u0; u < u1; ++u) {
		const int64_t g = off + u;
		double v = lane < n ? f[g * S + lane] * b[g * S + lane] * s[g] : -1.0;
		int k = lane;
		if constexpr (S == 128) {
			const int k2 = lane + 64;
			const double v2 = k2 < n ? f[g * S + k2] * b[g * S + k2] * s[g] : -1.0;
			if (v2 > v) { v = v2; k = k2; }
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

int launch_post_decode(hipStream_t st, const double *f, const double *b, const double *s, int64_t off, int L, int n,
                       int ns, int32_t *path, double *maxp)
{
	if (ns == 128)
		hipLaunchKernelGGL(k_post_decode<128>, dim3((L + 63) / 64), dim3(64), 0, st, f, b, s, off, L, n, path, maxp);
	else
		hipLaunchKernelGGL(k_post_decode<64>, dim3((L + 63) / 64), dim3(64), 0, st, f, b, s, off, L, n, path, maxp);
	return (int)hipGetLastError();
}

template <int S> __device__ __forceinline__ double ordered_sum_states(double t0, double t1);

// Full posterior decoding of one segment (psmc_decode's -D branch, aux.c:183-200) on the resident tables:
//   post[u][l]  = f[u][l]*b[u][l]*s[u]                                         (hmm_post_state, khmm.c:285-292)
//   recomb[u]   = 1 - sum_l f[u][l]*a[l][l]*b[u+1][l]*e[o_{u+1}][l]  (u < L),  0 at u = L      (aux.c:189-193)
// every product left to right and the sum in state order like the reference's loop, so the doubles are the
// reference's.  One wave per 64 positions, lane = state (S == 128: states 2*lane, 2*lane+1).
template <int S>
__global__ __launch_bounds__(64) void k_post_full(const double *__restrict__ a, const double *__restrict__ e,
                                                    const uint8_t *__restrict__ obs, const double *__restrict__ f,
                                                    const double *__restrict__ b, const double *__restrict__ s, int64_t off,
                                                    int L, int n, double *__restrict__ post, double *__restrict__ recomb)
{
	constexpr int PER = S / 64;
	const int lane = threadIdx.x;
	const int u0 = blockIdx.x * 64, u1 = min(L, u0 + 64);
	double dg[PER];
#pragma unroll
	for (int j = 0; j < PER; ++j) { const int k = PER * lane + j; dg[j] = a[(int64_t)k * S + k]; }
	for (int u = u0; u < u1; ++u) {
		const int64_t g = off + u;
		const double ss = s[g];
		double fu[PER], t[PER] = {};
#pragma unroll
		for (int j = 0; j < PER; ++j) {
			const int k = PER * lane + j;
			fu[j] = f[g * S + k];
			if (post && k < n) post[(int64_t)u * n + k] = fu[j] * b[g * S + k] * ss;
		}
		if (!recomb) continue;
		double pr = 0.0;
		if (u < L - 1) {
			const int sym = obs[g + 1];
#pragma unroll
			for (int j = 0; j < PER; ++j) {
				const int k = PER * lane + j;
				t[j] = fu[j] * dg[j] * b[(g + 1) * S + k] * e[sym * S + k]; // fu[l] * a[l][l] * bu1[l] * eu1[l]
			}
			const double sm = ordered_sum_states<S>(t[0], PER > 1 ? t[PER - 1] : 0.0);
			pr = sm != sm ? sm : 1.0 - sm; // (a NaN keeps its sign through x86's subsd; the source modifier of v_add_f64 would flip it: "nan" for the reference's "-nan" in a run whose parameters have already diverged)
		}
		if (lane == 0) recomb[u] = pr;
	}
}

// Posterior-weighted counts (psmc_decode's -c branch, aux.c:202-219): cnt[l][j] += post[u][l] * cnt1[u][j] for
// u = 1..min_l IN POSITION ORDER, continuing the caller's running totals (the reference keeps one accumulator per
// (state, column) across all segments).  One wave per count column j, lane = state; 16 positions are loaded at a
// time, the accumulation itself is sequential.
template <int S>
__global__ __launch_bounds__(64) void k_post_counts(const double *__restrict__ f, const double *__restrict__ b,
                                                      const double *__restrict__ s, int64_t off, int min_l,
                                                      const int32_t *__restrict__ cnt1, int n_cnt, int n,
                                                      double *__restrict__ cnt)
{
	constexpr int PER = S / 64, BLK = 16;
	const int lane = threadIdx.x, j = blockIdx.x;
	double acc[PER];
#pragma unroll
	for (int q = 0; q < PER; ++q) { const int k = PER * lane + q; acc[q] = k < n ? cnt[(int64_t)k * n_cnt + j] : 0.0; }
	for (int k0 = 0; k0 < min_l; k0 += BLK) {
		const int nb = min(BLK, min_l - k0);
		const int mine = min(k0 + min(lane, BLK - 1), min_l - 1);
		const double sv = s[off + mine];
		const int cv = cnt1[(int64_t)mine * n_cnt + j];
		double fu[BLK][PER], bu[BLK][PER];
#pragma unroll
		for (int t = 0; t < BLK; ++t) {
			const int64_t r = (off + min(k0 + t, min_l - 1)) * S + PER * lane;
#pragma unroll
			for (int q = 0; q < PER; ++q) { fu[t][q] = f[r + q]; bu[t][q] = b[r + q]; }
		}
#pragma unroll
		for (int t = 0; t < BLK; ++t) {
			if (t < nb) {
				const double ss = readlane_f64(sv, t);
				const double c = (double)__builtin_amdgcn_readlane(cv, t);
#pragma unroll
				for (int q = 0; q < PER; ++q) acc[q] += fu[t][q] * bu[t][q] * ss * c; // cnt += prob[l] * cnt1
			}
		}
	}
#pragma unroll
	for (int q = 0; q < PER; ++q) { const int k = PER * lane + q; if (k < n) cnt[(int64_t)k * n_cnt + j] = acc[q]; }
}

// ---------------------------------------------------------------- 65..128 states
// Same recursions with two ADJACENT states per lane (k = 2*lane and 2*lane + 1;
// `-p "64*2"` of the reference's README gives 128).  The 128x128 transition matrix
// (128 KB) no longer fits a lane's registers, so each block keeps it in LDS
// (M[l*128+k]) and streams it through the ordered dot product: one ds_read_b128
// per source state l feeds both of the lane's outputs, whose two add chains
// interleave.  e[b][l]*a[k][l] of the backward sweep is formed on the fly with one
// rounding, exactly as hmm_pre_backward does (khmm.c:194-206).
constexpr int S2 = 128;
typedef double d2_t __attribute__((ext_vector_type(2)));

// strict left-to-right sum over states 0..127; ev/od: replicated forms of the even / odd states
#define PSMC_FB2(N) "v_fmac_f64_dpp %0, %1, %3 row_newbcast:" #N " row_mask:0xf bank_mask:0xf\n\t"   \
                    "v_fmac_f64_dpp %0, %2, %3 row_newbcast:" #N " row_mask:0xf bank_mask:0xf\n\t"
__device__ __forceinline__ void add_bcast16x2(double &s, double ev, double od, double one) { // states 32B .. 32B+31 in order
	asm(PSMC_FB2(0) PSMC_FB2(1) PSMC_FB2(2) PSMC_FB2(3) PSMC_FB2(4) PSMC_FB2(5) PSMC_FB2(6) PSMC_FB2(7)
	    PSMC_FB2(8) PSMC_FB2(9) PSMC_FB2(10) PSMC_FB2(11) PSMC_FB2(12) PSMC_FB2(13) PSMC_FB2(14) PSMC_FB2(15)
	    : "+v"(s) : "v"(ev), "v"(od), "v"(one));
}
__device__ __forceinline__ double seq_sum_rep2(const double (&ev_)[4], const double (&od_)[4]) {
	double ev[4] = {ev_[0], ev_[1], ev_[2], ev_[3]}, od[4] = {od_[0], od_[1], od_[2], od_[3]};
	dpp_guard(ev); dpp_guard(od);
	double s = 0.0, one = 1.0; // one instruction per term: s + x as fma(x, 1.0, s) (add_bcast16, wave_prims.h)
	add_bcast16x2(s, ev[0], od[0], one); add_bcast16x2(s, ev[1], od[1], one);
	add_bcast16x2(s, ev[2], od[2], one); add_bcast16x2(s, ev[3], od[3], one);
	return s;
}

// two consecutive source states l = 2*(16 blk + N) and l + 1 (rows R, R+1 of the 16 staged rows)
#define PSMC_XT2(N, R)                                                                          \
	{ const double xb = bcast16<N>(r0); acc0 = acc0 + xb * mc[R].x; acc1 = acc1 + xb * mc[R].y; }     \
	{ const double xb = bcast16<N>(r1); acc0 = acc0 + xb * mc[(R) + 1].x; acc1 = acc1 + xb * mc[(R) + 1].y; }
// same with the emission factor: (e_l * M[l][k]) * x_l, the product e*a rounded first (khmm.c:203)
#define PSMC_XT2E(N, R)                                                                         \
	{ const double xb = bcast16<N>(r0), eb = bcast16<N>(e0);                                        \
	  acc0 = acc0 + (eb * mc[R].x) * xb; acc1 = acc1 + (eb * mc[R].y) * xb; }                       \
	{ const double xb = bcast16<N>(r1), eb = bcast16<N>(e1);                                        \
	  acc0 = acc0 + (eb * mc[(R) + 1].x) * xb; acc1 = acc1 + (eb * mc[(R) + 1].y) * xb; }
#define PSMC_XT2x8(T, N0)                                                                       \
	T(N0 + 0, 0) T(N0 + 1, 2) T(N0 + 2, 4) T(N0 + 3, 6) T(N0 + 4, 8) T(N0 + 5, 10) T(N0 + 6, 12) T(N0 + 7, 14)

// One group of 16 source states (rows 16*IT .. 16*IT+15 of M) of the ordered dot product below.
template <bool WITH_E, int IT>
__device__ __forceinline__ void xdot128_group(const d2_t *t, d2_t (&mc)[16], double &acc0, double &acc1,
                                              const double (&rx0)[4], const double (&rx1)[4], const double (&ee0)[4],
                                              const double (&ee1)[4])
{
	d2_t mn[16];
	if constexpr (IT < 7) {
#pragma unroll
		for (int l = 0; l < 16; ++l) mn[l] = t[(16 * (IT + 1) + l) * 64];
	}
	__builtin_amdgcn_sched_barrier(0);
	double r0 = rx0[IT >> 1], r1 = rx1[IT >> 1], e0 = WITH_E ? ee0[IT >> 1] : 0.0, e1 = WITH_E ? ee1[IT >> 1] : 0.0;
	asm volatile("" : "+v"(r0), "+v"(r1), "+v"(e0), "+v"(e1));
	if constexpr (WITH_E) {
		if constexpr ((IT & 1) == 0) { PSMC_XT2x8(PSMC_XT2E, 0) } else { PSMC_XT2x8(PSMC_XT2E, 8) }
	} else {
		if constexpr ((IT & 1) == 0) { PSMC_XT2x8(PSMC_XT2, 0) } else { PSMC_XT2x8(PSMC_XT2, 8) }
	}
	__builtin_amdgcn_sched_barrier(0);
	if constexpr (IT < 7) {
#pragma unroll
		for (int l = 0; l < 16; ++l) mc[l] = mn[l];
	}
}

// y[j] = sum_{l=0..127} (WITH_E ? e_l * M[l][k] : M[l][k]) * x_l,  k = 2*lane + j, strict l order.
// x[j] = value of state 2*lane + j; ee0/ee1[blk] = e of the even / odd states in replicated form.
// The 8 groups of 16 rows are software-pipelined by hand: the reads of the next group are
// issued before the 32 ordered terms of the current one; the scheduling barriers keep the
// compiler from hoisting all 128 reads, and the opaque copies from keeping (spilling) values
// across groups.
template <int REP, bool WITH_E>
__device__ __forceinline__ void xdot128_lds(const double *m_lds, int lane, const double (&x)[2], const double (&ee0)[4],
                                            const double (&ee1)[4], double (&y)[2])
{
	double rx0[4], rx1[4];
	rep_rows<REP>(x[0], rx0); rep_rows<REP>(x[1], rx1);
	asm volatile("" : "+v"(lane)); // opaque per call: the LDS reads are loop invariant and must not be hoisted out of the sweep
	const d2_t *t = reinterpret_cast<const d2_t *>(m_lds) + lane; // row stride 64 d2
	d2_t mc[16];
#pragma unroll
	for (int l = 0; l < 16; ++l) mc[l] = t[l * 64];
	double acc0 = 0.0, acc1 = 0.0;
	xdot128_group<WITH_E, 0>(t, mc, acc0, acc1, rx0, rx1, ee0, ee1);
	xdot128_group<WITH_E, 1>(t, mc, acc0, acc1, rx0, rx1, ee0, ee1);
	xdot128_group<WITH_E, 2>(t, mc, acc0, acc1, rx0, rx1, ee0, ee1);
	xdot128_group<WITH_E, 3>(t, mc, acc0, acc1, rx0, rx1, ee0, ee1);
	xdot128_group<WITH_E, 4>(t, mc, acc0, acc1, rx0, rx1, ee0, ee1);
	xdot128_group<WITH_E, 5>(t, mc, acc0, acc1, rx0, rx1, ee0, ee1);
	xdot128_group<WITH_E, 6>(t, mc, acc0, acc1, rx0, rx1, ee0, ee1);
	xdot128_group<WITH_E, 7>(t, mc, acc0, acc1, rx0, rx1, ee0, ee1);
	y[0] = acc0; y[1] = acc1;
}

// blockDim.x / 64 segments per block (1, 2 or 4 waves sharing the LDS copy of the matrix)
template <int REP>
__global__ __launch_bounds__(256) void k_fwd_exact128(const double *__restrict__ a, const double *__restrict__ e,
                                                        const double *__restrict__ a0, const uint8_t *__restrict__ obs,
                                                        const int64_t *__restrict__ seg_off,
                                                        const int32_t *__restrict__ seg_len, const ExWork wl,
                                                        double *__restrict__ f, double *__restrict__ s)
{
	extern __shared__ double lds_m[]; // a[l*128+k]: at[k][l] of khmm.c:162-166 read column-wise
	// the sweeps of a block share the LDS copy of the matrix: a batch keeps the items of one parameter set block-aligned
	{ const int64_t po = wl.par ? wl.par[blockIdx.x * (blockDim.x >> 6)] * wl.par_stride : 0; a += po; e += po; a0 += po; }
	for (int i = threadIdx.x; i < S2 * S2; i += blockDim.x) lds_m[i] = a[i];
	__syncthreads();
	const int lane = threadIdx.x & 63;
	const int w = blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	if (w >= wl.n) return;
	const int seg = wl.seg[w];
	if (seg < 0) return; // padding entry of a batch
	const int64_t off = seg_off[seg], toff = wl.tab ? wl.tab[w] : off;
	const int L = seg_len[seg];
	const double e0[2] = {e[2 * lane], e[2 * lane + 1]}, e1[2] = {e[S2 + 2 * lane], e[S2 + 2 * lane + 1]};
	const uint8_t *o = obs + off;
	double *fo = f + toff * S2, *so = s + toff;
	int symv = o[lane]; // obs is padded by >= 64 bytes at the end
	double x[2];
	const double no_e[4] = {0, 0, 0, 0};
	auto finish = [&](const double (&g)[2], int idx) { // normalise by the ordered sum, store (khmm.c:180-184)
		double q0[4], q1[4];
		rep_rows<REP>(g[0], q0); rep_rows<REP>(g[1], q1);
		const double sum = seq_sum_rep2(q0, q1);
		x[0] = g[0] / sum; x[1] = g[1] / sum;
		d2_t v; v.x = x[0]; v.y = x[1];
		reinterpret_cast<d2_t *>(fo + (int64_t)idx * S2)[lane] = v;
		if (lane == 0) so[idx] = sum;
	};
	{ // position 1 (khmm.c:171-174)
		const int sym = __builtin_amdgcn_readlane(symv, 0);
		const double g[2] = {a0[2 * lane] * pick_e(sym, e0[0], e1[0]), a0[2 * lane + 1] * pick_e(sym, e0[1], e1[1])};
		finish(g, 0);
	}
	for (int base = 0; base < L; base += 64) {
		const int nb = min(64, L - base);
		const int symn = (base + 64 < L) ? (int)o[base + 64 + lane] : 2;
		for (int i = (base == 0 ? 1 : 0); i < nb; ++i) { // khmm.c:176-185
			const int sym = __builtin_amdgcn_readlane(symv, i);
			double tmp[2];
			xdot128_lds<REP, false>(lds_m, lane, x, no_e, no_e, tmp);
			const double g[2] = {pick_e(sym, e0[0], e1[0]) * tmp[0], pick_e(sym, e0[1], e1[1]) * tmp[1]};
			finish(g, base + i);
		}
		symv = symn;
	}
}

template <int REP>
__global__ __launch_bounds__(256) void k_bwd_exact128(const double *__restrict__ aT, const double *__restrict__ e,
                                                        const double *__restrict__ a0, const uint8_t *__restrict__ obs,
                                                        const int64_t *__restrict__ seg_off,
                                                        const int32_t *__restrict__ seg_len, const ExWork wl,
                                                        const double *__restrict__ s, double *__restrict__ b,
                                                        double *__restrict__ chk)
{
	extern __shared__ double lds_m[]; // aT[l*128+k] = a[k][l], then e[0][*], e[1][*], e[2][*]
	{ const int64_t po = wl.par ? wl.par[blockIdx.x * (blockDim.x >> 6)] * wl.par_stride : 0; aT += po; e += po; a0 += po; }
	for (int i = threadIdx.x; i < S2 * S2; i += blockDim.x) lds_m[i] = aT[i];
	for (int i = threadIdx.x; i < 3 * S2; i += blockDim.x) lds_m[S2 * S2 + i] = e[i];
	__syncthreads();
	const int lane = threadIdx.x & 63;
	const int w = blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	if (w >= wl.n) return;
	const int seg = wl.seg[w];
	if (seg < 0) return; // padding entry of a batch
	const int64_t off = seg_off[seg], toff = wl.tab ? wl.tab[w] : off;
	const int L = seg_len[seg];
	const uint8_t *o = obs + off;
	const double *so = s + toff;
	double *bo = b + toff * S2;
	double x[2];
	x[0] = x[1] = 1.0 / so[L - 1]; // b[L][k] = 1/s[L] (khmm.c:226)
	{ d2_t v; v.x = x[0]; v.y = x[1]; reinterpret_cast<d2_t *>(bo + (int64_t)(L - 1) * S2)[lane] = v; }
	for (int top = L - 2; top >= 0; top -= 64) {
		const int nb = min(64, top + 1);
		const int lo = top - 63;
		const int idx = max(lo + lane, 0);
		const double sv = so[idx];
		const int symv = o[idx + 1]; // symbol of position u+1 (index u)
		for (int i = 63; i >= 64 - nb; --i) {
			const int sym = __builtin_amdgcn_readlane(symv, i);
			const double su = readlane_f64(sv, i);
			// emission row of this symbol in replicated form (even / odd states); row 2 is all 1.0
			// (khmm.c:21) and 1.0*a[k][l] is a[k][l] exactly, so one code path serves all symbols
			double y[2], ee0[4], ee1[4];
			const d2_t *ep = reinterpret_cast<const d2_t *>(lds_m + S2 * S2 + sym * S2) + (lane & 15);
#pragma unroll
			for (int blk = 0; blk < 4; ++blk) { const d2_t v = ep[16 * blk]; ee0[blk] = v.x; ee1[blk] = v.y; }
			xdot128_lds<REP, true>(lds_m, lane, x, ee0, ee1, y);
			x[0] = y[0] / su; x[1] = y[1] / su;
			d2_t v; v.x = x[0]; v.y = x[1];
			reinterpret_cast<d2_t *>(bo + (int64_t)(lo + i) * S2)[lane] = v;
		}
	}
	{ // underflow check value (khmm.c:237-238)
		const int sym = o[0];
		const double t0 = a0[2 * lane] * x[0] * e[sym * S2 + 2 * lane], t1 = a0[2 * lane + 1] * x[1] * e[sym * S2 + 2 * lane + 1];
		double q0[4], q1[4];
		rep_rows<REP>(t0, q0); rep_rows<REP>(t1, q1);
		const double c = seq_sum_rep2(q0, q1);
		if (lane == 0) chk[w] = c;
	}
}

template <int REP> static int launch_exact128_t(const EstepLaunch &p)
{
	const size_t lds = sizeof(double) * (S2 * S2 + 3 * S2); // 131 KB of the CU's 160 KB
	// per device and cheap: allow more than the default 64 KB of dynamic LDS
	if (hipFuncSetAttribute((const void *)k_fwd_exact128<REP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
	    hipFuncSetAttribute((const void *)k_bwd_exact128<REP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
		return (int)hipGetLastError();
	// the sweeps are LDS-bandwidth bound: spread the segments over all 256 CUs before stacking waves on one
	// (a batch list is padded per parameter set to p.work_align entries: that many waves per block at most)
	const int wpb = p.work_align > 0 ? p.work_align : (p.n_work <= 256 ? 1 : (p.n_work <= 512 ? 2 : 4));
	const int nb = (p.n_work + wpb - 1) / wpb;
	const ExWork wl = {p.d_work, p.d_work_par, p.d_work_tab, nullptr, p.n_work, p.par_stride};
	if (p.ev[0]) hipEventRecord(p.ev[0], p.stream);
	hipLaunchKernelGGL(k_fwd_exact128<REP>, dim3(nb), dim3(64 * wpb), lds, p.stream, p.d_a, p.d_e, p.d_a0, p.d_obs,
	                   p.d_seg_off, p.d_seg_len, wl, p.d_f, p.d_s);
	if (p.ev[1]) hipEventRecord(p.ev[1], p.stream);
	hipLaunchKernelGGL(k_bwd_exact128<REP>, dim3(nb), dim3(64 * wpb), lds, p.stream, p.d_aeT, p.d_e, p.d_a0, p.d_obs,
	                   p.d_seg_off, p.d_seg_len, wl, p.d_s, p.d_b, p.d_chk);
	if (p.ev[2]) hipEventRecord(p.ev[2], p.stream);
	hipLaunchKernelGGL(k_expect_exact<128>, dim3(p.n_work, 66), dim3(64), 0, p.stream, p.d_a, p.d_aeT, p.d_e, p.d_a0,
	                   p.d_obs, p.d_seg_off, p.d_seg_len, wl, p.d_f, p.d_b, p.d_s, p.d_segA, p.d_segE, p.d_segA0);
	if (p.ev[3]) hipEventRecord(p.ev[3], p.stream);
	if (p.ev[4]) hipEventRecord(p.ev[4], p.stream);
	return (int)hipGetLastError();
}

// strict state-order sums for the decoding kernels above: ((0 + t_0) + t_1) + ... like `p += ...` in aux.c:191-192
template <> __device__ __forceinline__ double ordered_sum_states<64>(double t0, double)
{
	double r[4];
	rep_rows<1>(t0, r);
	return seq_sum_rep(r);
}
template <> __device__ __forceinline__ double ordered_sum_states<128>(double t0, double t1)
{
	double q0[4], q1[4];
	rep_rows<1>(t0, q0); rep_rows<1>(t1, q1);
	return seq_sum_rep2(q0, q1);
}

int launch_post_full(hipStream_t st, const double *a, const double *e, const uint8_t *obs, const double *f, const double *b,
                     const double *s, int64_t off, int L, int n, int ns, double *post, double *recomb)
{
	if (ns == 128)
		hipLaunchKernelGGL(k_post_full<128>, dim3((L + 63) / 64), dim3(64), 0, st, a, e, obs, f, b, s, off, L, n, post, recomb);
	else
		hipLaunchKernelGGL(k_post_full<64>, dim3((L + 63) / 64), dim3(64), 0, st, a, e, obs, f, b, s, off, L, n, post, recomb);
	return (int)hipGetLastError();
}

int launch_post_counts(hipStream_t st, const double *f, const double *b, const double *s, int64_t off, int min_l,
                       const int32_t *cnt1, int n_cnt, int n, int ns, double *cnt)
{
	if (min_l <= 0 || n_cnt <= 0) return 0;
	if (ns == 128)
		hipLaunchKernelGGL(k_post_counts<128>, dim3(n_cnt), dim3(64), 0, st, f, b, s, off, min_l, cnt1, n_cnt, n, cnt);
	else
		hipLaunchKernelGGL(k_post_counts<64>, dim3(n_cnt), dim3(64), 0, st, f, b, s, off, min_l, cnt1, n_cnt, n, cnt);
	return (int)hipGetLastError();
}

// ---------------------------------------------------------------- hmm_lk without the table read-back
// hmm_lk (khmm.c:245-260) is a running product of the scale factors, `prod *= s[u]`, that is logged and reset whenever it leaves
// [1e-25, 1e25), plus one last log.  The logarithm must be the host's (the reference calls the platform libm), but the products are
// plain IEEE multiplications in position order: the device forms them -- the same multiplications in the same order, hence the same
// bits -- and hands over only the numbers that are logged, a few dozen per entry instead of 8 bytes per bin (the exact batch read 4 GB
// of scale factors per launch over PCIe for this: 0.08 s of every 1.2 s).  One wave per entry: 64 scale factors per load, the chain
// itself wave-uniform.  Entry i writes to out + lk_off[i], cap = L / LKP_DIV + LKP_MIN doubles (a reset every LKP_DIV bins on average would
// need scale factors around 0.03: ten times below a run of heterozygous bins): [0] = the count n as a double, [1 .. n] the products
// in order (the last one is the final `sum += log(prod)`); n > cap - 1: overflow, the host reads that entry's scale factors instead.
__global__ __launch_bounds__(64) void k_lk_products(const int32_t *__restrict__ seg_len, const ExWork wl, const int64_t *__restrict__ seg_off,
                                                    const double *__restrict__ s, const int64_t *__restrict__ lk_off, double *__restrict__ out)
{
	const int lane = threadIdx.x;
	const int seg = wl.seg[blockIdx.x];
	double *o = out + lk_off[blockIdx.x];
	if (seg < 0) { if (lane == 0) o[0] = 0.0; return; }
	const int L = seg_len[seg];
	const int cap = L / LKP_DIV + LKP_MIN;
	const double *so = s + (wl.tab_s ? wl.tab_s[blockIdx.x] : (wl.tab ? wl.tab[blockIdx.x] : seg_off[seg]));
	double prod = 1.0;
	int n = 0;
	for (int base = 0; base < L; base += 64) {
		const int nb = min(64, L - base);
		const double sv = so[base + min(lane, nb - 1)];
		for (int i = 0; i < nb; ++i) {
			prod = prod * readlane_f64(sv, i);
			if (prod < PSMC_TINY || prod >= 1.0 / PSMC_TINY) { // khmm.c:252-255 (wave-uniform)
				++n;
				if (lane == 0 && n < cap) o[n] = prod;
				prod = 1.0;
			}
		}
	}
	++n; // khmm.c:257: the last log, whatever the product
	if (lane == 0) { if (n < cap) o[n] = prod; o[0] = (double)n; }
}

int launch_lk_products(hipStream_t st, const EstepLaunch &p, const double *d_s, const int64_t *d_lk_off, double *d_out)
{
	if (p.n_work <= 0) return 0;
	const ExWork wl = {p.d_work, p.d_work_par, p.d_work_tab, p.d_work_tab_s, p.n_work, p.par_stride};
	hipLaunchKernelGGL(k_lk_products, dim3(p.n_work), dim3(64), 0, st, p.d_seg_len, wl, p.d_seg_off, d_s, d_lk_off, d_out);
	return (int)hipGetLastError();
}

// ---------------------------------------------------------------- launchers
int launch_exact(const EstepLaunch &p)
{
	// auto: a wave alone on its SIMD pays for every instruction and waits out the LDS round trip of ds_bpermute -- the register
	// swaps win; from two waves per SIMD on the vector pipe is the bound and the eight ds_bpermute cost it nothing
	const int rep = p.rep_impl < 0 ? (p.n_work > 1024 ? 0 : 1) : p.rep_impl;
	if (p.n_work <= 0) return 0;
	(void)hipGetLastError(); // the value returned below must be about THESE launches (polled events, elapsed-time queries leave errors behind)
	if (p.ns > 128) return launch_exact_wide(p);
	if (p.ns == 128) return rep == 0 ? launch_exact128_t<0>(p) : launch_exact128_t<1>(p);
	const ExWork wl = {p.d_work, p.d_work_par, p.d_work_tab, p.d_work_tab_s, p.n_work, p.par_stride};
	if (p.ev[0]) hipEventRecord(p.ev[0], p.stream);
	if (p.exact_only == 2) { /* the scale factors exist: the batch ran the forward pass of all its replicates at once */ }
	else if (p.exact_refwd) { // scale factors only
		if (rep == 0)
			hipLaunchKernelGGL((k_fwd_exact<0, false>), dim3(p.n_work), dim3(64), 0, p.stream, p.d_a, p.d_e, p.d_a0, p.d_obs,
			                   p.d_seg_off, p.d_seg_len, wl, p.d_f, p.d_s);
		else
			hipLaunchKernelGGL((k_fwd_exact<1, false>), dim3(p.n_work), dim3(64), 0, p.stream, p.d_a, p.d_e, p.d_a0, p.d_obs,
			                   p.d_seg_off, p.d_seg_len, wl, p.d_f, p.d_s);
	} else if (rep == 0)
		hipLaunchKernelGGL(k_fwd_exact<0>, dim3(p.n_work), dim3(64), 0, p.stream, p.d_a, p.d_e, p.d_a0, p.d_obs,
		                   p.d_seg_off, p.d_seg_len, wl, p.d_f, p.d_s);
	else
		hipLaunchKernelGGL(k_fwd_exact<1>, dim3(p.n_work), dim3(64), 0, p.stream, p.d_a, p.d_e, p.d_a0, p.d_obs,
		                   p.d_seg_off, p.d_seg_len, wl, p.d_f, p.d_s);
	if (p.ev[1]) hipEventRecord(p.ev[1], p.stream);
	if (p.exact_only == 1) { if (p.ev[4]) hipEventRecord(p.ev[4], p.stream); return (int)hipGetLastError(); }
	const int nb = (p.n_work + 3) / 4;
	if (rep == 0)
		hipLaunchKernelGGL(k_bwd_exact<0>, dim3(nb), dim3(256), 0, p.stream, p.d_aeT, p.d_e, p.d_a0, p.d_obs,
		                   p.d_seg_off, p.d_seg_len, wl, p.d_s, p.d_b, p.d_chk);
	else
		hipLaunchKernelGGL(k_bwd_exact<1>, dim3(nb), dim3(256), 0, p.stream, p.d_aeT, p.d_e, p.d_a0, p.d_obs,
		                   p.d_seg_off, p.d_seg_len, wl, p.d_s, p.d_b, p.d_chk);
	if (p.ev[2]) hipEventRecord(p.ev[2], p.stream);
	if (p.exact_refwd == 2 && p.work_align >= 2 && p.n_work % 2 == 0) { // two entries per work-group (k_expect_exact_rf2)
		if (p.d_cu_mask) (void)hipMemsetAsync(p.d_cu_mask, 0, 4096 * sizeof(int), p.stream);
		if (rep == 0)
			hipLaunchKernelGGL(k_expect_exact_rf2<0>, dim3(p.n_work / 2), dim3(256), 0, p.stream, p.d_a, p.d_e, p.d_a0, p.d_obs, p.d_seg_off, p.d_seg_len, wl,
			                   p.d_b, p.d_s, p.d_segA, p.d_segE, p.d_segA0, p.d_cu_mask);
		else
			hipLaunchKernelGGL(k_expect_exact_rf2<1>, dim3(p.n_work / 2), dim3(256), 0, p.stream, p.d_a, p.d_e, p.d_a0, p.d_obs, p.d_seg_off, p.d_seg_len, wl,
			                   p.d_b, p.d_s, p.d_segA, p.d_segE, p.d_segA0, p.d_cu_mask);
	} else if (p.exact_refwd) { // no f table: the third pass recomputes the forward sweep (k_expect_exact_rf)
		if (rep == 0)
			hipLaunchKernelGGL(k_expect_exact_rf<0>, dim3(p.n_work), dim3(192), 0, p.stream, p.d_a, p.d_e, p.d_a0, p.d_obs, p.d_seg_off, p.d_seg_len, wl,
			                   p.d_b, p.d_s, p.d_segA, p.d_segE, p.d_segA0);
		else
			hipLaunchKernelGGL(k_expect_exact_rf<1>, dim3(p.n_work), dim3(192), 0, p.stream, p.d_a, p.d_e, p.d_a0, p.d_obs, p.d_seg_off, p.d_seg_len, wl,
			                   p.d_b, p.d_s, p.d_segA, p.d_segE, p.d_segA0);
	} else
	hipLaunchKernelGGL(k_expect_exact<64>, dim3(p.n_work, 17), dim3(64), 0, p.stream, p.d_a, p.d_aeT, p.d_e, p.d_a0,
	                   p.d_obs, p.d_seg_off, p.d_seg_len, wl, p.d_f, p.d_b, p.d_s, p.d_segA, p.d_segE, p.d_segA0);
	if (p.ev[3]) hipEventRecord(p.ev[3], p.stream);
	if (p.ev[4]) hipEventRecord(p.ev[4], p.stream);
	return (int)hipGetLastError();
}

} // namespace psmc
