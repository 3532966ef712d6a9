// estep_fast.hip -- FAST mode of the PSMC E-step for gfx950: the protocol (speculate / verify / repair,
// streams, reductions: launch_fast), the kernels every variant shares (k_verify, k_expect_mfma, k_ll,
// k_reduce*) and the DENSE sweeps for arbitrary transition matrices.  Matrices of the PSMC form take the
// O(N) sweeps of estep_struct.hip instead (default whenever they apply); estep_fused.hip and
// estep_factored.hip are alternative back halves (counts fused into the backward sweep / no N x N counts).
//
// Same mathematics as khmm.c's hmm_forward / hmm_backward / hmm_expect
// (lh3/psmc khmm.c:145-190, 210-241, 297-324) but re-associated for the GPU:
//   * every segment is cut into tiles of `chunk` bins, one wavefront per tile,
//     lane = hidden state.  A tile SPECULATES: its sweep starts `warmup` bins
//     outside the tile from an arbitrary vector.  A VERIFY kernel compares the
//     vector each tile used at its boundary with the value its neighbour
//     computed with a whole tile of history behind it; only tiles whose
//     mismatch exceeds `warm_tol` are REPAIRED: re-run from the neighbour's
//     value until the new trajectory meets the stored one again.  (The chain
//     forgets its start in ~2-4 k bins almost everywhere but needs >60 k bins
//     in long runs of recent-coalescence states, so a fixed overlap is either
//     wrong or 5x redundant.)
//   * the 64-term dot products are 64 v_fmac_f64_dpp (row_newbcast operand
//     broadcast straight from the natural-layout state register; 64 matrix entries
//     in 128 VGPRs per lane; a 2-level permlane-swap transpose-reduce brings the
//     result back to natural layout -- wave_prims.h matvec64_nat);
//   * lagged, sparse normalisation: X_p = e[o_p]*(a^T X_{p-1}) / d_p with
//     d_p = sum(X_{p-1}) when p % 4 == 0 and 1 otherwise: the cross-lane reduction
//     is off the sequential critical path and paid every 4th bin only;
//     LL = sum_p log d_p + log sum(X_L) telescopes for ANY positive d_p;
//   * the backward sweep is the mirror image and needs NOTHING from the forward one:
//     bt_p = e[o_p]*(a bt_{p+1})*sb_p with its own lagged scale sb_p = 1/sum(bt_{p+1})
//     at p % 4 == 0, so the two sweeps (and their repair rounds) run concurrently
//     on two streams;
//   * counts from the stored tables, normalised per position in the expect kernel:
//     G_p = sum_k X_p bt_p / e[o_p] (= sb_p sum_kl X_p[k] a[k][l] bt_{p+1}[l]),
//     E[o_p] += X_p bt_p / e[o_p] / G_p, and C = sum_p (sb_p/G_p) X_p (x) bt_{p+1} is a
//     K=bins GEMM on the FP64 matrix cores (v_mfma_f64_16x16x4_f64), A = a .* C.
//     Per-wave partials are reduced in a fixed order (deterministic, no atomics).
// tests/fastmodel.py is the executable numpy specification of this file.
// HBM layout (g = seg_off + p - 1): X[g*64+k] (d_f), bt[g*64+k] (d_b),
// inv_d[g] = 1/d_p (d_s) and sb[g] (d_sb), both written at p % 4 == 0 only, obs[g].
#include <hip/hip_runtime.h>
#include <cmath>
#include <algorithm>
#include "wave_prims.h"
#include "psmc_hip_internal.h"

namespace psmc {

typedef double d4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double pick_ef(int sym, double e0, double e1) {
	return sym == 0 ? e0 : (sym == 1 ? e1 : 1.0);
}

// 1/x to ~1 ulp: v_rcp_f64 seed + two Newton steps (x is a sum of probabilities)
__device__ __forceinline__ double fast_rcp(double x) {
	double r = __builtin_amdgcn_rcp(x);
	double t = __builtin_fma(-x, r, 1.0);
	r = __builtin_fma(r, t, r);
	t = __builtin_fma(-x, r, 1.0);
	r = __builtin_fma(r, t, r);
	return r;
}

__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
	for (int m = 32; m >= 1; m >>= 1) v = fmax(v, __shfl_xor(v, m, 64));
	return v;
}
__device__ __forceinline__ double wave_add(double v) {
#pragma unroll
	for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
	return v;
}
// max_k |x-y| / max_k |y|  (NaN-safe: a NaN anywhere yields +inf)
__device__ __forceinline__ double rel_mismatch(double x, double y) {
	const double num = wave_max(fabs(x - y)), den = wave_max(fabs(y));
	const bool bad = __any((x != x) || (y != y));
	return bad ? __builtin_inf() : num / den;
}

// ------------------------------------------------------------------ forward
// REPAIR=false: speculative pass over every tile.  REPAIR=true: only tiles the
// verify kernel flagged; starts from the neighbour's stored X_{lo-1} and stops
// as soon as the new trajectory meets the stored one (checked every 16 bins).
template <int REP, bool REPAIR>
__global__ __launch_bounds__(64) void k_fwd_fast(const double *__restrict__ a, const double *__restrict__ e,
                                                   const double *__restrict__ a0, const uint8_t *__restrict__ obs,
                                                   const Chunk *__restrict__ chunks, int W, double tol,
                                                   const int *__restrict__ dirty, double *__restrict__ f,
                                                   double *__restrict__ invd, double *__restrict__ entry,
                                                   int *__restrict__ touch_f)
{
	if (REPAIR && !dirty[blockIdx.x]) return;
	if (REPAIR) __builtin_amdgcn_s_setprio(3); // few, latency-critical waves
	const int lane = threadIdx.x;
	if (REPAIR && lane == 0) touch_f[blockIdx.x] = 1; // X / inv_d of this tile change
	const Chunk c = chunks[blockIdx.x];
	const uint8_t *o = obs + c.off;
	double *fo = f + c.off * 64, *io = invd + c.off;
	double A[64]; // A[16j+N] = a[16r+N][16j+m]
	load_nat_matrix(a, lane, A);
	const double e0 = e[lane], e1 = e[64 + lane];
	double x;
	int p;
	if (REPAIR) { // c.lo >= 2 for every flagged tile
		x = fo[(int64_t)(c.lo - 2) * 64 + lane];
		p = c.lo;
	} else {
		const int ws = max(1, c.lo - W);
		if (ws == 1) { // true start: X_1 = a0*e[o_1], d_1 = 1 (khmm.c:171-174 without the division)
			x = a0[lane] * pick_ef((int)o[0], e0, e1);
			if (c.lo == 1) fo[lane] = x;
			p = 2;
		} else { // warm-up from the stationary prior
			x = a0[lane];
			p = ws;
		}
	}
	// The sweep runs in blocks of 64 bins (one coalesced symbol load each, prefetched a block
	// ahead).  Inside a block there is NO load, so the compiler places no s_waitcnt vmcnt in
	// the step and the per-step global stores stay fire-and-forget.
	int symn = o[(((p - 1) >> 6) << 6) + lane];
	bool done = false;
	while (p <= c.hi && !done) {
		const int blk = (p - 1) >> 6;
		const int symv = symn;
		symn = o[((blk + 1) << 6) + lane];
		const int pend = min(c.hi, (blk + 1) << 6); // last position of this block
		double oldv = 0.0;
		if (REPAIR) oldv = fo[(int64_t)(pend - 1) * 64 + lane]; // what is stored there now
		for (; p <= pend; ++p) {
			const int idx = p - 1;
			const int sym = __builtin_amdgcn_readlane(symv, idx & 63);
			if (p == c.lo) entry[(int64_t)blockIdx.x * 64 + lane] = x; // the X_{lo-1} this tile builds on
			double ev = pick_ef(sym, e0, e1);
			if ((p & (NORM_EVERY - 1)) == 0) { // d_p = sum(X_{p-1}), off the critical path
				const double inv = fast_rcp(first_lane_f64(wave_sum_nat(x))); // one value for all lanes and for the table
				ev *= inv;
				if (p >= c.lo && lane == 0) io[idx] = inv;
			}
			x = matvec64_nat(x, A) * ev;
			if (p >= c.lo) fo[(int64_t)idx * 64 + lane] = x;
		}
		// met the stored trajectory: everything after this block is already right
		if (REPAIR && rel_mismatch(x, oldv) <= tol) done = true;
	}
}

// ------------------------------------------------------------------ backward
template <int REP, bool REPAIR>
__global__ __launch_bounds__(64) void k_bwd_fast(const double *__restrict__ aT, const double *__restrict__ e,
                                                   const uint8_t *__restrict__ obs, const Chunk *__restrict__ chunks,
                                                   int W, double tol, const int *__restrict__ dirty,
                                                   double *__restrict__ bt, double *__restrict__ sb,
                                                   double *__restrict__ bentry, double *__restrict__ bexit,
                                                   int *__restrict__ touch_b)
{
	if (REPAIR && !dirty[blockIdx.x]) return;
	if (REPAIR) __builtin_amdgcn_s_setprio(3);
	const int lane = threadIdx.x;
	if (REPAIR && lane == 0) touch_b[blockIdx.x] = 1;
	const Chunk c = chunks[blockIdx.x];
	const int L = c.L, lo = c.lo, top = min(c.hi, L - 1);
	if (top < lo) return; // a tile holding only position L owns no transition
	const uint8_t *o = obs + c.off;
	double *bto = bt + c.off * 64, *sbo = sb + c.off;
	double A[64]; // A[16j+N] = a[16j+m][16r+N] = aT[16r+N][16j+m]
	load_nat_matrix(aT, lane, A);
	const double e0 = e[lane], e1 = e[64 + lane];
	double btn; // bt_{p+1} = e[o_{p+1}] * B_{p+1} (own scaling), natural layout
	int p;
	if (REPAIR) { // continue from the value the tile above computed at our top boundary
		btn = bexit[(int64_t)(blockIdx.x + 1) * 64 + lane];
		p = top;
	} else {
		const int q = min(c.hi + W + 1, L); // B_q := 1
		btn = pick_ef((int)o[q - 1], e0, e1);
		p = q - 1;
	}
	// blocks of 64 bins going down, one coalesced symbol load per block prefetched a block ahead,
	// no load inside a block (see k_fwd_fast)
	int symn = o[(max((p - 1) >> 6, 0) << 6) + lane];
	bool done = false;
	while (p >= lo && !done) {
		const int blk = (p - 1) >> 6;
		const int symv = symn;
		symn = o[(max(blk - 1, 0) << 6) + lane];
		const int pbeg = max(lo, (blk << 6) + 1);  // lowest position of this block
		const int pc = max(pbeg, lo + 1);          // lowest position of the block this tile stores bt for
		double oldv = 0.0, chkv = 0.0;
		const bool can_check = REPAIR && pc <= min(p, top);
		if (can_check) oldv = bto[(int64_t)(pc - 1) * 64 + lane];
		for (; p >= pbeg; --p) {
			const int idx = p - 1;
			const int sym = __builtin_amdgcn_readlane(symv, idx & 63);
			double ev = pick_ef(sym, e0, e1);
			if ((p & (NORM_EVERY - 1)) == 0) { // sb_p = 1/sum(bt_{p+1}), off the critical path
				const double sc = fast_rcp(first_lane_f64(wave_sum_nat(btn)));
				ev *= sc;
				if (p <= top && lane == 0) sbo[idx] = sc;
			}
			if (p == top) { // the boundary vector this tile builds on
				bto[(int64_t)top * 64 + lane] = btn; // bt[top+1]
				bentry[(int64_t)blockIdx.x * 64 + lane] = btn;
			}
			btn = matvec64_nat(btn, A) * ev;
			if (p <= top) {
				if (p > lo || lo == 1) bto[(int64_t)idx * 64 + lane] = btn; // bt[p]; bt[lo>1] belongs to the tile below
				if (p == lo) bexit[(int64_t)blockIdx.x * 64 + lane] = btn;
				if (p == pc) chkv = btn;
			}
		}
		if (can_check && rel_mismatch(chkv, oldv) <= tol) done = true;
	}
}

// ------------------------------------------------------------------ verify
// flags tiles whose boundary vector disagrees with the neighbour's converged
// one; cnt[0] = number flagged, warm[which] = largest mismatch seen this round
// max_k |x/|x| - y/|y|| / max_k (y/|y|) over S = 64 or 128 states (lane and lane + 64), |.| = sum: boundary
// vectors are compared as DIRECTIONS (a vector from the transfer-matrix chain has an arbitrary scale; every
// consumer -- per-position normaliser of the counts, k_ll -- is scale-free)
template <int S> __device__ __forceinline__ double rel_mismatch_vec(const double *x, const double *y, int lane, bool with_scale = false) {
	double xv = x[lane], yv = y[lane], xw = 0.0, yw = 0.0;
	if (S == 128) { xw = x[64 + lane]; yw = y[64 + lane]; }
	const double sx = wave_add(xv + xw), sy = wave_add(yv + yw);
	const double ix = 1.0 / sx, iy = 1.0 / sy;
	double num = fabs(xv * ix - yv * iy), den = fabs(yv * iy);
	if (with_scale) num = fmax(num, fabs(sx - sy) * iy * den); // the two vectors themselves, not only their directions
	bool bad = (xv != xv) || (yv != yv) || (ix != ix) || (iy != iy);
	if (S == 128) {
		num = fmax(num, fabs(xw * ix - yw * iy)); den = fmax(den, fabs(yw * iy));
		bad = bad || (xw != xw) || (yw != yw);
	}
	num = wave_max(num); den = wave_max(den);
	return __any(bad) ? __builtin_inf() : num / den;
}
template <bool BWD, int S>
__global__ __launch_bounds__(64) void k_verify(const Chunk *__restrict__ chunks, int n_chunks, double tol,
                                                 const double *__restrict__ f, const double *__restrict__ mine,
                                                 const double *__restrict__ bexit, int *__restrict__ dirty,
                                                 int *__restrict__ cnt, unsigned long long *__restrict__ warm, int with_scale,
                                                 double *__restrict__ mis)
{
	// mis (first verify of an E-step; host-mapped memory): the mismatch itself, for the per-tile warm-ups
	// with_scale (backward, bt table in use): row lo-1 of the table is written by the tile below as ITS start vector
	// and read by this tile's counts next to its own bt[lo+1], so the two tiles must agree on the scale as well.  A
	// warm-up of at least one normalising position leaves the natural scale; without one (warmup < 4) a start vector
	// can have the right direction and the wrong length.
	const int lane = threadIdx.x, b = blockIdx.x;
	const Chunk c = chunks[b];
	double m = 0.0;
	bool check;
	if (!BWD) {
		check = c.lo > 1 && !(c.flags & CHUNK_ANCHOR_F);
		if (check) m = rel_mismatch_vec<S>(mine + (int64_t)b * S, f + (c.off + c.lo - 2) * S, lane);
	} else {
		check = !(c.flags & (CHUNK_ANCHOR_B | CHUNK_LAST)) && min(c.hi, c.L - 1) >= c.lo && b + 1 < n_chunks;
		if (check) m = rel_mismatch_vec<S>(mine + (int64_t)b * S, bexit + (int64_t)(b + 1) * S, lane, with_scale != 0);
	}
	if (lane == 0) {
		const int bad = check && !(m <= tol);
		dirty[b] = bad;
		if (bad) atomicAdd(cnt, 1);
		if (mis) mis[b] = check ? m : -1.0;
		if (check) atomicMax(&warm[BWD ? 1 : 0], (unsigned long long)__double_as_longlong(m));
	}
}

// ------------------------------------------------------------------ expect
// Inputs of a tile's counts that a repair may have rewritten after the early expect pass:
// X (forward repair of the tile), bt[lo+1..top+1] and sb (its backward repair) and bt[lo]
// (stored by the tile below as its top boundary value).
__device__ __forceinline__ bool tile_touched(const Chunk *__restrict__ chunks, int b, const int *__restrict__ touch_f,
                                             const int *__restrict__ touch_b)
{
	bool t = touch_f[b] || touch_b[b];
	if (b > 0 && chunks[b - 1].off == chunks[b].off) t = t || touch_b[b - 1];
	return t;
}
// Over the tile's positions lo..min(hi,L-1), split over n_sub waves:
//   g_p[k] = X_p[k] bt_p[k] / e[o_p][k],  G_p = sum_k g_p[k]      (posterior normaliser)
//   S[o_p][k] += g_p[k] / G_p                                      (emission counts)
//   C[k][l]  += (sb_p / G_p) X_p[k] * bt_{p+1}[l]                   (transition counts / a[k][l])
// C runs on the FP64 matrix cores: D(16x16) += A(16x4) B(4x16) with
//   A[i][t] = (sb/G)_{p+t} X_{p+t}[16m+i]  (lane = 16t+i),  B[t][j] = bt_{p+t+1}[16n+j] (lane = 16t+j)
//   D[(lane>>4)+4r][lane&15] = acc[r]
// so row group t of the wave holds position p+t and G is a 16-lane (one DPP row) reduction.
// S = 128 (the fast path of -p "64*2"): C is cut into (S/64)^2 quadrants of 64x64, one wave each; every
// wave still reduces G over all S states.  blockIdx.x = partial * (S/64)^2 + quadrant.
template <int S>
__global__ __launch_bounds__(64, 2) void k_expect_mfma(const Chunk *__restrict__ chunks, int n_sub,
                                                         const uint8_t *__restrict__ obs, const double *__restrict__ f,
                                                         const double *__restrict__ bt, const double *__restrict__ sb,
                                                         const double *__restrict__ re, double *__restrict__ Cpart,
                                                         double *__restrict__ Spart, const int *__restrict__ touch_f,
                                                         const int *__restrict__ touch_b, int redo)
{
	constexpr int Q = S / 64, NB = S / 16;
	const int lane = threadIdx.x, t = lane >> 4, i = lane & 15;
	const int part = blockIdx.x / (Q * Q), quad = blockIdx.x % (Q * Q), qm = quad / Q, qn = quad % Q;
	const Chunk c = chunks[part / n_sub];
	if (redo && !tile_touched(chunks, part / n_sub, touch_f, touch_b)) return;
	const int sub = part % n_sub;
	const int top = min(c.hi, c.L - 1), n = top - c.lo + 1;
	const int per = n > 0 ? ((((n + n_sub - 1) / n_sub) + 3) & ~3) : 0;
	const int p0 = c.lo + sub * per, p1 = min(top, p0 + per - 1);
	const double *fo = f + c.off * S + i, *bo = bt + c.off * S + i, *sbo = sb + c.off;
	const uint8_t *o = obs + c.off;
	double re0[NB], re1[NB]; // 1/e[b][16m+i]
#pragma unroll
	for (int m = 0; m < NB; ++m) { re0[m] = re[16 * m + i]; re1[m] = re[S + 16 * m + i]; }
	d4_t acc[4][4];
	double Sacc[3][4];
#pragma unroll
	for (int m = 0; m < 4; ++m) {
#pragma unroll
		for (int nn = 0; nn < 4; ++nn) acc[m][nn] = (d4_t){0.0, 0.0, 0.0, 0.0};
		Sacc[0][m] = Sacc[1][m] = Sacc[2][m] = 0.0;
	}
	auto load = [&](int p, double (&FA)[NB], double (&BM)[4], double (&BP)[NB], double &sc, int &sym, bool &ok) {
		const int pp = p + t;
		ok = pp <= p1;
		const int64_t idx = (int64_t)(ok ? pp : p1) - 1;
		const double *fr = fo + idx * S, *br = bo + idx * S;
#pragma unroll
		for (int m = 0; m < NB; ++m) { FA[m] = fr[16 * m]; BP[m] = br[16 * m]; }
#pragma unroll
		for (int m = 0; m < 4; ++m) BM[m] = br[S + 64 * qn + 16 * m];
		sc = 1.0; // sb_p at the normalising positions, 1 elsewhere
		if (((int)(idx + 1) & (NORM_EVERY - 1)) == 0) sc = sbo[idx];
		sym = o[idx];
	};
	if (p0 <= p1) {
		double FA[NB], BM[4], BP[NB], sc; int sym; bool ok;
		load(p0, FA, BM, BP, sc, sym, ok);
		for (int p = p0; p <= p1; p += 4) {
			double FN[NB], BN[4] = {0, 0, 0, 0}, BQ[NB], scn = 1.0; int symn = 2; bool okn = false;
#pragma unroll
			for (int m = 0; m < NB; ++m) { FN[m] = 0.0; BQ[m] = 0.0; }
			// S = 64: the next group's operands are fetched a group ahead; S = 128 has no registers to spare for that and
			// relies on the second wave of the SIMD instead (the loads are issued after the MFMAs)
			if (S == 64 && p + 4 <= p1) load(p + 4, FN, BN, BQ, scn, symn, okn);
			// per-position normaliser: row group t reduces its own position over the S states
			double g[NB], G = 0.0;
#pragma unroll
			for (int m = 0; m < NB; ++m) {
				const double r = sym == 0 ? re0[m] : (sym == 1 ? re1[m] : 1.0);
				g[m] = FA[m] * BP[m] * r;
				G += g[m];
			}
			G = G + dpp_mov<0xB1>(G);  // quad_perm:[1,0,3,2]
			G = G + dpp_mov<0x4E>(G);  // quad_perm:[2,3,0,1]
			G = G + dpp_mov<0x124>(G); // row_ror:4
			G = G + dpp_mov<0x128>(G); // row_ror:8
			const double iG = ok ? 1.0 / G : 0.0; // padded rows of the last group contribute nothing
			const double w0 = sym == 0 ? iG : 0.0, w1 = sym == 1 ? iG : 0.0, w2 = sym == 2 ? iG : 0.0, wa = sc * iG;
			double FM[4];
#pragma unroll
			for (int m = 0; m < 4; ++m) {
				const double gm = g[4 * qm + m];
				Sacc[0][m] = __builtin_fma(gm, w0, Sacc[0][m]);
				Sacc[1][m] = __builtin_fma(gm, w1, Sacc[1][m]);
				Sacc[2][m] = __builtin_fma(gm, w2, Sacc[2][m]);
				FM[m] = FA[4 * qm + m] * wa;
			}
#pragma unroll
			for (int m = 0; m < 4; ++m)
#pragma unroll
				for (int nn = 0; nn < 4; ++nn)
					acc[m][nn] = __builtin_amdgcn_mfma_f64_16x16x4f64(FM[m], BM[nn], acc[m][nn], 0, 0, 0);
			if (S != 64 && p + 4 <= p1) load(p + 4, FA, BM, BP, sc, sym, ok);
			else if (S != 64) ok = false;
			if (S == 64) {
#pragma unroll
				for (int m = 0; m < NB; ++m) { FA[m] = FN[m]; BP[m] = BQ[m]; }
#pragma unroll
				for (int m = 0; m < 4; ++m) BM[m] = BN[m];
				sc = scn; sym = symn; ok = okn;
			}
		}
	}
	const double mult = (double)c.mult;
	double *out = Cpart + (int64_t)part * (S * S) + (int64_t)(64 * qm) * S + 64 * qn;
#pragma unroll
	for (int m = 0; m < 4; ++m)
#pragma unroll
		for (int nn = 0; nn < 4; ++nn)
#pragma unroll
			for (int r = 0; r < 4; ++r) out[(16 * m + t + 4 * r) * S + 16 * nn + i] = acc[m][nn][r] * mult;
	if (qn == 0) {
		double *os = Spart + (int64_t)part * (3 * S) + 64 * qm;
#pragma unroll
		for (int b = 0; b < 3; ++b)
#pragma unroll
			for (int m = 0; m < 4; ++m) {
				double v = Sacc[b][m];
				v += __shfl_xor(v, 16, 64);
				v += __shfl_xor(v, 32, 64);
				if (t == 0) os[b * S + 16 * m + i] = v * mult;
			}
	}
}

// ------------------------------------------------------------------ log-likelihood
// LL of a tile = sum over its normalising positions of log d_p = -log inv_d  (+ log sum(X_L)
// for the last tile): running products flushed through log() like hmm_lk (khmm.c:245-260).
template <int S>
__global__ __launch_bounds__(64) void k_ll(const Chunk *__restrict__ chunks, const double *__restrict__ f,
                                             const double *__restrict__ invd, const double *__restrict__ entry,
                                             double *__restrict__ LLpart, const int *__restrict__ fmerge, const double *__restrict__ finv)
{
	const int lane = threadIdx.x;
	const Chunk c = chunks[blockIdx.x];
	const double *io = invd + c.off;
	double prod = 1.0, ll = 0.0;
	const int first = (max(c.lo, 2) + NORM_EVERY - 1) & ~(NORM_EVERY - 1);
	for (int p = first + NORM_EVERY * lane; p <= c.hi; p += NORM_EVERY * 64) {
		prod *= io[p - 1];
		if (prod > 1e280 || prod < 1e-280) { ll -= log(prod); prod = 1.0; }
	}
	ll -= log(prod);
	ll = wave_add(ll);
	if (c.lo > 1) { // the tile was computed from entry = X_{lo-1} up to a factor: put the telescoping sum back in step
		double u = f[(c.off + c.lo - 2) * S + lane], v = entry[(int64_t)blockIdx.x * S + lane];
		if (S == 128) { u += f[(c.off + c.lo - 2) * S + 64 + lane]; v += entry[(int64_t)blockIdx.x * S + 64 + lane]; }
		ll += log(wave_add(u)) - log(wave_add(v));
	}
	if (c.hi == c.L) {
		double v = f[(c.off + c.L - 1) * S + lane];
		if (S == 128) v += f[(c.off + c.L - 1) * S + 64 + lane];
		ll += log(wave_add(v));
	}
	if (fmerge && fmerge[blockIdx.x] > 0) ll -= log(finv[blockIdx.x]); // a merging repair: the rows above its merge point are 1/finv times what the rows below continue to (estep_struct.hip FwdCtl)
	if (lane == 0) LLpart[blockIdx.x] = ll * (double)c.mult;
}

// ------------------------------------------------------------------ reduce
// Fixed-order two-stage reduction of the per-wave partials (deterministic).
template <int S>
__global__ __launch_bounds__(256) void k_reduce1(const double *__restrict__ Cpart, const double *__restrict__ Spart,
                                                   int nC, int nS, const double *__restrict__ LLpart, int nchunks,
                                                   double *__restrict__ stage)
{
	constexpr int SL = S * S + 3 * S + 1; // C | S-counts | LL of one stage row
	const int y = blockIdx.y, tid = threadIdx.x;
	double *st = stage + (int64_t)y * SL;
	if ((int)blockIdx.x < S * S / 256) {
		const int i = blockIdx.x * 256 + tid;
		double s = 0.0;
		for (int j = y; j < nC; j += RED_ROWS) s += Cpart[(int64_t)j * (S * S) + i];
		st[i] = s;
	} else {
		for (int i = tid; i < 3 * S; i += 256) {
			double s = 0.0;
			for (int j = y; j < nS; j += RED_ROWS) s += Spart[(int64_t)j * (3 * S) + i];
			st[S * S + i] = s;
		}
		if (tid == 255) {
			double s = 0.0;
			for (int j = y; j < nchunks; j += RED_ROWS) s += LLpart[j];
			st[S * S + 3 * S] = s;
		}
	}
}
// Writes the final statistics UNPADDED: out = [A n*n | E 2*n | LL].
template <int S>
__global__ __launch_bounds__(256) void k_reduce2(const double *__restrict__ stage, const double *__restrict__ a,
                                                   const double *__restrict__ e, double tiny_total, int n,
                                                   double *__restrict__ out)
{
	constexpr int SL = S * S + 3 * S + 1;
	const int i = blockIdx.x * 256 + threadIdx.x;
	if (i >= SL) return;
	double s = 0.0;
	for (int y = 0; y < RED_ROWS; ++y) s += stage[(int64_t)y * SL + i];
	if (i < S * S) { // A = a .* C + n_seg*HMM_TINY (khmm.c:305-306,316)
		const int k = i / S, l = i % S;
		if (k < n && l < n) out[k * n + l] = a[i] * s + tiny_total;
	} else if (i < S * S + 3 * S) { // E + n_seg*HMM_TINY; the missing-symbol row is dropped (khmm.c:355)
		const int b = (i - S * S) / S, k = (i - S * S) % S;
		if (b < 2 && k < n) out[n * n + b * n + k] = s + tiny_total;
	} else {
		out[n * n + 2 * n] = s;
	}
}

// ------------------------------------------------------------------ launcher
template <bool REPAIR>
static void launch_fwd(const EstepLaunch &p, hipStream_t st)
{
	const dim3 g(p.n_chunks), b(64);
	if (p.rep_impl == 0)
		hipLaunchKernelGGL((k_fwd_fast<0, REPAIR>), g, b, 0, st, p.d_a, p.d_e, p.d_a0, p.d_obs, p.d_chunks, p.warmup,
		                   p.tol, p.d_dirty, p.d_f, p.d_s, p.d_entry, p.d_touch_f);
	else
		hipLaunchKernelGGL((k_fwd_fast<1, REPAIR>), g, b, 0, st, p.d_a, p.d_e, p.d_a0, p.d_obs, p.d_chunks, p.warmup,
		                   p.tol, p.d_dirty, p.d_f, p.d_s, p.d_entry, p.d_touch_f);
}
template <bool REPAIR>
static void launch_bwd(const EstepLaunch &p, hipStream_t st)
{
	const dim3 g(p.n_chunks), b(64);
	const double *aT = p.d_aeT + 2 * 4096;
	if (p.rep_impl == 0)
		hipLaunchKernelGGL((k_bwd_fast<0, REPAIR>), g, b, 0, st, aT, p.d_e, p.d_obs, p.d_chunks, p.warmup, p.tol,
		                   p.d_dirty_b, p.d_b, p.d_sb, p.d_bentry, p.d_bexit, p.d_touch_b);
	else
		hipLaunchKernelGGL((k_bwd_fast<1, REPAIR>), g, b, 0, st, aT, p.d_e, p.d_obs, p.d_chunks, p.warmup, p.tol,
		                   p.d_dirty_b, p.d_b, p.d_sb, p.d_bentry, p.d_bexit, p.d_touch_b);
}
static void launch_expect(const EstepLaunch &p, hipStream_t st, int redo)
{
	const int nC = p.n_chunks * p.n_sub;
	{
		if (p.ns == 128)
			hipLaunchKernelGGL(k_expect_mfma<128>, dim3(nC * 4), dim3(64), 0, st, p.d_chunks, p.n_sub, p.d_obs, p.d_f, p.d_b, p.d_sb,
			                   p.d_re, p.d_Cpart, p.d_Epart, p.d_touch_f, p.d_touch_b, redo);
		else
			hipLaunchKernelGGL(k_expect_mfma<64>, dim3(nC), dim3(64), 0, st, p.d_chunks, p.n_sub, p.d_obs, p.d_f, p.d_b, p.d_sb,
			                   p.d_re, p.d_Cpart, p.d_Epart, p.d_touch_f, p.d_touch_b, redo);
	}
}

// One fast-mode E-step.  The forward and the backward sweep do not depend on each other, so
// with p.overlap they run on two streams from the start, each followed by its own
// verify / repair rounds (a few dozen latency-bound waves); as soon as both speculative
// sweeps are done an early pass of the counts runs over every tile on a third stream, and
// tiles a repair rewrote afterwards are simply recomputed (their per-wave partials are
// overwritten):
//   main: fwd speculate | fwd verify/repair ...          | k_ll | expect(redo) reduce
//   aux : bwd speculate | bwd verify/repair ...          |
//   exp :               | expect (all tiles)             |
// With the fused back half (p.fused == 1, the default for PSMC-form matrices up to 64 states) there is no
// backward table and no separate counts pass:
//   main : fwd sweep (all tiles, or list A then B)          | fwd verify/repair ... | k_ll | redo of touched groups, reduce
//   aux  : bwd warm-up only (start vectors) | fused bwd + counts, list A | list B     | bwd verify / boundary-only repairs
//   walk : walks of the glued runs | chain of transfer matrices | run tiles (fwd)     -> their vectors gate the fused launch
//   cols : transfer-matrix columns (wave priority 2: they head the longest dependency chain)
// and p.fused == 2 (psmc_hip_estep_factored) puts the O(N) statistics kernel where the fused one is.
// p.overlap == 0 runs the same kernels back to back on one stream (bit-identical result).
int launch_fast(const EstepLaunch &p, FastReport *rep)
{
	if (p.n_chunks <= 0) return 0;
	(void)hipGetLastError(); // the value returned at the end must be about THESE launches
	const dim3 g(p.n_chunks), b(64);
	const bool ov = p.overlap != 0;
	hipStream_t sm = p.stream, sa = ov ? p.stream2 : p.stream, sx = ov ? p.stream3 : p.stream;
	rep->fwd_rounds = rep->bwd_rounds = rep->fwd_tiles = rep->bwd_tiles = rep->merged = rep->recounted = 0;
	rep->converged = 1;
	(void)hipMemsetAsync(p.d_warm, 0, 2 * sizeof(unsigned long long), sm);
	(void)hipMemsetAsync(p.d_touch_f, 0, 3 * sizeof(int) * (size_t)p.n_chunks, sm); // touch_f | touch_b | fmerge
	if (p.d_gate) (void)hipMemsetAsync(p.d_gate, 0, 2 * sizeof(int), sm); // started-counters of the walks and of the bulk grid
	if (p.ev[0]) (void)hipEventRecord(p.ev[0], sm);
	(void)hipEventRecord(p.evx[0], sm);
	if (ov) (void)hipStreamWaitEvent(sa, p.evx[0], 0); // parameters uploaded, flags cleared
	// ---- both speculative sweeps
	// Glued runs (structured sweeps only) are long sequential chains.  A WALK leaves only the boundary
	// vector of every tile it passes (no table stores: pure latency, untouched by the bulk's HBM traffic);
	// it starts now, beside the bulk, on a stream of its own.  Afterwards all tiles of the runs are recomputed
	// in parallel from those boundary vectors; they are flagged for the redo pass of the counts.
	const bool lw = p.structured && (p.n_long_f > 0 || p.n_long_b > 0) && (ov || p.fused);
	const bool lf = lw && p.n_long_f > 0, lb = lw && p.n_long_b > 0;
	const int ff0 = lf ? p.n_long_f : 0, fb0 = lb ? p.n_long_b : 0;
	// Shard-sized inputs ("merge1", api_fast.hip plan_fast): the bulk forward sweep and the backward warm-up pass are ONE grid
	// (k_sweep_struct), so that the dispatcher puts their waves on distinct SIMDs.  The backward chain's stream is then
	// idle during phase 1 and carries the whole dependent chain walks -> transfer-matrix chain -> run tiles -> back half ->
	// verify: every hand-over between streams costs 50-70 us (rocprofv3 timelines, profiles/r03_shard_timeline_*.txt),
	// which is nothing beside a 13 ms genome E-step and a fifth of a 1.5 ms chromosome E-step.
	const bool mg = p.merge1 && p.structured && p.fused && ov;
	hipStream_t sw = ov ? (mg ? sa : p.stream4) : sm, sk = ov ? p.stream5 : sm;
	if (lw) { // the few latency-critical waves first: they get SIMDs of their own before the bulk grid fills the device
		if (ov && sw != sa) (void)hipStreamWaitEvent(sw, p.evx[0], 0);
		launch_walks(p, sw); // short runs, and the start vector of every chain run
	}
	// dispatch order walks -> bulk -> transfer matrices (estep_struct.hip k_gate): the bulk grid's stream waits until every walk has its slot
	const bool gated = lw && ov && p.d_gate != nullptr && sw != sm;
	const int rw = p.lanes8 && p.ns == 64 && p.fused ? 8 : 4; // tiles per wave of the bulk grid
	if (gated) launch_gate(sm, p.d_gate, walk_blocks(p));
	if (mg) {
		launch_sweeps(p, sm, ff0, p.n_items_f - ff0, fb0, p.n_items_b - fb0 - p.n_B_b, true);
		if (p.merge) launch_fwd_struct(p, sm, 6, 0, p.n_fix_f); // the fix pass: the back half finds every bulk tile's forward table final
		(void)hipEventRecord(p.evx[1], sm);
	}
	if (lw) {
		if (p.n_kc > 0) { // long runs: transfer matrices of their tiles (a stream of their own), then the chain
			if (ov) (void)hipStreamWaitEvent(sk, p.evx[0], 0);
			// ... and the matrices wait until the bulk grid has been dispatched (merged: both directions; else the forward sweep)
			if (gated && sk != sm) launch_gate(sk, p.d_gate + 1, std::min(2048, mg ? (p.n_items_f - ff0 + rw - 1) / rw + (p.n_items_b - fb0 - p.n_B_b + rw - 1) / rw : (p.structured ? (p.n_items_f - ff0 + rw - 1) / rw : 0)));
			launch_kchain(p, sk, sw, p.evx[8]);
		}
		(void)hipEventRecord(p.evx[6], sw);
		// All boundary vectors of the runs exist: recompute their tiles right away, beside the bulk (forward on
		// the walk stream, backward on the transfer-matrix stream), so that the counts find them finished.
		if (lf) launch_fwd_struct(p, sw, 3, 0, p.n_mem_f);
		(void)hipEventRecord(p.evx[7], sw);
		if (ov) (void)hipStreamWaitEvent(sk, p.fused ? p.evx[7] : p.evx[6], 0); // fused: the backward pass reads the run tiles' X
		// unfused: the tiles of backward runs need their boundary vectors only.  fused == 1: the one pass over all tile groups
		// below covers them.  fused == 2: they also read X -- of tiles that may belong to the BULK forward sweep (a tile can
		// be glued backward only), so that launch follows the sweep, further down.
		if (lb && !p.fused) launch_bwd_struct(p, sk, 3, 0, p.n_mem_b);
		if (p.fused != 2) (void)hipEventRecord(p.evx[9], sk);
	}
	const bool one = p.structured && !p.fused; // both bulk sweeps in one launch (k_sweep_struct), tables of both directions
	if (mg) {
		// launched above
	} else if (one) {
		if (p.ev[7]) (void)hipEventRecord(p.ev[7], sm);
		launch_sweeps(p, sm, ff0, p.n_items_f - ff0, fb0, p.n_items_b - fb0, false);
		if (p.ev[6]) (void)hipEventRecord(p.ev[6], sm);
	} else if (p.structured) launch_fwd_struct(p, sm, 0, ff0, p.n_items_f - ff0);
	else launch_fwd<false>(p, sm);
	if (!mg && p.merge) launch_fwd_struct(p, sm, 6, 0, p.n_fix_f); // the fix pass over the bulk tiles (estep_struct.hip FwdCtl)
	if (!mg) (void)hipEventRecord(p.evx[1], sm);
	if (p.merge && lf) { // ... and over the run tiles: the head of a run is checked against the last row of a bulk tile
		if (ov && sw != sm) (void)hipStreamWaitEvent(sw, p.evx[1], 0);
		launch_fwd_struct(p, sw, 7, 0, p.n_mem_f);
		(void)hipEventRecord(p.evx[7], sw); // (what waits for the run tiles from here on waits for their check as well)
	}
	if (p.ev[5]) (void)hipEventRecord(p.ev[5], sm);
	if (one && ov) (void)hipStreamWaitEvent(sa, p.evx[1], 0); // the backward chain follows the same launch
	if (!one && p.ev[7]) (void)hipEventRecord(p.ev[7], sa);
	if (p.fused) {
		// Fused backward + counts (estep_fused.hip): a warm-up-only pass of the 4-tiles-per-wave sweep leaves
		// every bulk tile's start vector (beside the forward sweep, or in its grid); then one wave per group of four
		// tiles walks them backwards and feeds the matrix cores.  bt never reaches HBM.
		if (!mg) launch_bwd_struct(p, sa, 4, fb0, p.n_items_b - fb0 - p.n_B_b);
		// fused == 2 (item lists): the tiles of backward runs are a side pass on the transfer-matrix stream; it reads X of tiles
		// that may belong to the BULK forward sweep (a tile can be glued backward only), so it follows that sweep as well as
		// the run tiles.  (Until round 2 it followed the run tiles only and could read a bulk tile's X of the PREVIOUS E-step;
		// with the same parameters that is the same direction, and the per-position normaliser of the old kernels hid the
		// rest.  Found when the forward scale factors became part of the backward recursion.)  Tried in round 3 and removed:
		// the single tiles that are members of FORWARD runs in that side pass too, so that the main pass would wait for the
		// bulk sweeps only -- 13.7 against 10.8 ms: a full-length item that starts at the end of the runs' path, in a phase
		// whose vector units are saturated, ends long after the main pass (the fused back half has a second launch to hide
		// them in; this one has not).
		if (p.fused == 2 && lw) {
			if (lb) { if (ov) (void)hipStreamWaitEvent(sk, p.evx[1], 0); launch_bwd_acc(p, sk, 3, 0, p.n_mem_b); }
			(void)hipEventRecord(p.evx[9], sk);
		}
		(void)hipStreamWaitEvent(sa, p.evx[1], 0); // X of the bulk tiles; merged: and the start vectors of the same grid
		// boundary vectors and X of the run tiles (fused == 2, missing until round 2: a tile of a forward run is an ordinary
		// single tile of the bulk pass below, which could read its X before it was written; found with PSMC_HIP_POISON=vary).
		// fused == 1 with every run tile in the second list (runs_in_b): only that launch waits for the runs' path.
		const bool late = p.fused == 1 && p.runs_in_b && p.n_list_b > 0;
		if (lw && ov && sw != sa && !late) (void)hipStreamWaitEvent(sa, p.evx[7], 0);
		if (p.ev[8]) (void)hipEventRecord(p.ev[8], sa);
		if (p.fused == 2) { if (p.coarse > 1) launch_bwd_acc(p, sa, 6, 0, p.n_singles_b); else launch_bwd_acc(p, sa, 0, fb0, p.n_items_b - fb0); }
		else {
			launch_bwd_count(p, sa, 0, false);
			if (p.n_list_b > 0) { // its tiles start from the exit vectors list A left
				if (lw && ov && sw != sa && late) (void)hipStreamWaitEvent(sa, p.evx[7], 0);
				(void)hipEventRecord(p.evx[11], sa);
				(void)hipEventRecord(p.evx[12], sa);
				launch_bwd_count(p, sa, 1, false);
			}
		}
		if (p.ev[9]) (void)hipEventRecord(p.ev[9], sa);
		if (p.ev[6]) (void)hipEventRecord(p.ev[6], sa);
	} else {
		if (!one) {
			if (p.structured) launch_bwd_struct(p, sa, 0, fb0, p.n_items_b - fb0);
			else launch_bwd<false>(p, sa);
			if (p.ev[6]) (void)hipEventRecord(p.ev[6], sa);
		}
		if (ov) { // early expect over every tile once both sweeps exist
			(void)hipEventRecord(p.evx[2], sa);
			(void)hipStreamWaitEvent(sx, p.evx[1], 0); (void)hipStreamWaitEvent(sx, p.evx[2], 0);
			if (lw) { (void)hipStreamWaitEvent(sx, p.evx[7], 0); (void)hipStreamWaitEvent(sx, p.evx[9], 0); } // run tiles
			if (p.ev[8]) (void)hipEventRecord(p.ev[8], sx);
			launch_expect(p, sx, 0);
			if (p.ev[9]) (void)hipEventRecord(p.ev[9], sx);
			(void)hipEventRecord(p.evx[3], sx);
		}
	}
	// ---- verify / repair rounds: each direction advances on its own stream as soon as ITS
	// flagged-tile count is back (polled, so the faster chain never waits for the slower one)
	struct Chain { hipStream_t st; hipEvent_t rb; int slot; bool bwd, done, pending; int round; };
	Chain ch[2] = {{sm, p.evx[4], 0, false, false, false, 0}, {sa, p.evx[5], 1, true, false, false, 0}};
	auto post_verify = [&](Chain &c) -> int {
		(void)hipMemsetAsync(p.d_cnt + c.slot, 0, sizeof(int), c.st);
		(void)hipMemsetAsync(p.d_warm + c.slot, 0, sizeof(unsigned long long), c.st);
#define PSMC_LV(BW, S, MINE, DIRTY, CNT) hipLaunchKernelGGL((k_verify<BW, S>), g, b, 0, c.st, p.d_chunks, p.n_chunks, p.tol, p.d_f, \
			MINE, p.d_bexit, DIRTY, CNT, p.d_warm, (BW && !p.fused) ? 1 : 0, \
			(p.m_mis && c.round == 0 && (BW || !p.merge)) ? p.m_mis + (BW ? p.n_chunks : 0) : nullptr) /* (forward, with the fix pass: it has recorded them) */
		if (!c.bwd) { if (p.ns == 128) PSMC_LV(false, 128, p.d_entry, p.d_dirty, p.d_cnt); else PSMC_LV(false, 64, p.d_entry, p.d_dirty, p.d_cnt); }
		else { if (p.ns == 128) PSMC_LV(true, 128, p.d_bentry, p.d_dirty_b, p.d_cnt + 1); else PSMC_LV(true, 64, p.d_bentry, p.d_dirty_b, p.d_cnt + 1); }
#undef PSMC_LV
		launch_compact(p, c.st, c.bwd); // flagged tiles in ascending order + their number, straight into host-mapped memory
		if (hipEventRecord(c.rb, c.st) != hipSuccess) return -1;
		c.pending = true;
		return 0;
	};
	if (lw && ov) { if (sw != sm) (void)hipStreamWaitEvent(sm, p.evx[7], 0); (void)hipStreamWaitEvent(sa, p.evx[9], 0); } // run tiles done
	auto launch_ll = [&](hipStream_t st) { // the log-likelihood needs the forward tables only
		const int *fm = p.merge ? p.d_fmerge : nullptr;
		if (p.ns == 128) hipLaunchKernelGGL(k_ll<128>, g, b, 0, st, p.d_chunks, p.d_f, p.d_s, p.d_entry, p.d_LLpart, fm, p.d_finv);
		else hipLaunchKernelGGL(k_ll<64>, g, b, 0, st, p.d_chunks, p.d_f, p.d_s, p.d_entry, p.d_LLpart, fm, p.d_finv);
		PSMC_DBG("k_ll", p.n_chunks, 0, 0);
	};
	auto launch_reduce = [&](hipStream_t sm) { // (the parameter shadows the main stream on purpose: the optimistic tail reduces on another one)
		const int nS = p.n_chunks * p.n_sub, nC = p.fused == 1 ? (p.n_list_a + p.count_group - 1) / p.count_group + (p.n_list_b + p.count_group - 1) / p.count_group : nS; // fused: one C partial per group of four tiles
		if (p.fused == 2) {
			launch_reduce_factored(p, sm);
		} else if (p.ns == 128) {
			hipLaunchKernelGGL(k_reduce1<128>, dim3(128 * 128 / 256 + 1, RED_ROWS), dim3(256), 0, sm, p.d_Cpart, p.d_Epart, nC, nS, p.d_LLpart,
			                   p.n_chunks, p.d_stage);
			hipLaunchKernelGGL(k_reduce2<128>, dim3((128 * 128 + 3 * 128 + 1 + 255) / 256), dim3(256), 0, sm, p.d_stage, p.d_a, p.d_e,
			                   p.tiny_total, p.n_states, p.d_stats);
		} else {
			hipLaunchKernelGGL(k_reduce1<64>, dim3(17, RED_ROWS), dim3(256), 0, sm, p.d_Cpart, p.d_Epart, nC, nS, p.d_LLpart,
			                   p.n_chunks, p.d_stage);
			hipLaunchKernelGGL(k_reduce2<64>, dim3((STATS_LEN + 255) / 256), dim3(256), 0, sm, p.d_stage, p.d_a, p.d_e,
			                   p.tiny_total, p.n_states, p.d_stats);
		}
		if (p.ev[4]) (void)hipEventRecord(p.ev[4], sm);
	};
	if (post_verify(ch[0]) || post_verify(ch[1])) return -1;
	// OPTIMISTIC TAIL (round 3): after the first E-step of a context the speculation almost never fails (the plan has learned
	// where it does), so the log-likelihood and the reductions are enqueued NOW, behind the two verify kernels, instead of
	// after the host has read both flagged-tile counts -- two host round trips (~0.1 ms) leave the critical path.  If a
	// verify does flag tiles, the repair rounds run as before and the tail is simply enqueued again: it overwrites
	// d_LLpart, the stage buffer and d_stats from the repaired tables.  (With the fused / factored back half only: the
	// unfused one takes its counts in a separate pass that has its own dependencies.)  The tail runs on the counts' stream
	// of the unfused plan, which is idle here: on the main stream it would sit between the forward verify and a forward
	// REPAIR round, and the repair would wait for the whole back half instead of running beside it.
	static const bool recheck = getenv("PSMC_HIP_DEBUG_RECHECK") != nullptr; // the diagnostic below runs after the E-step: keep the ordinary tail
	const bool optimistic = p.fused != 0 && ov && !recheck;
	if (optimistic) {
		(void)hipStreamWaitEvent(sx, p.evx[4], 0); // forward verify (and compaction) done: the forward tables are final unless it flagged tiles
		launch_ll(sx);
		(void)hipEventRecord(p.evx[2], sa); (void)hipStreamWaitEvent(sx, p.evx[2], 0); // back half + backward verify done
		if (p.ev[1]) (void)hipEventRecord(p.ev[1], sx);
		if (p.ev[2]) (void)hipEventRecord(p.ev[2], sx);
		if (p.ev[3]) (void)hipEventRecord(p.ev[3], sx);
		launch_reduce(sx);
		(void)hipEventRecord(p.evx[10], sx);
	}
	bool ll_stale = !optimistic;
	while (!(ch[0].done && ch[1].done)) {
		bool progressed = false;
		for (int d = 0; d < 2; ++d) {
			Chain &c = ch[d];
			if (c.done || !c.pending) continue;
			const hipError_t q = ov ? hipEventQuery(c.rb) : hipEventSynchronize(c.rb);
			if (q == hipErrorNotReady) continue;
			if (q != hipSuccess) return -1;
			c.pending = false; progressed = true;
			const int nd = p.h_cnt[c.slot];
			if (nd == 0) { c.done = true; continue; }
			if (c.round >= p.max_rounds) { rep->converged = 0; c.done = true; continue; }
			++c.round;
			if (p.structured) {
				std::vector<int> *fl = c.bwd ? p.flagged_b : p.flagged_f;
				if (fl) for (int i = 0; i < nd; ++i) fl->push_back(p.h_ritems[(size_t)c.slot * 2 * p.n_chunks + 2 * i]);
			}
			if (!c.bwd) {
				rep->fwd_rounds++; rep->fwd_tiles += nd; ll_stale = true;
				if (p.structured) launch_fwd_struct(p, c.st, 1, 0, nd); else launch_fwd<true>(p, c.st);
			} else {
				rep->bwd_rounds++; rep->bwd_tiles += nd;
				if (p.fused == 2) launch_bwd_acc(p, c.st, 1, 0, nd);
				else if (p.fused) launch_bwd_struct(p, c.st, 5, 0, nd); // boundary vectors only; the counts follow below
				else if (p.structured) launch_bwd_struct(p, c.st, 1, 0, nd);
				else launch_bwd<true>(p, c.st);
			}
			if (post_verify(c)) return -1;
		}
		if (!progressed) __builtin_ia32_pause();
	}
	const bool repaired = rep->fwd_rounds + rep->bwd_rounds > 0;
	if (optimistic) (void)hipStreamWaitEvent(sm, p.evx[10], 0); // the caller synchronises the main stream; a second tail must not overtake the first
	if (optimistic && !repaired) return (int)hipGetLastError(); // the tail is already in flight
	if (hipStreamSynchronize(sm) != hipSuccess) return -1;
	if (p.ev[1]) (void)hipEventRecord(p.ev[1], sm);
	// ---- counts from the final tables
	if (ll_stale) launch_ll(sm);
	if (p.fused) {
		if (ov) { (void)hipEventRecord(p.evx[2], sa); (void)hipStreamWaitEvent(sm, p.evx[2], 0); } // backward chain done
		if (p.ev[2]) (void)hipEventRecord(p.ev[2], sm);
		// tiles whose X a forward repair rewrote after their counts were taken (only repairs set the touch flags)
		const bool recount = repaired;
		rep->recounted = recount ? 1 : 0;
		if (recount) {
			if (p.fused == 2) launch_bwd_acc(p, sm, 2, 0, p.n_chunks); else { launch_bwd_count(p, sm, 0, true); launch_bwd_count(p, sm, 1, true); }
		}
	} else if (ov) {
		(void)hipEventRecord(p.evx[2], sa); (void)hipStreamWaitEvent(sm, p.evx[2], 0); // backward chain done
		(void)hipStreamWaitEvent(sm, p.evx[3], 0);                                     // early expect done
		if (p.ev[2]) (void)hipEventRecord(p.ev[2], sm);
		if (repaired) launch_expect(p, sm, 1); // only tiles a repair rewrote
	} else {
		if (p.ev[2]) (void)hipEventRecord(p.ev[2], sm);
		if (p.ev[8]) (void)hipEventRecord(p.ev[8], sm);
		launch_expect(p, sm, 0);
		if (p.ev[9]) (void)hipEventRecord(p.ev[9], sm);
	}
	if (p.fused == 1 && recheck) {
		// diagnostic: recompute EVERY group from the final tables, every tile from its own start vector, and name the groups
		// whose partial differs from what the protocol (first pass + redo of the touched groups) left
		const int ga = (p.n_list_a + 3) / 4, gb = (p.n_list_b + 3) / 4, ng = ga + gb;
		const size_t gs = (size_t)p.ns * p.ns;
		std::vector<double> c0(gs * ng), c1(gs * ng);
		std::vector<int> tl(4 * (size_t)ng), tf(p.n_chunks), tb(p.n_chunks);
		std::vector<Chunk> ch(p.n_chunks);
		(void)hipDeviceSynchronize();
		(void)hipMemcpy(c0.data(), p.d_Cpart, sizeof(double) * c0.size(), hipMemcpyDeviceToHost);
		launch_bwd_count(p, sm, 0, false, true); launch_bwd_count(p, sm, 1, false, true);
		(void)hipDeviceSynchronize();
		(void)hipMemcpy(c1.data(), p.d_Cpart, sizeof(double) * c1.size(), hipMemcpyDeviceToHost);
		(void)hipMemcpy(tl.data(), p.d_ftiles, sizeof(int) * tl.size(), hipMemcpyDeviceToHost);
		(void)hipMemcpy(tf.data(), p.d_touch_f, sizeof(int) * tf.size(), hipMemcpyDeviceToHost);
		(void)hipMemcpy(tb.data(), p.d_touch_b, sizeof(int) * tb.size(), hipMemcpyDeviceToHost);
		(void)hipMemcpy(ch.data(), p.d_chunks, sizeof(Chunk) * ch.size(), hipMemcpyDeviceToHost);
		for (int g2 = 0; g2 < ng; ++g2) {
			double worst = 0.0, big = 0.0;
			for (size_t i = 0; i < gs; ++i) { big = std::max(big, std::fabs(c1[g2 * gs + i])); worst = std::max(worst, std::fabs(c0[g2 * gs + i] - c1[g2 * gs + i])); }
			if (worst > 1e-9 * big) {
				fprintf(stderr, "[psmc_hip] RECHECK group %d (list %c) differs by %.2e of %.2e; rounds %d/%d; tiles:", g2, g2 < ga ? 'A' : 'B', worst, big, rep->fwd_rounds, rep->bwd_rounds);
				for (int r = 0; r < 4; ++r) {
					const int en = tl[4 * (size_t)g2 + r];
					if (en < 0) { fprintf(stderr, " -"); continue; }
					const int t = en & ~(1 << 30);
					fprintf(stderr, " %d%s[lo %d hi %d L %d off %lld mult %d flags %d warm-up %d/%d tf %d tb %d; above tf %d tb %d]", t, (en & (1 << 30)) ? "^" : "", ch[t].lo, ch[t].hi, ch[t].L, (long long)ch[t].off,
					        (int)ch[t].mult, (int)ch[t].flags, (int)ch[t].wf, (int)ch[t].wb, tf[t], tb[t], t + 1 < p.n_chunks ? tf[t + 1] : -1, t + 1 < p.n_chunks ? tb[t + 1] : -1);
				}
				fprintf(stderr, "\n");
			}
		}
	}
	if (p.ev[3]) (void)hipEventRecord(p.ev[3], sm);
	launch_reduce(sm);
	return (int)hipGetLastError();
}

} // namespace psmc
