// estep_factored.hip -- FAST mode, structured matrices: the sufficient statistics in O(N) per bin.
//
// With a[k][l] = P_k qa_l (l<k), R_k c_l (l>k) the EM objective needs of the N x N expected transition
// counts A only five vectors (psmc_amd/host/mstep.c neg_Q_fast):
//   SL_k = sum_{l<k} A[k][l]   SU_k = sum_{l>k} A[k][l]   DG_k = A[k][k]
//   CL_l = sum_{k>l} A[k][l]   CU_l = sum_{k<l} A[k][l]
// and with A[k][l] = a[k][l] sum_p w_p X_p[k] bt_{p+1}[l] every one of them is a scan away from what the
// backward step computes anyway (z = bt_{p+1}; w_p = 1/I for every p, see below):
//   SL_k = P_k   sum_p w_p X_p[k] PREexcl_k(z.qa)      SU_k = R_k sum_p w_p X_p[k] SUFexcl_k(z.c)
//   CL_l = qa_l  sum_p w_p z_l    SUFexcl_l(X_p.P)     CU_l = c_l sum_p w_p z_l    PREexcl_l(X_p.R)
//   DG_k = a_kk  sum_p w_p X_p[k] z_k
// So the counts GEMM (2 N^2 flop per bin, 1 KB of table reads per bin, the bt table) disappears: the wave
// that walks four tiles backwards reads X, keeps bt in registers and accumulates 7 x 4 numbers per lane.
// HBM per bin: 8N (write X) + 8N (read X).  Entry point psmc_hip_estep_factored (include/psmc_hip.h).
//
// Scaling (round 2).  With y_p = a bt_{p+1} the number I_p = sum_k X_p[k] y_p[k] is what turns X_p (x) bt_{p+1} into a posterior,
// and it is NOT a new number at every position: between two normalising positions it is constant (an algebraic identity of
// the two recursions), and across one it changes by a known factor, I_{p-1} = I_p sb_p / inv_p (sb_p = 1/sum(bt_{p+1}), bt's own
// scale factor; inv_p = the forward sweep's, a power of two from the d_s table).  So nothing is normalised per position
// (no row sum of g, no reciprocal, no weight multiplications: round 1's step had all three): the partial sums are kept in
// units of the current I, rescaled by sb_p / inv_p at the normalising positions (p % 4 == 0), and divided by the I of the
// last position when the tile stores them.  bt itself keeps its own scaling: an earlier version of this step let bt follow
// the forward scale factors as khmm.c:228-235 does (I is then constant over the whole tile and even the rescaling
// goes away), but then the exit vector a tile hands on depends on a forward table, and a table that a forward
// repair is rewriting can be read torn -- different lanes of a row saw different scale factors, the exit vector
// came out bent, and the boundary repair of the tile below copied it (1 E-step in 1 500 off by up to 3e-2 in a stress
// of tiny tiles, profiles/experiments/dbg_flaky_tiling.py; never with this version).
#include <hip/hip_runtime.h>
#include "wave_prims.h"
#include "struct_prims.h"
#include "psmc_hip_internal.h"

namespace psmc {

struct SweepItemA { int first, count; };
constexpr int NACC = 7; // SL SU DG CL CU E0 E1

__device__ __forceinline__ int64_t readlane_i64a(int64_t v, int lane) {
	const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, lane);
	const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)v >> 32), lane);
	return (int64_t)(((unsigned long long)hi << 32) | lo);
}

template <int NPL> __device__ __forceinline__ double lane_sum_a(const double (&x)[NPL]) {
	double t = (x[0] + x[1]) + (x[2] + x[3]);
	if constexpr (NPL == 8) t = t + ((x[4] + x[5]) + (x[6] + x[7]));
	return t;
}

// One position p of one row: x = bt_{p+1} on entry, bt_p on exit; X = X_p.  NORM: p % NORM_EVERY == 0.
// NPLA_ states per lane: 4 (up to 64 states) or 8 (up to 128, `-p "64*2"`); 16 lanes = one tile either way.
// LPT lanes per tile: 16 (a DPP row), or 8 with 8 states per lane -- 64 states as eight tiles per wave (round 5 experiment, removed)
template <bool NORM, int NPLA = 4, int LPT = 16>
__device__ __forceinline__ void acc_step(const StructParN<NPLA> &sc, const double *lds_e, const double *lds_m, int k0, int sym,
                                         const double (&X)[NPLA], double (&x)[NPLA], double inv, double (&acc)[NACC][NPLA],
                                         double &I_lane, const Half8Masks &hm = Half8Masks())
{
	constexpr int SA = LPT * NPLA;
	double ev[NPLA];
	ev_load<NPLA, LPT>(lds_e + sym * SA, k0, ev);
	const d2v_t mk = *reinterpret_cast<const d2v_t *>(lds_m + 2 * sym); // (1,0) hom, (0,1) het, (0,0) missing
	double f = 1.0;
	if (NORM) {
		// bt keeps its OWN scaling, sb_p = 1/sum(bt_{p+1}): the vectors handed from tile to tile (bexit) must not depend on a
		// forward table that a repair may be rewriting.  I then changes at this position, I_{p-1} = I_p sb_p / inv_p (inv_p = the
		// forward sweep's scale factor, a power of two): the partial sums, kept in units of the current I, are rescaled below.
		const double tot = LPT == 8 ? half8_sum(lane_sum_a<NPLA>(x)) : row_sum16(lane_sum_a<NPLA>(x));
		const double sb = rcp_newton(tot);
#pragma unroll
		for (int i = 0; i < NPLA; ++i) ev[i] *= sb;
		f = sb * pow2_rcp(inv);
	}
	// lane-local inclusive scans: z.c / z.qa (the backward step) and X.P / X.R (the column sums)
	double su[NPLA + 1], pv[NPLA + 1], sx[NPLA + 1], px[NPLA + 1];
	su[NPLA] = 0.0; sx[NPLA] = 0.0; pv[0] = 0.0; px[0] = 0.0; // pv / px are shifted by one: pv[i+1] = inclusive at i
#pragma unroll
	for (int i = NPLA - 1; i >= 0; --i) { su[i] = __builtin_fma(x[i], sc.mS[i], su[i + 1]); sx[i] = __builtin_fma(X[i], sc.wP[i], sx[i + 1]); }
#pragma unroll
	for (int i = 0; i < NPLA; ++i) { pv[i + 1] = __builtin_fma(x[i], sc.mP[i], pv[i]); px[i + 1] = __builtin_fma(X[i], sc.wS[i], px[i]); }
	double ES, EP, EX, PX;
	if constexpr (LPT == 8) { ES = half8_excl_suffix(su[0], hm); EP = half8_excl_prefix(pv[NPLA], hm); EX = half8_excl_suffix(sx[0], hm); PX = half8_excl_prefix(px[NPLA], hm); }
	else { ES = row_excl_suffix(su[0]); EP = row_excl_prefix(pv[NPLA]); EX = row_excl_suffix(sx[0]); PX = row_excl_prefix(px[NPLA]); }
	double Il = 0.0;
#pragma unroll
	for (int i = 0; i < NPLA; ++i) {
		const double t = __builtin_fma(sc.wS[i], su[i], __builtin_fma(sc.wP[i], pv[i + 1], sc.dd[i] * x[i]));
		const double y = __builtin_fma(sc.wS[i], ES, __builtin_fma(sc.wP[i], EP, t)); // (a bt_{p+1})[k]
		const double gk = X[i] * y;                                                    // I * posterior of state k at p
		acc[0][i] = __builtin_fma(X[i], EP + pv[i], acc[0][i]);     // SL: strictly below k
		acc[1][i] = __builtin_fma(X[i], ES + su[i + 1], acc[1][i]); // SU: strictly above k
		acc[2][i] = __builtin_fma(X[i], x[i], acc[2][i]);           // DG
		acc[3][i] = __builtin_fma(x[i], EX + sx[i + 1], acc[3][i]); // CL: rows k > l
		acc[4][i] = __builtin_fma(x[i], PX + px[i], acc[4][i]);     // CU: rows k < l
		acc[5][i] = __builtin_fma(gk, mk.x, acc[5][i]);
		acc[6][i] = __builtin_fma(gk, mk.y, acc[6][i]);
		Il += gk;
		x[i] = y * ev[i];
	}
	I_lane = Il; // this lane's share of I at position p
	if (NORM) { // ... now in units of I_{p-1}
#pragma unroll
		for (int q = 0; q < NACC; ++q)
#pragma unroll
			for (int i = 0; i < NPLA; ++i) acc[q][i] *= f;
		I_lane *= f;
	}
}
// 1/I of a tile (n_pos = 0: an empty tile, whose partials are zero anyway)
template <int LPT = 16> __device__ __forceinline__ double tile_inv_I(double I_lane, int n_pos)
{
	const double tot = LPT == 8 ? half8_sum(I_lane) : row_sum16(I_lane);
	return n_pos > 0 ? rcp_newton(tot) : 1.0;
}
// scaled partials of one tile: the constant factors of the five sums and the multiplicity
template <int NPLA = 4, int LPT = 16>
__device__ __forceinline__ void acc_store(const StructParN<NPLA> &sc, double mult, double (&acc)[NACC][NPLA], double *out)
{
	constexpr int SA = LPT * NPLA;
#pragma unroll
	for (int i = 0; i < NPLA; ++i) {
		const double akk = sc.dd[i] + sc.wP[i] * sc.mP[i] + sc.wS[i] * sc.mS[i]; // a[k][k]
		acc[0][i] *= sc.wP[i] * mult; acc[1][i] *= sc.wS[i] * mult; acc[2][i] *= akk * mult;
		acc[3][i] *= sc.mP[i] * mult; acc[4][i] *= sc.mS[i] * mult; acc[5][i] *= mult; acc[6][i] *= mult;
	}
#pragma unroll
	for (int q = 0; q < NACC; ++q) storeN<NPLA>(out + q * SA, acc[q]);
}

// mode 0: tiles items[0..n) from bentry;  mode 1: flagged tiles items[0..n) from the exit vector of the tile
// above (which becomes their bentry);  mode 2: every tile b < n whose X a forward repair rewrote, from bentry.
template <int NPLA>
__global__ __launch_bounds__(64, NPLA == 4 ? 2 : 1) void k_bwd_acc_struct(const double *__restrict__ sp, const double *__restrict__ e,
                                                            const double *__restrict__ invd, const uint8_t *__restrict__ obs,
                                                            const Chunk *__restrict__ chunks, const SweepItemA *__restrict__ items,
                                                            int n, int mode, const double *__restrict__ f,
                                                            double *__restrict__ bentry, double *__restrict__ bexit,
                                                            double *__restrict__ part, const int *__restrict__ touch_f,
                                                            int *__restrict__ touch_b)
{
	constexpr int SA = 16 * NPLA;
	__shared__ double lds_e[4 * SA], lds_m[8]; // e rows: hom, het, 1, 1;  emission-count masks per symbol
	const int lane = threadIdx.x, row = lane >> 4, m = lane & 15, k0 = NPLA * m;
#pragma unroll
	for (int i = lane; i < SA; i += 64) { const int q = ev_slot<NPLA>(i); lds_e[q] = e[i]; lds_e[SA + q] = e[SA + i]; lds_e[2 * SA + q] = 1.0; lds_e[3 * SA + q] = 1.0; } // ev_load layout
	if (lane < 8) lds_m[lane] = (lane == 0 || lane == 3) ? 1.0 : 0.0;
	__syncthreads();
	const int slot = blockIdx.x * 4 + row;
	int tile; bool valid = slot < n;
	if (mode == 2) { tile = valid ? slot : 0; valid = valid && touch_f[tile] != 0; }
	else tile = items[valid ? slot : 0].first;
	if (!__any(valid)) return;
	if (mode == 1) __builtin_amdgcn_s_setprio(3);
	const Chunk c = chunks[tile];
	const int L = c.L, lo = c.lo, top = min(c.hi, L - 1);
	const bool work = valid && top >= lo; // a tile holding only position L owns no transition: zero partials
	const double *fo = f + c.off * SA + k0;
	const double *io = invd + c.off;
	StructParN<NPLA> sc; // backward: mS = c, wS = R, mP = qa, wP = P
	loadN<NPLA>(sp + 3 * SA + k0, sc.mS); loadN<NPLA>(sp + SA + k0, sc.wS);
	loadN<NPLA>(sp + 2 * SA + k0, sc.mP); loadN<NPLA>(sp + k0, sc.wP); loadN<NPLA>(sp + 4 * SA + k0, sc.dd);
	double x[NPLA];
	if (mode == 1) {
		loadN<NPLA>(bexit + (int64_t)(tile + 1) * SA + k0, x);
		if (work) { storeN<NPLA>(bentry + (int64_t)tile * SA + k0, x); if (m == 0) touch_b[tile] = 1; }
	} else {
		loadN<NPLA>(bentry + (int64_t)tile * SA + k0, x);
	}
	double acc[NACC][NPLA], accI = 1.0;
#pragma unroll
	for (int q = 0; q < NACC; ++q)
#pragma unroll
		for (int i = 0; i < NPLA; ++i) acc[q][i] = 0.0;
	const int n_pos = work ? top - lo + 1 : 0;
	auto load_inv = [&](int g) { return io[min(max(4 * g + 4, lo), max(top, lo)) - 1]; }; // 1/d of the group's position 4g + 4 (unused when that lies outside the tile)
	// groups of four positions 4g+1 .. 4g+4 (indices 4g .. 4g+3), highest first; the group's last
	// position (p % 4 == 0) carries the scale factor
	const int g_hi = work ? (top - 1) >> 2 : -1, g_lo = work ? (lo - 1) >> 2 : 0;
	const int ng = g_hi - g_lo + 1;
	const int64_t off0 = readlane_i64a(c.off, 0), off1 = readlane_i64a(c.off, 16), off2 = readlane_i64a(c.off, 32), off3 = readlane_i64a(c.off, 48);
	const int gh0 = __builtin_amdgcn_readlane(g_hi, 0), gh1 = __builtin_amdgcn_readlane(g_hi, 16), gh2 = __builtin_amdgcn_readlane(g_hi, 32), gh3 = __builtin_amdgcn_readlane(g_hi, 48);
	const int n0 = __builtin_amdgcn_readlane(ng, 0), n1 = __builtin_amdgcn_readlane(ng, 16), n2 = __builtin_amdgcn_readlane(ng, 32), n3 = __builtin_amdgcn_readlane(ng, 48);
	const int ng_max = max(max(n0, n1), max(n2, n3));
	auto load_group = [&](int g, double (&X)[4][NPLA]) { // X rows of the group's positions, clamped into the tile
#pragma unroll
		for (int j = 0; j < 4; ++j) {
			const int p = min(max(4 * g + j + 1, lo), max(top, lo));
			loadN<NPLA>(fo + (int64_t)(p - 1) * SA, X[j]);
		}
	};
	double Xg[4][NPLA];
	load_group(max(g_hi, 0), Xg);
	double inv_cur = load_inv(max(g_hi, 0));
	if constexpr (NPLA == 8) {
		// 128 states: 7 x 8 accumulators and 5 x 8 constants per lane leave no room for a second X buffer -- every row
		// is reloaded for the next group as soon as its step has used it (as in estep_fused.hip)
		for (int gi = 0; gi < ng_max; ++gi) {
			const unsigned w0 = *reinterpret_cast<const unsigned *>(obs + off0 + 4 * (int64_t)max(gh0 - min(gi, max(n0 - 1, 0)), 0));
			const unsigned w1 = *reinterpret_cast<const unsigned *>(obs + off1 + 4 * (int64_t)max(gh1 - min(gi, max(n1 - 1, 0)), 0));
			const unsigned w2 = *reinterpret_cast<const unsigned *>(obs + off2 + 4 * (int64_t)max(gh2 - min(gi, max(n2 - 1, 0)), 0));
			const unsigned w3 = *reinterpret_cast<const unsigned *>(obs + off3 + 4 * (int64_t)max(gh3 - min(gi, max(n3 - 1, 0)), 0));
			const unsigned w = row == 0 ? w0 : (row == 1 ? w1 : (row == 2 ? w2 : w3));
			if (gi < ng) {
				const int g = g_hi - gi;
				const double inv = inv_cur;
				if (gi + 1 < ng) inv_cur = load_inv(g - 1);
#pragma unroll
				for (int j = 3; j >= 0; --j) {
					const int p = 4 * g + j + 1;
					if (p <= top && p >= lo) {
						const int sym = (int)((w >> (8 * j)) & 3u);
						if (j == 3) acc_step<true, NPLA>(sc, lds_e, lds_m, k0, sym, Xg[j], x, inv, acc, accI);
						else acc_step<false, NPLA>(sc, lds_e, lds_m, k0, sym, Xg[j], x, inv, acc, accI);
						if (p == lo) storeN<NPLA>(bexit + (int64_t)tile * SA + k0, x);
					}
					if (gi + 1 < ng) { // this row of the next group (positions 4(g-1)+j+1), clamped into the tile
						const int pn = min(max(4 * (g - 1) + j + 1, lo), max(top, lo));
						loadN<NPLA>(fo + (int64_t)(pn - 1) * SA, Xg[j]);
					}
				}
			}
		}
		const double iI = tile_inv_I(accI, n_pos);
		if (valid) acc_store<NPLA>(sc, (double)c.mult * iI, acc, part + (int64_t)tile * (NACC * SA) + k0);
		return;
	}
	double Xn[4][NPLA];
	for (int gi = 0; gi < ng_max; ++gi) {
		// the group's four symbols of every row: scalar loads (see estep_struct.hip row_symbols)
		const unsigned w0 = *reinterpret_cast<const unsigned *>(obs + off0 + 4 * (int64_t)max(gh0 - min(gi, max(n0 - 1, 0)), 0));
		const unsigned w1 = *reinterpret_cast<const unsigned *>(obs + off1 + 4 * (int64_t)max(gh1 - min(gi, max(n1 - 1, 0)), 0));
		const unsigned w2 = *reinterpret_cast<const unsigned *>(obs + off2 + 4 * (int64_t)max(gh2 - min(gi, max(n2 - 1, 0)), 0));
		const unsigned w3 = *reinterpret_cast<const unsigned *>(obs + off3 + 4 * (int64_t)max(gh3 - min(gi, max(n3 - 1, 0)), 0));
		const unsigned w = row == 0 ? w0 : (row == 1 ? w1 : (row == 2 ? w2 : w3));
		if (gi < ng) {
			const int g = g_hi - gi;
			const double inv = inv_cur;
			if (gi + 1 < ng) { load_group(g - 1, Xn); inv_cur = load_inv(g - 1); }
#pragma unroll
			for (int j = 3; j >= 0; --j) {
				const int p = 4 * g + j + 1;
				if (p > top || p < lo) continue;
				const int sym = (int)((w >> (8 * j)) & 3u);
				if (j == 3) acc_step<true, NPLA>(sc, lds_e, lds_m, k0, sym, Xg[j], x, inv, acc, accI);
				else acc_step<false, NPLA>(sc, lds_e, lds_m, k0, sym, Xg[j], x, inv, acc, accI);
				if (p == lo) storeN<NPLA>(bexit + (int64_t)tile * SA + k0, x);
			}
#pragma unroll
			for (int j = 0; j < 4; ++j)
#pragma unroll
				for (int i = 0; i < NPLA; ++i) Xg[j][i] = Xn[j][i];
		}
	}
	const double iI = tile_inv_I(accI, n_pos);
	if (valid) acc_store<NPLA>(sc, (double)c.mult * iI, acc, part + (int64_t)tile * (NACC * SA) + k0);
}

// (Round 5 measured this back half with EIGHT tiles per wave, 8 lanes x 8 states, k_bwd_acc_struct_h8 / option "lanes8b": 5.17 against 4.51 ms --
// 7 x 8 accumulators + 5 x 8 constants per lane spill; removed in round 6, profiles/r05_lanes8b_ab.txt keeps the record.)
constexpr int NPLA = 4, SA = 64; // the checkpointed variant below is the 64-state one

// The same without the X table: the forward sweep left only the checkpoints X_p, p % 8 == 0 (SWEEP_CKPT in
// estep_struct.hip), and the row recomputes the eight X of a block from the checkpoint below it before it walks
// the block backwards -- 84 more VALU instructions per step in exchange for 7/8 of the table traffic
// (HBM per bin: 8N/8 written + 8N/8 read).  The first block of a tile starts from the tile's own start vector
// `entry` (what its checkpoints were computed from; position 1: X_1 itself, which the sweep stores), so a forward
// repair of the tile below never races with this kernel.  The recomputation applies the forward sweep's scale factors
// at p % 4 == 0 exactly as the sweep did (the backward recursion relies on X and bt sharing them, see the top of the file).
// Needs tile_len % 8 == 0 (every lo = 1 mod 8): only the top block of a segment's last tile is partial.
template <bool NORM>
__device__ __forceinline__ void fwd_recompute_step(const StructParN<NPLA> &sc, const double *lds_e, int k0, int sym, double inv, double (&x)[NPLA])
{	// forward roles of the five vectors: mS = P (bwd wP), wS = qa (bwd mP), mP = R (bwd wS), wP = c (bwd mS)
	double ev[NPLA], su[NPLA], pv[NPLA];
	ev_load<NPLA>(lds_e + sym * SA, k0, ev);
	if (NORM) {
#pragma unroll
		for (int i = 0; i < NPLA; ++i) ev[i] *= inv;
	}
	su[NPLA - 1] = x[NPLA - 1] * sc.wP[NPLA - 1];
#pragma unroll
	for (int i = NPLA - 2; i >= 0; --i) su[i] = __builtin_fma(x[i], sc.wP[i], su[i + 1]);
	pv[0] = x[0] * sc.wS[0];
#pragma unroll
	for (int i = 1; i < NPLA; ++i) pv[i] = __builtin_fma(x[i], sc.wS[i], pv[i - 1]);
	const double ES = row_excl_suffix(su[0]), EP = row_excl_prefix(pv[NPLA - 1]);
#pragma unroll
	for (int i = 0; i < NPLA; ++i) {
		const double t = __builtin_fma(sc.mP[i], su[i], __builtin_fma(sc.mS[i], pv[i], sc.dd[i] * x[i]));
		x[i] = __builtin_fma(sc.mP[i], ES, __builtin_fma(sc.mS[i], EP, t)) * ev[i];
	}
}

__global__ __launch_bounds__(64, 2) void k_bwd_acc_ckpt(const double *__restrict__ sp, const double *__restrict__ e,
                                                          const double *__restrict__ invd, const uint8_t *__restrict__ obs,
                                                          const Chunk *__restrict__ chunks, const SweepItemA *__restrict__ items,
                                                          int n, int mode, const double *__restrict__ f,
                                                          const double *__restrict__ entry, double *__restrict__ bentry,
                                                          double *__restrict__ bexit, double *__restrict__ part,
                                                          const int *__restrict__ touch_f, int *__restrict__ touch_b)
{
	__shared__ double lds_e[4 * SA], lds_m[8];
	__shared__ double lds_x[4 * 8 * SA]; // the block's X, private to the lane that wrote it: [row][j][half][2 m + i]
	const int lane = threadIdx.x, row = lane >> 4, m = lane & 15, k0 = NPLA * m;
	{ const int q = ev_slot<4>(lane); lds_e[q] = e[lane]; lds_e[SA + q] = e[SA + lane]; lds_e[2 * SA + q] = 1.0; lds_e[3 * SA + q] = 1.0; } // ev_load layout
	if (lane < 8) lds_m[lane] = (lane == 0 || lane == 3) ? 1.0 : 0.0;
	__syncthreads();
	double *xs = lds_x + row * (8 * SA) + 2 * m;
	const int slot = blockIdx.x * 4 + row;
	int tile; bool valid = slot < n;
	if (mode == 2) { tile = valid ? slot : 0; valid = valid && touch_f[tile] != 0; }
	else tile = items[valid ? slot : 0].first;
	if (!__any(valid)) return;
	if (mode == 1) __builtin_amdgcn_s_setprio(3);
	const Chunk c = chunks[tile];
	const int L = c.L, lo = c.lo, top = min(c.hi, L - 1);
	const bool work = valid && top >= lo;
	const double *fo = f + c.off * SA + k0;
	const double *io = invd + c.off;
	StructParN<NPLA> sc; // backward: mS = c, wS = R, mP = qa, wP = P
	loadN<NPLA>(sp + 3 * SA + k0, sc.mS); loadN<NPLA>(sp + SA + k0, sc.wS);
	loadN<NPLA>(sp + 2 * SA + k0, sc.mP); loadN<NPLA>(sp + k0, sc.wP); loadN<NPLA>(sp + 4 * SA + k0, sc.dd);
	double x[NPLA];
	if (mode == 1) {
		loadN<NPLA>(bexit + (int64_t)(tile + 1) * SA + k0, x);
		if (work) { storeN<NPLA>(bentry + (int64_t)tile * SA + k0, x); if (m == 0) touch_b[tile] = 1; }
	} else {
		loadN<NPLA>(bentry + (int64_t)tile * SA + k0, x);
	}
	double acc[NACC][NPLA], accI = 1.0;
#pragma unroll
	for (int q = 0; q < NACC; ++q)
#pragma unroll
		for (int i = 0; i < NPLA; ++i) acc[q][i] = 0.0;
	const int n_pos = work ? top - lo + 1 : 0;
	// blocks of eight positions 8b+1 .. 8b+8 (indices 8b .. 8b+7), highest first
	const int b_hi = work ? (top - 1) >> 3 : -1, b_lo = work ? (lo - 1) >> 3 : 0;
	const int nb = b_hi - b_lo + 1;
	const int64_t off0 = readlane_i64a(c.off, 0), off1 = readlane_i64a(c.off, 16), off2 = readlane_i64a(c.off, 32), off3 = readlane_i64a(c.off, 48);
	const int bh0 = __builtin_amdgcn_readlane(b_hi, 0), bh1 = __builtin_amdgcn_readlane(b_hi, 16), bh2 = __builtin_amdgcn_readlane(b_hi, 32), bh3 = __builtin_amdgcn_readlane(b_hi, 48);
	const int n0 = __builtin_amdgcn_readlane(nb, 0), n1 = __builtin_amdgcn_readlane(nb, 16), n2 = __builtin_amdgcn_readlane(nb, 32), n3 = __builtin_amdgcn_readlane(nb, 48);
	const int nb_max = max(max(n0, n1), max(n2, n3));
	// start vector of block b: X_{8b}; the tile's first block: its entry vector, or X_1 itself when lo == 1
	const double *ck_first = lo > 1 ? entry + (int64_t)tile * SA + k0 : fo;
	// ... and the forward scale factors of its positions 8b+4 and 8b+8 (anything where those lie above the tile's top: the
	// recomputed X from there on are not used)
	auto load_ck = [&](int b, double (&ck)[NPLA], double &i4, double &i8) {
		const double *src = b <= b_lo ? ck_first : fo + (int64_t)(8 * b - 1) * SA;
		loadN<NPLA>(src, ck);
		i4 = io[8 * b + 3]; i8 = io[8 * b + 7];
	};
	double ck[NPLA], ckn[NPLA], inv4, inv8, inv4n = 1.0, inv8n = 1.0;
	load_ck(max(b_hi, b_lo), ck, inv4, inv8);
	for (int bi = 0; bi < nb_max; ++bi) {
		// the block's eight symbols of every row: scalar loads (see estep_struct.hip row_symbols)
		const uint2 w0 = *reinterpret_cast<const uint2 *>(obs + off0 + 8 * (int64_t)max(bh0 - min(bi, max(n0 - 1, 0)), 0));
		const uint2 w1 = *reinterpret_cast<const uint2 *>(obs + off1 + 8 * (int64_t)max(bh1 - min(bi, max(n1 - 1, 0)), 0));
		const uint2 w2 = *reinterpret_cast<const uint2 *>(obs + off2 + 8 * (int64_t)max(bh2 - min(bi, max(n2 - 1, 0)), 0));
		const uint2 w3 = *reinterpret_cast<const uint2 *>(obs + off3 + 8 * (int64_t)max(bh3 - min(bi, max(n3 - 1, 0)), 0));
		const unsigned wa = row == 0 ? w0.x : (row == 1 ? w1.x : (row == 2 ? w2.x : w3.x));
		const unsigned wb = row == 0 ? w0.y : (row == 1 ? w1.y : (row == 2 ? w2.y : w3.y));
		if (bi < nb) {
			const int b = b_hi - bi;
			if (bi + 1 < nb) load_ck(b - 1, ckn, inv4n, inv8n);
			// ---- forward: X of the block's positions from the checkpoint below them
			const bool is_x1 = b == b_lo && lo == 1; // the start vector IS position 8b+1 = 1
			double t[NPLA];
#pragma unroll
			for (int i = 0; i < NPLA; ++i) t[i] = ck[i];
#pragma unroll
			for (int j = 0; j < 8; ++j) {
				const int sym = (int)(((j < 4 ? wa : wb) >> (8 * (j & 3))) & 3u);
				if (j == 0) {
					double u[NPLA];
#pragma unroll
					for (int i = 0; i < NPLA; ++i) u[i] = t[i];
					fwd_recompute_step<false>(sc, lds_e, k0, sym, 1.0, u);
#pragma unroll
					for (int i = 0; i < NPLA; ++i) t[i] = is_x1 ? t[i] : u[i];
				} else if (j == 3) fwd_recompute_step<true>(sc, lds_e, k0, sym, inv4, t);
				else if (j == 7) fwd_recompute_step<true>(sc, lds_e, k0, sym, inv8, t);
				else fwd_recompute_step<false>(sc, lds_e, k0, sym, 1.0, t);
				*reinterpret_cast<d2v_t *>(xs + j * SA) = (d2v_t){t[0], t[1]};
				*reinterpret_cast<d2v_t *>(xs + j * SA + 32) = (d2v_t){t[2], t[3]};
			}
			// ---- backward through the block
#pragma unroll
			for (int j = 7; j >= 0; --j) {
				const int p = 8 * b + j + 1;
				if (p > top || p < lo) continue;
				const int sym = (int)(((j < 4 ? wa : wb) >> (8 * (j & 3))) & 3u);
				const d2v_t xa = *reinterpret_cast<const d2v_t *>(xs + j * SA), xb = *reinterpret_cast<const d2v_t *>(xs + j * SA + 32);
				const double X[NPLA] = {xa.x, xa.y, xb.x, xb.y};
				if ((j & 3) == 3) acc_step<true>(sc, lds_e, lds_m, k0, sym, X, x, j == 7 ? inv8 : inv4, acc, accI);
				else acc_step<false>(sc, lds_e, lds_m, k0, sym, X, x, 1.0, acc, accI);
				if (p == lo) storeN<NPLA>(bexit + (int64_t)tile * SA + k0, x);
			}
#pragma unroll
			for (int i = 0; i < NPLA; ++i) ck[i] = ckn[i];
			inv4 = inv4n; inv8 = inv8n;
		}
	}
	const double iI = tile_inv_I(accI, n_pos);
	if (valid) acc_store(sc, (double)c.mult * iI, acc, part + (int64_t)tile * (NACC * SA) + k0);
}

// Fixed-order two-stage sum over the tiles: stage[y][q*64+k] = sum of the tiles j = y (mod RED_ROWS), then
// out (unpadded) = SL | SU | DG | CL | CU | E0 | E1 | LL; the HMM_TINY seeds of khmm.c:305-308 are added per
// cell (k cells below the diagonal of row k, ...).
template <int SA>
__global__ __launch_bounds__(SA) void k_reduce_factored1(const double *__restrict__ part, int n_tiles,
                                                           const double *__restrict__ LLpart, double *__restrict__ stage)
{
	constexpr int FSL = NACC * SA + 1;
	const int k = threadIdx.x, q = blockIdx.x, y = blockIdx.y;
	if (q < NACC) {
		double s = 0.0;
		for (int j = y; j < n_tiles; j += RED_ROWS) s += part[(int64_t)j * (NACC * SA) + q * SA + k];
		stage[(int64_t)y * FSL + q * SA + k] = s;
	} else if (k == 0) {
		double s = 0.0;
		for (int j = y; j < n_tiles; j += RED_ROWS) s += LLpart[j];
		stage[(int64_t)y * FSL + NACC * SA] = s;
	}
}
template <int SA>
__global__ __launch_bounds__(SA) void k_reduce_factored2(const double *__restrict__ stage, double tiny_total, int n,
                                                           double *__restrict__ out)
{
	constexpr int FSL = NACC * SA + 1;
	const int k = threadIdx.x, q = blockIdx.x;
	if (q == NACC) {
		if (k == 0) {
			double s = 0.0;
			for (int y = 0; y < RED_ROWS; ++y) s += stage[(int64_t)y * FSL + NACC * SA];
			out[NACC * n] = s;
		}
		return;
	}
	double s = 0.0;
	for (int y = 0; y < RED_ROWS; ++y) s += stage[(int64_t)y * FSL + q * SA + k];
	if (k < n) {
		const double cells = q == 0 ? k : (q == 1 ? n - 1 - k : (q == 2 ? 1 : (q == 3 ? n - 1 - k : (q == 4 ? k : 1))));
		out[q * n + k] = s + cells * tiny_total;
	}
}

void launch_bwd_acc(const EstepLaunch &p, hipStream_t st, int which, int first, int n)
{
	if (n <= 0) return;
	// which == 6: the main pass when the bulk items span several tiles (coarse): every tile outside the backward runs, one by one
	const SweepItemA *items = (const SweepItemA *)(which == 1 ? p.d_ritems_b : (which == 3 ? p.d_members_b : (which == 6 ? p.d_singles_b : p.d_items_b))) + first;
	const int mode = which == 1 ? 1 : (which == 2 ? 2 : 0);
	if (p.ckpt)
		hipLaunchKernelGGL(k_bwd_acc_ckpt, dim3((n + 3) / 4), dim3(64), 0, st, p.d_sp, p.d_e, p.d_s, p.d_obs, p.d_chunks, items, n,
		                   mode, p.d_f, p.d_entry, p.d_bentry, p.d_bexit, p.d_Cpart, p.d_touch_f, p.d_touch_b);
	else if (p.ns == 128)
		hipLaunchKernelGGL(k_bwd_acc_struct<8>, dim3((n + 3) / 4), dim3(64), 0, st, p.d_sp, p.d_e, p.d_s, p.d_obs, p.d_chunks, items, n,
		                   mode, p.d_f, p.d_bentry, p.d_bexit, p.d_Cpart, p.d_touch_f, p.d_touch_b);
	else
		hipLaunchKernelGGL(k_bwd_acc_struct<4>, dim3((n + 3) / 4), dim3(64), 0, st, p.d_sp, p.d_e, p.d_s, p.d_obs, p.d_chunks, items, n,
		                   mode, p.d_f, p.d_bentry, p.d_bexit, p.d_Cpart, p.d_touch_f, p.d_touch_b);
	PSMC_DBG("launch_bwd_acc", which, first, n);
}
void launch_reduce_factored(const EstepLaunch &p, hipStream_t st)
{
	if (p.ns == 128) {
		hipLaunchKernelGGL(k_reduce_factored1<128>, dim3(NACC + 1, RED_ROWS), dim3(128), 0, st, p.d_Cpart, p.n_chunks, p.d_LLpart, p.d_stage);
		hipLaunchKernelGGL(k_reduce_factored2<128>, dim3(NACC + 1), dim3(128), 0, st, p.d_stage, p.tiny_total, p.n_states, p.d_stats);
		return;
	}
	hipLaunchKernelGGL(k_reduce_factored1<64>, dim3(NACC + 1, RED_ROWS), dim3(64), 0, st, p.d_Cpart, p.n_chunks, p.d_LLpart, p.d_stage);
	hipLaunchKernelGGL(k_reduce_factored2<64>, dim3(NACC + 1), dim3(64), 0, st, p.d_stage, p.tiny_total, p.n_states, p.d_stats);
}

} // namespace psmc
