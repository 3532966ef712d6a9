// estep_fused.hip -- FAST mode, structured matrices: backward sweep and expected counts in ONE kernel.
//
// The counts kernel of estep_fast.hip reads X and bt back from HBM (1 KB per bin) after the backward
// sweep wrote bt (0.5 KB per bin).  Here the wave that walks its tiles backwards feeds the bt vectors
// straight into the f64 matrix cores: bt never goes to memory, HBM traffic per bin drops from 2.1 KB to
// 1.0 KB (write X, read X), and the separate counts pass disappears.
//
// The wave keeps the layout of the structured sweeps (estep_struct.hip): four tiles, one per 16-lane row,
// four adjacent states per lane.  That register layout already IS an operand layout of
// v_mfma_f64_16x16x4: lane (t, i) supplies A[M = i][K = t] and B[K = t][N = i], so with
//   A_j = w X[4i + j]  (j = 0..3: the lane's j-th state),   B_j' = bt_{p+1}[4i + j']
// the K dimension of the instruction runs over the wave's FOUR TILES (one position each) instead of four
// positions of one tile -- legitimate because the counts of all tiles are summed anyway:
//   C[4M + j][4N + j'] += sum_t  w^(t) X^(t)[4M + j] bt^(t)[4N + j'],     16 instructions per step.
// Per position p of a tile (z = bt_{p+1}, sb_p the lagged scale factor of p % 4 == 0):
//   g_p[k] = X_p[k] bt_p[k] / e[o_p][k],  G_p = sum_k g_p[k],  w = mult sb_p / G_p
//   E[o_p][k] += mult g_p[k] / G_p,   C[k][l] += w X_p[k] z[l],   A = a .* C (k_reduce2)
// A group of four consecutive tiles shares one partial C (Cpart[group]); E partials are per tile.  A tile whose
// X or start vector a repair changed makes its whole group recompute (mode 2), which overwrites the partial.
// The backward repair rounds themselves only move boundary vectors (launch_bwd_struct, which = 5).
#include <hip/hip_runtime.h>
#include "wave_prims.h"
#include "struct_prims.h"
#include <type_traits>
#include "psmc_hip_internal.h"

namespace psmc {

#ifdef PSMC_NO_SB
#define PSMC_SB
#else
#define PSMC_SB __builtin_amdgcn_sched_barrier(0);
#endif

typedef double d4f_t __attribute__((ext_vector_type(4)));
constexpr int NPLF = 4, SF = 64;

__device__ __forceinline__ int64_t readlane_i64f(int64_t v, int lane) {
	const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, lane);
	const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)v >> 32), lane);
	return (int64_t)(((unsigned long long)hi << 32) | lo);
}

// ---- The posterior weight without a normaliser per position (round 2).
//
// Round 1's kernel computed G_p = sum_k g_p[k] at every position to turn X_p (x) bt_{p+1} into a posterior: a 16-lane
// reduction, a reciprocal and 13 multiplications per step (removed in round 3; 6.97 vs 6.2 ms per E-step's back half).
// But with y_p = a bt_{p+1} the normaliser
// I_p = sum_k X_p[k] y_p[k] is not a new number at every position: between two normalising positions it is constant (an
// algebraic identity of the forward and the backward recursion), and across one it changes by a known factor,
// I_{p-1} = I_p sb_p / inv_p (sb_p = 1/sum(bt_{p+1}): bt's own scale factor; inv_p: the forward sweep's, from the d_s table).
// So one pre-step measures I at the tile's top position, rho = mult / I (mult = the segment's multiplicity), and
//   E[o_p][k] += rho X_p[k] y_p[k],    C[k][l] += rho X_p[k] bt_{p+1}[l],    rho <- rho inv_p / sb_p at p % 4 == 0.
// rho drifts by rounding only (a few ulp per normalising position; every tile measures its own).
// bt keeps ITS OWN scaling, as in round 1.  The first version of this kernel let bt follow the forward scale factors
// (khmm.c:228-235: I is then constant over the whole tile) -- but then the exit vector a tile hands to the tile below, and
// to the boundary repairs, is computed from a forward table, and a forward REPAIR may be rewriting that table while it is
// read (by design: the repair flags the tile, which is recomputed at the end).  The 16 lanes of a row could see different
// scale factors, the exit vector came out bent instead of merely rescaled, and the repair of the tile below adopted it:
// 1 E-step in 1 500 off by up to 3e-2 in a stress of tiny tiles (profiles/experiments/dbg_flaky_tiling.py), caught first as a flaky
// test_fast_odd_tilings.  The vectors that travel between tiles must depend on the observations and the parameters only.
// Tried on top of this and removed: issuing the matrix instructions of step p between the vector instructions of step
// p-1 (sched_group_barrier pattern, operands held for a step) -- slower, 6.49 vs 6.22 ms: the dependent chains of the scans
// were never the problem (the compiler already fills them), the instruction count is.
template <bool NORM, bool MASKED>
__device__ __forceinline__ void count4f_step(const StructParN<NPLF> &sc, const double *lds_e, const double *lds_m, int k0, int sym,
                                             const double (&X)[NPLF], double (&x)[NPLF], bool active, double inv, double &rho,
                                             double (&FA)[NPLF], double (&FB)[NPLF], double (&S)[2][NPLF])
{
	double ev[NPLF], y[NPLF];
	ev_load<NPLF>(lds_e + sym * SF, k0, ev);
	const d2v_t mk = *reinterpret_cast<const d2v_t *>(lds_m + 2 * sym); // (1,0) hom, (0,1) het, (0,0) missing
	double rho_next = rho;
	if (NORM) { // bt keeps its OWN scaling (sb_p = 1/sum(bt_{p+1})): the vectors handed from tile to tile must not depend on a forward table
		const double tot = row_sum16((x[0] + x[1]) + (x[2] + x[3]));
		const double sb = rcp_newton(tot);
#pragma unroll
		for (int i = 0; i < NPLF; ++i) ev[i] *= sb;
		rho_next = rho * (inv * tot); // 1/I_{p-1} = (1/I_p) inv_p / sb_p: the weight follows both scale factors
		if (MASKED) rho_next = active ? rho_next : rho;
	}
#pragma unroll
	for (int i = 0; i < NPLF; ++i) y[i] = x[i];
	struct_step<NPLF>(sc, y);
#pragma unroll
	for (int i = 0; i < NPLF; ++i) {
		FA[i] = rho * X[i];
		if (MASKED) FA[i] = active ? FA[i] : 0.0; // an idle row may hold anything
		// (an idle row -- a tile that owns no transition, a padding entry -- may hold anything, NaN included: its start vector was never
		// written.  0 x NaN is NaN, so the product is SELECTED away, not multiplied away; found with PSMC_HIP_POISON=1 in round 4.)
		const double gk = MASKED ? (active ? FA[i] * y[i] : 0.0) : FA[i] * y[i];
		S[0][i] = __builtin_fma(gk, mk.x, S[0][i]);
		S[1][i] = __builtin_fma(gk, mk.y, S[1][i]);
		FB[i] = MASKED ? (active ? x[i] : 0.0) : x[i];
		const double nb = y[i] * ev[i];
		x[i] = MASKED ? (active ? nb : x[i]) : nb;
	}
	rho = rho_next;
}
__device__ __forceinline__ void count4f_mfma(const double (&FA)[NPLF], const double (&FB)[NPLF], d4f_t (&acc)[4][4])
{
#pragma unroll
	for (int j = 0; j < 4; ++j)
#pragma unroll
		for (int j2 = 0; j2 < 4; ++j2) acc[j][j2] = __builtin_amdgcn_mfma_f64_16x16x4f64(FA[j], FB[j2], acc[j][j2], 0, 0, 0);
}
#ifdef PSMC_TRACE_SWEEP
__device__ unsigned long long g_trace_c[4 * 8192];
#endif
__global__ __launch_bounds__(64, 1) void k_bwd_count4f_struct(const double *__restrict__ sp, const double *__restrict__ e,
                                                                const double *__restrict__ invd, const uint8_t *__restrict__ obs,
                                                                const Chunk *__restrict__ chunks, const int *__restrict__ tiles,
                                                                int group0, int mode, const double *__restrict__ f,
                                                                double *__restrict__ bentry, double *__restrict__ bexit,
                                                                double *__restrict__ Cpart,
                                                                double *__restrict__ Epart, const int *__restrict__ touch_f,
                                                                const int *__restrict__ touch_b, const int *__restrict__ fmerge, const double *__restrict__ finv)
{
	__shared__ double lds_e[4 * SF], lds_m[8]; // e rows: hom, het, 1, 1;  count masks per symbol
	const int lane = threadIdx.x, row = lane >> 4, m = lane & 15, k0 = NPLF * m;
	{ const int q = ev_slot<NPLF>(lane); lds_e[q] = e[lane]; lds_e[SF + q] = e[SF + lane]; lds_e[2 * SF + q] = 1.0; lds_e[3 * SF + q] = 1.0; } // ev_load layout
	if (lane < 8) lds_m[lane] = (lane == 0 || lane == 3) ? 1.0 : 0.0;
	__syncthreads();
	const int group = group0 + blockIdx.x;
#ifdef PSMC_TRACE_SWEEP
	// debug build (scripts/sweep_trace.py): wall-clock start / end and shader-clock cycles of every wave of the counts
	struct TraceC { int g; unsigned long long w0, c0; __device__ ~TraceC() { if (threadIdx.x == 0 && g < 8192) { g_trace_c[4 * g] = w0; g_trace_c[4 * g + 1] = __builtin_readcyclecounter() - c0; g_trace_c[4 * g + 2] = wall_clock64(); } } }
		trace_c{group, (unsigned long long)wall_clock64(), (unsigned long long)__builtin_readcyclecounter()};
#endif
	const int entry = tiles[4 * blockIdx.x + row];
	const bool valid = entry >= 0, from_above = valid && mode == 0 && (entry & (1 << 30)) != 0;
	const int tile = valid ? (entry & ~(1 << 30)) : 0;
	if (mode == 2 && !__any(valid && (touch_f[tile] | touch_b[tile]) != 0)) return;
	const Chunk c = chunks[tile];
	const int L = c.L, lo = c.lo, top = min(c.hi, L - 1);
	const bool work = valid && top >= lo; // a tile holding only position L owns no transition
	const double *fo = f + c.off * SF + k0;
	const double *io = invd + c.off;
	StructParN<NPLF> sc; // backward: mS = c, wS = R, mP = qa, wP = P
	loadN<NPLF>(sp + 3 * SF + k0, sc.mS); loadN<NPLF>(sp + SF + k0, sc.wS);
	loadN<NPLF>(sp + 2 * SF + k0, sc.mP); loadN<NPLF>(sp + k0, sc.wP); loadN<NPLF>(sp + 4 * SF + k0, sc.dd);
	double x[NPLF];
	loadN<NPLF>((from_above ? bexit + (int64_t)(tile + 1) * SF : bentry + (int64_t)tile * SF) + k0, x);
	if (from_above) storeN<NPLF>(bentry + (int64_t)tile * SF + k0, x); // what verify compares and a redo starts from
	const int p_min = lo, p_max = max(top, lo);
	const int g_hi = work ? (top - 1) >> 2 : -1, g_lo = work ? (lo - 1) >> 2 : 0;
	const int ng = g_hi - g_lo + 1;
	const int64_t off0 = readlane_i64f(c.off, 0), off1 = readlane_i64f(c.off, 16), off2 = readlane_i64f(c.off, 32), off3 = readlane_i64f(c.off, 48);
	const int gh0 = __builtin_amdgcn_readlane(g_hi, 0), gh1 = __builtin_amdgcn_readlane(g_hi, 16), gh2 = __builtin_amdgcn_readlane(g_hi, 32), gh3 = __builtin_amdgcn_readlane(g_hi, 48);
	const int n0 = __builtin_amdgcn_readlane(ng, 0), n1 = __builtin_amdgcn_readlane(ng, 16), n2 = __builtin_amdgcn_readlane(ng, 32), n3 = __builtin_amdgcn_readlane(ng, 48);
	const int ng_max = max(max(n0, n1), max(n2, n3));
	// Where the forward fix pass stopped rewriting a tile (estep_struct.hip FwdCtl), the rows above and below differ by a recorded factor: the
	// weight takes it when the sweep steps from group g_merge + 1 into group g_merge
	int g_merge = -1;
	double rho_fix = 1.0;
	if (fmerge != nullptr && work) {
		const int bm = fmerge[tile];
		if (bm > 0 && lo + 16 * bm - 1 < p_max) { g_merge = (lo + 16 * bm - 2) >> 2; rho_fix = finv[tile]; }
	}
	double rho; // mult / I of this row's tile
	{ // the pre-step: I = sum_k X_top[k] (a bt_{top+1})[k]
		double y[NPLF], Xt[NPLF];
		loadN<NPLF>(fo + (int64_t)(p_max - 1) * SF, Xt);
#pragma unroll
		for (int i = 0; i < NPLF; ++i) y[i] = x[i];
		struct_step<NPLF>(sc, y);
		const double I = row_sum16((Xt[0] * y[0] + Xt[1] * y[1]) + (Xt[2] * y[2] + Xt[3] * y[3]));
		rho = work ? (double)c.mult * rcp_newton(I) : 0.0;
	}
	d4f_t acc[4][4];
	double S[2][NPLF];
#pragma unroll
	for (int j = 0; j < 4; ++j) {
#pragma unroll
		for (int j2 = 0; j2 < 4; ++j2) acc[j][j2] = (d4f_t){0.0, 0.0, 0.0, 0.0};
		S[0][j] = S[1][j] = 0.0;
	}
	auto load_row = [&](int g, int j, double (&Xq)[NPLF]) { // X of position 4g + j + 1, clamped into the tile
		const int p = min(max(4 * g + j + 1, p_min), p_max);
		loadN<NPLF>(fo + (int64_t)(p - 1) * SF, Xq);
	};
	auto load_inv = [&](int g) { // the forward scale factor of the group's normalising position 4g + 4 (anything for a position outside the tile: masked)
		return io[min(max(4 * g + 4, p_min), p_max) - 1];
	};
	double Xg[4][NPLF];
#pragma unroll
	for (int j = 0; j < 4; ++j) load_row(max(g_hi, 0), j, Xg[j]);
	double inv_cur = load_inv(max(g_hi, 0));
	int gi = 0;
	struct Words { unsigned w0, w1, w2, w3; };
	auto load_words = [&](int gi_) { // the four rows' symbol words of group gi_ as scalar loads (wave-uniform addresses), issued one group ahead
		Words q;
		const uint8_t *p0 = obs + off0 + 4 * (int64_t)max(gh0 - min(gi_, max(n0 - 1, 0)), 0);
		const uint8_t *p1 = obs + off1 + 4 * (int64_t)max(gh1 - min(gi_, max(n1 - 1, 0)), 0);
		const uint8_t *p2 = obs + off2 + 4 * (int64_t)max(gh2 - min(gi_, max(n2 - 1, 0)), 0);
		const uint8_t *p3 = obs + off3 + 4 * (int64_t)max(gh3 - min(gi_, max(n3 - 1, 0)), 0);
		asm volatile("s_load_dword %0, %4, 0x0\n\ts_load_dword %1, %5, 0x0\n\ts_load_dword %2, %6, 0x0\n\ts_load_dword %3, %7, 0x0"
		             : "=&s"(q.w0), "=&s"(q.w1), "=&s"(q.w2), "=&s"(q.w3) : "s"(p0), "s"(p1), "s"(p2), "s"(p3) : "memory");
		return q;
	};
	auto select_word = [&](Words q) {
		asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(q.w0), "+s"(q.w1), "+s"(q.w2), "+s"(q.w3));
		return row == 0 ? q.w0 : (row == 1 ? q.w1 : (row == 2 ? q.w2 : q.w3));
	};
	unsigned w_cur = select_word(load_words(0));
	auto all_full = [&](int gi_) {
		const int g = max(g_hi - gi_, g_lo);
		return __all(gi_ < ng && 4 * g + 1 >= lo && 4 * g + 4 <= top) != 0;
	};
	auto do_group = [&](auto masked_tag) {
		constexpr bool MASKED = decltype(masked_tag)::value;
		const unsigned w = w_cur;
		const Words wn = load_words(gi + 1);
		__builtin_amdgcn_sched_barrier(0);
		const int g = max(g_hi - gi, g_lo); // rows that are done idle on their last group
		const bool in_tile = gi < ng;
		const int s3 = (int)((w >> 24) & 3u), s2 = (int)((w >> 16) & 3u), s1 = (int)((w >> 8) & 3u), s0 = (int)(w & 3u);
		const int pb = 4 * g + 1;
		const double inv = inv_cur;
		rho = (in_tile && g == g_merge) ? rho * rho_fix : rho;
		double FAn[NPLF], FBn[NPLF];
#define PSMC_C4F(NORM, J, SYM)                                                                                                   \
		count4f_step<NORM, MASKED>(sc, lds_e, lds_m, k0, SYM, Xg[J], x, in_tile && pb + J <= top && pb + J >= lo, inv, rho, FAn, FBn, S); \
		count4f_mfma(FAn, FBn, acc);                                                                                        \
		load_row(g - 1, J, Xg[J]);                                                                                          \
		if (J == 3) inv_cur = load_inv(g - 1);                                                                              \
		__builtin_amdgcn_sched_barrier(0);
		PSMC_C4F(true, 3, s3) PSMC_C4F(false, 2, s2) PSMC_C4F(false, 1, s1) PSMC_C4F(false, 0, s0)
#undef PSMC_C4F
		if (in_tile && g == g_lo) storeN<NPLF>(bexit + (int64_t)tile * SF + k0, x); // x = bt_lo: the group holding lo is the row's last
		w_cur = select_word(wn);
	};
	for (; gi < ng_max && !all_full(gi); ++gi) do_group(std::true_type{});  // a top that is not a multiple of 4
	for (; gi < ng_max && all_full(gi); ++gi) do_group(std::false_type{});
	for (; gi < ng_max; ++gi) do_group(std::true_type{});                    // rows of unequal length, a lo that is not 1 mod 4
	// the matrix-core results are not interlocked against plain reads: let the last instructions drain
	asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15"
	             : "+a"(acc[0][0]), "+a"(acc[0][1]), "+a"(acc[0][2]), "+a"(acc[0][3]), "+a"(acc[1][0]), "+a"(acc[1][1]), "+a"(acc[1][2]),
	               "+a"(acc[1][3]), "+a"(acc[2][0]), "+a"(acc[2][1]), "+a"(acc[2][2]), "+a"(acc[2][3]), "+a"(acc[3][0]), "+a"(acc[3][1]),
	               "+a"(acc[3][2]), "+a"(acc[3][3]));
	double *out = Cpart + (int64_t)group * (SF * SF);
#pragma unroll
	for (int j = 0; j < 4; ++j)
#pragma unroll
		for (int r = 0; r < 4; ++r) {
			const double v[NPLF] = {acc[j][0][r], acc[j][1][r], acc[j][2][r], acc[j][3][r]};
			storeN<NPLF>(out + (4 * (row + 4 * r) + j) * SF + k0, v);
		}
	if (valid) {
		double *os = Epart + (int64_t)tile * (3 * SF) + k0;
		const double zero[NPLF] = {0.0, 0.0, 0.0, 0.0};
		storeN<NPLF>(os, S[0]); storeN<NPLF>(os + SF, S[1]); storeN<NPLF>(os + 2 * SF, zero); // missing symbols are not counted (khmm.c:355)
	}
}

constexpr int NPL8 = 8, S8 = 128;
// ---- 128 states (`-p "64*2"`), ONE sweep per tile (round 4): k_bwd_count8x_struct, "fuse128" = 2.
// Round 3's kernel (k_bwd_count8_struct, "fuse128" = 1; removed in round 6) let each of a group's four waves redo the whole eight-states-per-lane sweep of the SAME four tiles to
// own a quarter of C: three of the four sweeps are redundant (9.4 vector instructions per matrix instruction, 0.45 of the FP64 peak
// on the algorithmic flops).  Here a work-group holds SIXTEEN tiles, four per wave.  Every wave sweeps only its own four tiles and
// publishes, per position, both operands of the matrix instructions -- rho.X (8 per lane) and bt (8 per lane) -- to LDS; after one
// barrier every wave runs the K dimension over all sixteen tiles for ITS quarter of C (rows j = 2w, 2w+1 of every block):
// 4 producer waves x 16 matrix instructions = 64 per step and wave instead of 16, on one sweep instead of four.  The waves sit on
// four different SIMDs, so the pipe rule of DESIGN.md section 3 (nothing overlaps a matrix instruction on the SAME SIMD) does not
// stand in the way.  LDS: [parity][producer wave][pair p][lane] as 16-byte cells -- a lane writes its 8 + 8 values with eight
// conflict-free ds_write_b128 and reads a producer's operands with five ds_read_b128 (its A pair, all four B pairs); two parities,
// one barrier per position.  X is read from HBM once, by the tile's owner (whole rows, one position ahead).
constexpr int G8X = 16; // tiles per work-group
// the structured step with its five constant vectors (mS | wS | mP | wP | dd, S8 each) read from LDS just in time: 80 registers the
// kernel cannot spare next to 128 accumulators, two X rows, sixteen emission sums and the operands in flight
// LDS layout of a 128-vector for the eight-states-per-lane kernels: [pair p = 0..3][lane m = 0..15] cells of 16 bytes holding states
// 8m + 2p, 8m + 2p + 1 -- the sixteen lanes of a row read consecutive cells (a ds_read_b128 of the natural layout, 64 bytes apart per
// lane, is a four-way bank conflict); the four rows read the same cells (a broadcast)
__device__ __forceinline__ void load8p(const d2v_t *v, int m, double (&out)[NPL8]) {
#pragma unroll
	for (int p = 0; p < 4; ++p) { const d2v_t t = v[16 * p + m]; out[2 * p] = t.x; out[2 * p + 1] = t.y; }
}
__device__ __forceinline__ void fill8p(d2v_t *v, int k, double val) { reinterpret_cast<double *>(v + 16 * ((k & 7) >> 1) + (k >> 3))[k & 1] = val; } // state k = 8m + 2p + h
__device__ __forceinline__ void struct_step8_lds(const d2v_t *lds_sc, int m, double (&x)[NPL8])
{
	double cS[NPL8], cP[NPL8], su[NPL8], pv[NPL8];
	load8p(lds_sc, m, cS); load8p(lds_sc + 2 * 64, m, cP);
	su[NPL8 - 1] = x[NPL8 - 1] * cS[NPL8 - 1];
#pragma unroll
	for (int i = NPL8 - 2; i >= 0; --i) su[i] = __builtin_fma(x[i], cS[i], su[i + 1]);
	pv[0] = x[0] * cP[0];
#pragma unroll
	for (int i = 1; i < NPL8; ++i) pv[i] = __builtin_fma(x[i], cP[i], pv[i - 1]);
	const double ES = row_excl_suffix(su[0]), EP = row_excl_prefix(pv[NPL8 - 1]);
	double wS[NPL8], wP[NPL8], dd[NPL8];
	load8p(lds_sc + 64, m, wS); load8p(lds_sc + 3 * 64, m, wP); load8p(lds_sc + 4 * 64, m, dd);
#pragma unroll
	for (int i = 0; i < NPL8; ++i) { // the same expression tree as struct_step (struct_prims.h): the two kernels round alike
		const double t = __builtin_fma(wS[i], su[i], __builtin_fma(wP[i], pv[i], dd[i] * x[i]));
		x[i] = __builtin_fma(wS[i], ES, __builtin_fma(wP[i], EP, t));
	}
}
template <bool NORM, bool MASKED>
__device__ __forceinline__ void count8x_step(const d2v_t *lds_sc, const d2v_t *lds_e, const double *lds_m, int m, int sym,
                                             const double (&X)[NPL8], double (&x)[NPL8], bool active, double inv, double &rho,
                                             double (&FA)[NPL8], double (&FB)[NPL8], double (&S)[2][NPL8])
{
	double ev[NPL8], y[NPL8];
	load8p(lds_e + sym * 64, m, ev);
	const d2v_t mk = *reinterpret_cast<const d2v_t *>(lds_m + 2 * sym); // (1,0) hom, (0,1) het, (0,0) missing
	double rho_next = rho;
	if (NORM) { // bt keeps its own scaling, the weight follows both scale factors (see count4f_step)
		double t = 0.0;
#pragma unroll
		for (int i = 0; i < NPL8; ++i) t += x[i];
		const double tot = row_sum16(t);
		const double sb = rcp_newton(tot);
#pragma unroll
		for (int i = 0; i < NPL8; ++i) ev[i] *= sb;
		rho_next = rho * (inv * tot);
		if (MASKED) rho_next = active ? rho_next : rho;
	}
#pragma unroll
	for (int i = 0; i < NPL8; ++i) y[i] = x[i];
	struct_step8_lds(lds_sc, m, y);
#pragma unroll
	for (int i = 0; i < NPL8; ++i) {
		FA[i] = rho * X[i];
		if (MASKED) FA[i] = active ? FA[i] : 0.0; // an idle row may hold anything
		const double gk = MASKED ? (active ? FA[i] * y[i] : 0.0) : FA[i] * y[i]; // E[o_p][k] += rho X_p[k] y_p[k]; selected, not multiplied away (count4f_step)
		S[0][i] = __builtin_fma(gk, mk.x, S[0][i]);
		S[1][i] = __builtin_fma(gk, mk.y, S[1][i]);
		FB[i] = MASKED ? (active ? x[i] : 0.0) : x[i];
		const double nb = y[i] * ev[i];
		x[i] = MASKED ? (active ? nb : x[i]) : nb;
	}
	rho = rho_next;
}

__global__ __launch_bounds__(256, 1) void k_bwd_count8x_struct(const double *__restrict__ sp, const double *__restrict__ e,
                                                                 const double *__restrict__ invd, const uint8_t *__restrict__ obs,
                                                                 const Chunk *__restrict__ chunks, const int *__restrict__ tiles,
                                                                 int group0, int mode, const double *__restrict__ f,
                                                                 double *__restrict__ bentry, double *__restrict__ bexit,
                                                                 double *__restrict__ Cpart,
                                                                 double *__restrict__ Epart, const int *__restrict__ touch_f,
                                                                 const int *__restrict__ touch_b)
{
	__shared__ d2v_t lds_e[4 * 64];   // emission rows hom | het | 1 | 1 in the load8p layout
	__shared__ double lds_m[8];
	__shared__ d2v_t xa[2][4][4][64], xb[2][4][4][64]; // operands A = rho.X, B = bt: [parity][producer wave][state pair][lane], 32 KB each
	__shared__ int lds_ng[4];
	__shared__ d2v_t lds_sc[5 * 64]; // backward roles: mS = c | wS = R | mP = qa | wP = P | dd (sp = P | R | qa | c | dd), load8p layout
	const int tid = threadIdx.x, lane = tid & 63, row = lane >> 4, m = lane & 15, k0 = NPL8 * m;
	const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
	if (tid < S8) {
		fill8p(lds_e, tid, e[tid]); fill8p(lds_e + 64, tid, e[S8 + tid]); fill8p(lds_e + 2 * 64, tid, 1.0); fill8p(lds_e + 3 * 64, tid, 1.0);
		fill8p(lds_sc, tid, sp[3 * S8 + tid]); fill8p(lds_sc + 64, tid, sp[S8 + tid]); fill8p(lds_sc + 2 * 64, tid, sp[2 * S8 + tid]);
		fill8p(lds_sc + 3 * 64, tid, sp[tid]); fill8p(lds_sc + 4 * 64, tid, sp[4 * S8 + tid]);
	}
	if (tid < 8) lds_m[tid] = (tid == 0 || tid == 3) ? 1.0 : 0.0;
	__syncthreads();
	const int group = group0 + blockIdx.x;
	const int entry = tiles[G8X * blockIdx.x + 4 * w + row];
	const bool valid = entry >= 0, from_above = valid && mode == 0 && (entry & (1 << 30)) != 0;
	const int tile = valid ? (entry & ~(1 << 30)) : 0;
	// redo pass: the whole group is recomputed when a repair touched any of its sixteen tiles (block-uniform)
	if (mode == 2 && !__syncthreads_or(valid && (touch_f[tile] | touch_b[tile]) != 0)) return;
	const Chunk c = chunks[tile];
	const int L = c.L, lo = c.lo, top = min(c.hi, L - 1);
	const bool work = valid && top >= lo;
	const double *fo = f + c.off * S8 + k0;
	const double *io = invd + c.off;
	double x[NPL8];
	loadN<NPL8>((from_above ? bexit + (int64_t)(tile + 1) * S8 : bentry + (int64_t)tile * S8) + k0, x);
	if (from_above) storeN<NPL8>(bentry + (int64_t)tile * S8 + k0, x); // what verify compares and a redo starts from (this wave owns the tile)
	const int p_min = lo, p_max = max(top, lo);
	double rho;
	{
		double y[NPL8], Xt[NPL8];
		loadN<NPL8>(fo + (int64_t)(p_max - 1) * S8, Xt);
#pragma unroll
		for (int i = 0; i < NPL8; ++i) y[i] = x[i];
		struct_step8_lds(lds_sc, m, y);
		double t = 0.0;
#pragma unroll
		for (int i = 0; i < NPL8; ++i) t = __builtin_fma(Xt[i], y[i], t);
		rho = work ? (double)c.mult * rcp_newton(row_sum16(t)) : 0.0;
	}
	d4f_t acc[2][NPL8];
	double S[2][NPL8];
#pragma unroll
	for (int j2 = 0; j2 < NPL8; ++j2) {
		acc[0][j2] = (d4f_t){0.0, 0.0, 0.0, 0.0}; acc[1][j2] = (d4f_t){0.0, 0.0, 0.0, 0.0};
		S[0][j2] = S[1][j2] = 0.0;
	}
	const int g_hi = work ? (top - 1) >> 2 : -1, g_lo = work ? (lo - 1) >> 2 : 0;
	const int ng = g_hi - g_lo + 1;
	const int64_t off0 = readlane_i64f(c.off, 0), off1 = readlane_i64f(c.off, 16), off2 = readlane_i64f(c.off, 32), off3 = readlane_i64f(c.off, 48);
	const int gh0 = __builtin_amdgcn_readlane(g_hi, 0), gh1 = __builtin_amdgcn_readlane(g_hi, 16), gh2 = __builtin_amdgcn_readlane(g_hi, 32), gh3 = __builtin_amdgcn_readlane(g_hi, 48);
	const int n0 = __builtin_amdgcn_readlane(ng, 0), n1 = __builtin_amdgcn_readlane(ng, 16), n2 = __builtin_amdgcn_readlane(ng, 32), n3 = __builtin_amdgcn_readlane(ng, 48);
	if (lane == 0) lds_ng[w] = max(max(n0, n1), max(n2, n3));
	__syncthreads();
	const int ng_max = max(max(lds_ng[0], lds_ng[1]), max(lds_ng[2], lds_ng[3])); // every wave takes the same number of steps: one barrier each
	auto row_pos = [&](int g, int j) { return min(max(4 * g + j + 1, p_min), p_max); };
	auto load_inv = [&](int g) { return io[min(max(4 * g + 4, p_min), p_max) - 1]; };
	double Xa[NPL8], Xb[NPL8]; // X rows, one position ahead, in two alternating buffers
	loadN<NPL8>(fo + (int64_t)(row_pos(max(g_hi, 0), 3) - 1) * S8, Xa);
	double inv_cur = load_inv(max(g_hi, 0));
	auto all_full = [&](int gi_) {
		const int g = max(g_hi - gi_, g_lo);
		return __all(gi_ < ng && 4 * g + 1 >= lo && 4 * g + 4 <= top) != 0;
	};
	int gi = 0, par = 0;
	auto exchange = [&](const double (&FA)[NPL8], const double (&FB)[NPL8]) {
#pragma unroll
		for (int p = 0; p < 4; ++p) {
			d2v_t u, v; u.x = FA[2 * p]; u.y = FA[2 * p + 1]; v.x = FB[2 * p]; v.y = FB[2 * p + 1];
			xa[par][w][p][lane] = u; xb[par][w][p][lane] = v;
		}
		__syncthreads();
		// K over the four tiles of producer wave g, the next producer's operands in flight while this one's sixteen matrix
		// instructions issue (1024 cycles: the LDS latency disappears behind them; left to itself hipcc fetched each operand right
		// before its use, 26 s_waitcnt per step).  The sched_barriers keep that order.
		d2v_t oa[2], ob[2][4];
		oa[0] = xa[par][0][w][lane];
#pragma unroll
		for (int p = 0; p < 4; ++p) ob[0][p] = xb[par][0][p][lane];
#pragma unroll
		for (int g = 0; g < 4; ++g) {
			if (g < 3) {
				oa[(g + 1) & 1] = xa[par][g + 1][w][lane];
#pragma unroll
				for (int p = 0; p < 4; ++p) ob[(g + 1) & 1][p] = xb[par][g + 1][p][lane];
			}
			__builtin_amdgcn_sched_barrier(0);
			const d2v_t a = oa[g & 1];
			const double B[NPL8] = {ob[g & 1][0].x, ob[g & 1][0].y, ob[g & 1][1].x, ob[g & 1][1].y, ob[g & 1][2].x, ob[g & 1][2].y, ob[g & 1][3].x, ob[g & 1][3].y};
#pragma unroll
			for (int j2 = 0; j2 < NPL8; ++j2) {
				acc[0][j2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x, B[j2], acc[0][j2], 0, 0, 0);
				acc[1][j2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.y, B[j2], acc[1][j2], 0, 0, 0);
			}
			__builtin_amdgcn_sched_barrier(0);
		}
		par ^= 1;
	};
	auto do_group = [&](auto masked_tag) {
		constexpr bool MASKED = decltype(masked_tag)::value;
		const unsigned w0 = *reinterpret_cast<const unsigned *>(obs + off0 + 4 * (int64_t)max(gh0 - min(gi, max(n0 - 1, 0)), 0));
		const unsigned w1 = *reinterpret_cast<const unsigned *>(obs + off1 + 4 * (int64_t)max(gh1 - min(gi, max(n1 - 1, 0)), 0));
		const unsigned w2 = *reinterpret_cast<const unsigned *>(obs + off2 + 4 * (int64_t)max(gh2 - min(gi, max(n2 - 1, 0)), 0));
		const unsigned w3 = *reinterpret_cast<const unsigned *>(obs + off3 + 4 * (int64_t)max(gh3 - min(gi, max(n3 - 1, 0)), 0));
		const unsigned ww = row == 0 ? w0 : (row == 1 ? w1 : (row == 2 ? w2 : w3));
		const int g = max(g_hi - gi, g_lo);
		const bool in_tile = gi < ng;
		const int s3 = (int)((ww >> 24) & 3u), s2 = (int)((ww >> 16) & 3u), s1 = (int)((ww >> 8) & 3u), s0 = (int)(ww & 3u);
		const int pb = 4 * g + 1;
		const double inv = inv_cur;
		double FA[NPL8], FB[NPL8];
		// position J of the group: its X row is in XC; the next position's row (J - 1, or J = 3 of the group below) is fetched into XN
#define PSMC_C8X(NORM, J, SYM, XC, XN, GN, JN)                                                                                   \
		loadN<NPL8>(fo + (int64_t)(row_pos(GN, JN) - 1) * S8, XN);                                                          \
		count8x_step<NORM, MASKED>(lds_sc, lds_e, lds_m, m, SYM, XC, x, in_tile && pb + J <= top && pb + J >= lo, inv, rho, FA, FB, S); \
		if (J == 3) inv_cur = load_inv(g - 1);                                                                               \
		exchange(FA, FB);
		PSMC_C8X(true, 3, s3, Xa, Xb, g, 2) PSMC_C8X(false, 2, s2, Xb, Xa, g, 1) PSMC_C8X(false, 1, s1, Xa, Xb, g, 0) PSMC_C8X(false, 0, s0, Xb, Xa, g - 1, 3)
#undef PSMC_C8X
		if (in_tile && g == g_lo) storeN<NPL8>(bexit + (int64_t)tile * S8 + k0, x);
	};
	// three single-path loops, as in the kernels above (with both paths in one body hipcc shuffles the accumulators between VGPRs and
	// AGPRs every iteration).  The choice is per wave; a wave passes four barriers per group whichever loop it is in, and ng_max
	// groups in all
	for (; gi < ng_max && !all_full(gi); ++gi) do_group(std::true_type{});
	for (; gi < ng_max && all_full(gi); ++gi) do_group(std::false_type{});
	for (; gi < ng_max; ++gi) do_group(std::true_type{});
	// the matrix-core results are not interlocked against plain reads: let the last instructions drain
	asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15"
	             : "+a"(acc[0][0]), "+a"(acc[0][1]), "+a"(acc[0][2]), "+a"(acc[0][3]), "+a"(acc[0][4]), "+a"(acc[0][5]), "+a"(acc[0][6]),
	               "+a"(acc[0][7]), "+a"(acc[1][0]), "+a"(acc[1][1]), "+a"(acc[1][2]), "+a"(acc[1][3]), "+a"(acc[1][4]), "+a"(acc[1][5]),
	               "+a"(acc[1][6]), "+a"(acc[1][7]));
	// C[8 M + j][8 N + j2], j = 2 w + jj, M = row + 4 r, N = m: eight adjacent columns per lane
	double *out = Cpart + (int64_t)group * (S8 * S8);
#pragma unroll
	for (int jj = 0; jj < 2; ++jj)
#pragma unroll
		for (int r = 0; r < 4; ++r) {
			double v[NPL8];
#pragma unroll
			for (int j2 = 0; j2 < NPL8; ++j2) v[j2] = acc[jj][j2][r];
			storeN<NPL8>(out + (int64_t)(8 * (row + 4 * r) + 2 * w + jj) * S8 + k0, v);
		}
	if (valid) {
		double *os = Epart + (int64_t)tile * (3 * S8) + k0;
		const double zero[NPL8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
		storeN<NPL8>(os, S[0]); storeN<NPL8>(os + S8, S[1]); storeN<NPL8>(os + 2 * S8, zero); // missing symbols are not counted (khmm.c:355)
	}
}

// list 0 / 1: tile list A / B of the plan (api_fast.hip build_items);  redo: only the groups a repair touched
void launch_bwd_count(const EstepLaunch &p, hipStream_t st, int list, bool redo, bool all_from_bentry)
{
	const int G = p.count_group; // tiles per work-group / per C partial: 4, or 16 (k_bwd_count8x_struct)
	const int ga = (p.n_list_a + G - 1) / G, gb = (p.n_list_b + G - 1) / G;
	const int n_groups = list == 0 ? ga : gb;
	if (n_groups <= 0) return;
	const int *tl = p.d_ftiles + (list == 0 ? 0 : G * ga);
	const int g0 = list == 0 ? 0 : ga, md = all_from_bentry ? 3 : (redo ? 2 : 0); // 3 (diagnostic): every group, every tile from its bentry
	if (p.ns == 128) {
		hipLaunchKernelGGL(k_bwd_count8x_struct, dim3(n_groups), dim3(256), 0, st, p.d_sp, p.d_e, p.d_s, p.d_obs, p.d_chunks, tl, g0, md,
		                   p.d_f, p.d_bentry, p.d_bexit, p.d_Cpart, p.d_Epart, p.d_touch_f, p.d_touch_b);
		PSMC_DBG("launch_bwd_count (128 states)", list, redo, n_groups);
		return;
	}
	hipLaunchKernelGGL(k_bwd_count4f_struct, dim3(n_groups), dim3(64), 0, st, p.d_sp, p.d_e, p.d_s, p.d_obs, p.d_chunks, tl, g0, md,
	                   p.d_f, p.d_bentry, p.d_bexit, p.d_Cpart, p.d_Epart, p.d_touch_f, p.d_touch_b, p.merge ? p.d_fmerge : nullptr, p.d_finv);
	PSMC_DBG("launch_bwd_count", list, redo, n_groups);
}

} // namespace psmc

#ifdef PSMC_TRACE_SWEEP
extern "C" int psmc_hip_debug_trace_counts(unsigned long long *out, int n) // debug build only: 4 * n stamps of k_bwd_count4f_struct (by group)
{
	if (n > 8192) n = 8192;
	return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(psmc::g_trace_c), sizeof(unsigned long long) * 4 * n);
}
#endif
