// estep_struct.hip -- FAST-mode sweeps for transition matrices with the PSMC structure.
//
// psmc_update_hmm (lh3/psmc core.c:112-122) builds a[k][l] from two rank-1 triangles:
//   l < k:  a[k][l] = P_k * qa_l        (tmp * ak1/cpik  times  q_aux[l])
//   l > k:  a[k][l] = R_k * c_l         (tmp * q_aux[k]/cpik  times  alpha_l - alpha_{l+1})
// so a matrix-vector product is two scans instead of N^2 multiply-adds:
//   forward   (a^T x)_j = qa_j * SUF_j(x.P) + c_j * PRE_j(x.R) + dd_j * x_j
//   backward  (a z)_k   = R_k * SUF_k(z.c) + P_k * PRE_k(z.qa) + dd_k * z_k
// with SUF / PRE the INCLUSIVE suffix / prefix sums over the states and
// dd = diag(a) - P.qa - R.c >= 0 (checked on the host), i.e. only additions of
// non-negative terms: no cancellation.  The host factors `a` numerically
// (api.hip factor_structure) and falls back to the dense kernels of estep_fast.hip
// when the matrix does not have this form (e.g. after psmc_cap_matrix, aux.c:115-127).
//
// Mapping: a step costs O(N), so a wavefront carries FOUR tiles, one per 16-lane DPP
// row, four adjacent states per lane (k = 4*(lane&15) + i).  The scans are three local
// adds plus a 4-level row_shr / row_shl DPP scan that never leaves the row, so rows are
// independent "mini waves": each has its own tile, position, symbols and predicates.
// Everything else (speculate / verify / repair protocol, lagged sparse normalisation,
// table layout X[g*64+k], bt[g*64+k], inv_d[g], sb[g]) is the one of estep_fast.hip, so
// the verify, LL, expect and reduce kernels are shared.
#include <hip/hip_runtime.h>
#include "wave_prims.h"
#include "struct_prims.h"
#include "psmc_hip_internal.h"

namespace psmc {

#ifdef PSMC_TRACE_SWEEP
// Debug build only (make EXTRA=-DPSMC_TRACE_SWEEP; scripts/sweep_trace.py): wall-clock stamps (100 MHz) of every wave of the bulk
// forward sweep (f), the backward warm-up pass (b) and the walks (w) -- [0] start, [1] first stored block (end of the warm-up),
// [2] end, [3] HW_ID | XCC_ID << 32 | 16-step blocks of the longest row << 40 -- read by psmc_hip_debug_trace.
__device__ unsigned long long g_trace_f[4 * 16384], g_trace_b[4 * 16384], g_trace_w[4 * 16384];
__device__ __forceinline__ void trace_put(unsigned long long *t, int block, int slot, unsigned long long v) {
	if ((threadIdx.x & 63) == 0 && block < 16384) t[4 * block + slot] = v;
}
__device__ __forceinline__ unsigned long long trace_hw() {
	const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);
	return (unsigned long long)hw | ((unsigned long long)(xcc & 0xf) << 32);
}
#define PSMC_TRACE(t, block, slot, v) trace_put(t, block, slot, v)
#else
#define PSMC_TRACE(t, block, slot, v) ((void)0)
#endif

// symbol i (0..15) of a 16-byte block, clamped to 0..3 (row 3 of the LDS emission table is 1.0 like row 2)
template <int J> __device__ __forceinline__ int sym_of(unsigned w) { return (int)((w >> (8 * J)) & 3u); }
// word g (0..3) of the block's 16 symbols
__device__ __forceinline__ unsigned sym_word(const uint4 sv, int g) { return g == 0 ? sv.x : (g == 1 ? sv.y : (g == 2 ? sv.z : sv.w)); }
// wave-uniform copy of a 64-bit lane value
__device__ __forceinline__ int64_t readlane_i64(int64_t v, int lane) {
	const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, lane);
	const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)v >> 32), lane);
	return (int64_t)(((unsigned long long)hi << 32) | lo);
}
// The 16 symbols of block `bb` of each row's segment, selected per lane.  The four loads have
// wave-uniform addresses, so they are SCALAR loads (SMEM, lgkmcnt): a vector load inside the sweep
// would make every block wait for the stores of the previous one (on gfx9 vmcnt is ONE in-order
// counter for loads and stores), which is what bounds a latency-critical repair wave.
__device__ __forceinline__ uint4 row_symbols(const uint8_t *__restrict__ obs, const int64_t (&roff)[4], const int (&bb)[4],
                                             int row)
{
	const uint4 s0 = *reinterpret_cast<const uint4 *>(obs + roff[0] + ((int64_t)bb[0] << 4));
	const uint4 s1 = *reinterpret_cast<const uint4 *>(obs + roff[1] + ((int64_t)bb[1] << 4));
	const uint4 s2 = *reinterpret_cast<const uint4 *>(obs + roff[2] + ((int64_t)bb[2] << 4));
	const uint4 s3 = *reinterpret_cast<const uint4 *>(obs + roff[3] + ((int64_t)bb[3] << 4));
	uint4 v;
	v.x = row == 0 ? s0.x : (row == 1 ? s1.x : (row == 2 ? s2.x : s3.x));
	v.y = row == 0 ? s0.y : (row == 1 ? s1.y : (row == 2 ? s2.y : s3.y));
	v.z = row == 0 ? s0.z : (row == 1 ? s1.z : (row == 2 ? s2.z : s3.z));
	v.w = row == 0 ? s0.w : (row == 1 ? s1.w : (row == 2 ? s2.w : s3.w));
	return v;
}
// Tile geometry of a sweep wave: LPT lanes per tile (16: four tiles per wave, a tile is a DPP row; 8: eight tiles per wave, 8 lanes x 8
// states -- 64 states only; see struct_prims.h), NPL adjacent states per lane.
template <int NPL, int LPT> __device__ __forceinline__ void tile_step(const StructParN<NPL> &c, double (&x)[NPL], const Half8Masks &hm) {
	if constexpr (LPT == 8) struct_step_h8(c, x, hm); else struct_step<NPL>(c, x);
}
// eight rows (8 lanes per tile)
__device__ __forceinline__ uint4 row_symbols(const uint8_t *__restrict__ obs, const int64_t (&roff)[8], const int (&bb)[8], int row)
{
	uint4 v = *reinterpret_cast<const uint4 *>(obs + roff[0] + ((int64_t)bb[0] << 4));
#pragma unroll
	for (int r = 1; r < 8; ++r) {
		const uint4 s = *reinterpret_cast<const uint4 *>(obs + roff[r] + ((int64_t)bb[r] << 4));
		v.x = row == r ? s.x : v.x; v.y = row == r ? s.y : v.y; v.z = row == r ? s.z : v.z; v.w = row == r ? s.w : v.w;
	}
	return v;
}
template <int NPL, int LPT = 16> __device__ __forceinline__ void load_struct_par(const double *__restrict__ sp, int k0, bool fwd, StructParN<NPL> &c) {
	constexpr int S = LPT * NPL; // sp = P | R | qa | c | dd, S each
	loadN<NPL>(sp + (fwd ? 0 : 3 * S) + k0, c.mS);  // forward: P,  backward: c
	loadN<NPL>(sp + (fwd ? 2 * S : S) + k0, c.wS);  // forward: qa, backward: R
	loadN<NPL>(sp + (fwd ? S : 2 * S) + k0, c.mP);  // forward: R,  backward: qa
	loadN<NPL>(sp + (fwd ? 3 * S : 0) + k0, c.wP);  // forward: c,  backward: P
	loadN<NPL>(sp + 4 * S + k0, c.dd);
}
template <int NPL> __device__ __forceinline__ double lane_sum(const double (&x)[NPL]) {
	double t = (x[0] + x[1]) + (x[2] + x[3]);
	if constexpr (NPL == 8) t = t + ((x[4] + x[5]) + (x[6] + x[7]));
	return t;
}
// sum over the states of the lane's tile, identical in all of its lanes
template <int NPL, int LPT = 16> __device__ __forceinline__ double tile_sum(const double (&x)[NPL]) {
	if constexpr (LPT == 8) return half8_sum(lane_sum<NPL>(x)); else return row_sum16(lane_sum<NPL>(x));
}
// e[0] | e[1] | 1 | 1 (rows of S) for the per-symbol emission fetch
template <int NPL, int LPT = 16> __device__ __forceinline__ void fill_lds_e(double *lds_e, const double *__restrict__ e, int lane) {
	constexpr int S = NPL * LPT;
#pragma unroll
	for (int i = lane; i < S; i += 64) { const int q = ev_slot<NPL, LPT>(i); lds_e[q] = e[i]; lds_e[S + q] = e[S + i]; lds_e[2 * S + q] = 1.0; lds_e[3 * S + q] = 1.0; } // ev_load layout (struct_prims.h)
}

// ---- Dispatch order across streams (round 4).
// Phase 1 runs three kinds of waves side by side, on streams of their own: a few hundred WALKS (one latency-critical wave each,
// priority 3), the BULK grid, and the transfer-matrix blocks (two waves x 176 registers each, several per SIMD).  The dispatcher
// places the waves of one grid on distinct SIMDs but serves the queues in whatever order their first packets become ready; when the
// matrices and the bulk were dispatched before the walks, the walks found no free slot: they started up to 0.6 ms late and up to
// five on one SIMD, and a 7.5 M-bin share took 5.4 instead of 4.3 ms in two E-steps out of three (scripts/sweep_trace.py with
// TRACE_STEPS; profiles/r04_walk_placement_trace.txt).  Kernels of different streams cannot be ordered by START with events (an event
// waits for completion), so every wave group announces its start in a counter and a one-wave gate kernel ahead of the next kind
// waits for it: walks -> bulk -> matrices.  The gate gives up after ~200 us (streams that share a hardware queue would otherwise wait
// for a kernel queued behind them).
__device__ __forceinline__ void announce_start(int *ctr) { if (ctr && threadIdx.x == 0) atomicAdd(ctr, 1); }
__global__ __launch_bounds__(64) void k_gate(const int *__restrict__ ctr, int want)
{
	const unsigned long long t0 = wall_clock64();
	while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want && wall_clock64() - t0 < 20000ull) __builtin_amdgcn_s_sleep(8);
}
void launch_gate(hipStream_t st, const int *ctr, int want)
{
	if (want > 0) hipLaunchKernelGGL(k_gate, dim3(1), dim3(64), 0, st, ctr, want);
}

// A sweep ITEM is a run of `count` consecutive tiles of one segment that one row walks through in
// sequence (count == 1 unless the host glued tiles: see api_fast.hip learn_groups -- where the chain forgets
// slowly a speculative start is wrong and the repair rounds would walk the region tile by tile anyway;
// gluing lets one row do that walk while the rest of the sweep is still running).  All tiles of an item
// but the last are T positions long.
struct SweepItem { int first, count; };
constexpr int SWEEP_WALK = 1;       // leave only the boundary vectors (entry / bentry / bexit) of the tiles walked through
constexpr int SWEEP_TOP_ONLY = 4;   // backward: stop once the top tile's start vector (bentry) is stored: the warm-up alone
constexpr int SWEEP_NO_TOUCH = 8;   // REPAIR kernels: do not flag the tile for the redo pass of the counts
constexpr int SWEEP_FROM_ENTRY = 2; // REPAIR kernels: start from the boundary vector a walk left instead of the neighbour's table row
constexpr int SWEEP_MERGED = 64;    // backward blocks of the list-order merged phase-1 grid
constexpr int SWEEP_ALTERNATE = 32; // k_sweep_struct: even blocks forward, odd blocks backward (default: list order)
constexpr int SWEEP_COARSE = 128;   // bulk items may span several tiles (api_fast.hip build_items, "coarse"): they are not the few latency-critical runs
constexpr int SWEEP_CKPT = 16;     // forward: store X only at the positions p % 8 == 0 (and the item's last one): the
                                   // factored counts recompute the rest from these checkpoints (estep_factored.hip)
constexpr int SWEEP_MERGE = 256;    // forward REPAIR over a list of ALL tiles: verify each, rewrite the flagged ones until they meet their stored trajectory (FwdCtl)
// What the forward sweeps need beyond the tables.
// prevx (speculative sweeps, round 6): the start vector of every item's warm-up, taken from the PREVIOUS E-step's X at the position the warm-up
// starts (api_fast.hip), instead of the stationary vector: the parameters of consecutive EM rounds are close, so the error the warm-up has to
// forget starts decades lower.
// The rest: the FIX pass (REPAIR with SWEEP_MERGE; launch_fwd_fix), which runs between the forward sweep and the back half.  Every row takes a
// tile, compares the vector the tile was built on with what its neighbour computed (k_verify's test), and where they disagree rewrites the
// tile from the true vector -- but only until the new trajectory has the DIRECTION of the stored one again (checked at the end of every full
// 16-bin block): a speculation that fell short by a decade or two is wrong for a few hundred bins, not for a tile.  Two forward vectors over
// the same observations never meet in VALUE (A x' -> c A x, c the ratio of the likelihoods of what follows given the two starts, 1 + O(mismatch),
// for ever), so the pass records the factor between the stored rows above the merge point and its own rows below it (finv = 1/c, fmerge = the
// block) and the two consumers of both halves put it in: the back half scales its posterior weight by finv when it steps from the stored part
// into the rewritten part (estep_fused.hip), k_ll adds log c.  Everything else that reads X is scale-free.  Because the pass is over before the
// back half starts, nothing is counted twice (rounds 1-5: the whole tile again, then its group of the counts again, 8 ms per round at genome
// size), and nothing waits for anything: a first version let the repairs run UNDER the counts, behind a per-tile flag the counts waited on --
// a vector wave beside a wave that streams f64 matrix instructions crawls (ten times slower: HISTORY.md), and the waits timed out.
// mis / mlen (host-mapped): each tile's mismatch and the blocks its repair took -- what the per-tile warm-ups follow (api_fast.hip adapt_warmups).
struct FwdCtl { const double *prevx; double tol; double *mis; int *mlen; int *fmerge; double *finv; };

// ------------------------------------------------------------------ forward
// per-row bookkeeping of the tile boundary the sweep crosses next
struct FwdCursor { int next_lo, tile; };

// MODE 0: per-step range / store / boundary predicates;  MODE 1: all 16 positions are computed and
// stored and no tile starts inside the block;  MODE 2: warm-up, nothing stored.
template <int MODE, int J, int NPL, bool CK, int LPT>
__device__ __forceinline__ void fwd_step(const StructParN<NPL> &c, const Half8Masks &hm, const double *lds_e, int k0, int m, unsigned w, int base,
                                         int p_first, int p_last, int lo0, int T, FwdCursor &cur, double (&x)[NPL],
                                         double *fo, double *io, double *entry, int g, double &inv_keep)
{
	// CK: checkpoints only (X at p % 8 == 0 and the item's last position); g (wave-uniform): the group's number in its block
	// inv_keep: lane m < 4 of a tile keeps the scale factor of group m; the block stores the four together (MODE 1)
	// base = index of the group's first position (multiple of 4), J = step inside the group;
	// lo0 = first position whose X is stored (INT_MAX for a walk, which only leaves the boundary vectors)
	const int p = base + J + 1, idx = base + J;
	if (MODE == 0 && !(p >= p_first && p <= p_last)) return;
	constexpr int S = LPT * NPL;
	if (MODE == 0 && p == cur.next_lo) { // the X_{lo-1} this tile builds on
		storeN<NPL>(entry + (int64_t)cur.tile * S + k0, x);
		cur.tile += 1; cur.next_lo += T;
	}
	double ev[NPL];
	ev_load<NPL, LPT>(lds_e + sym_of<J>(w) * S, k0, ev);
	if (J == 3) { // p % NORM_EVERY == 0 (groups are 4-aligned): d_p = sum(X_{p-1}) rounded down to a power of two, off the critical path
		const double inv = pow2_rcp(tile_sum<NPL, LPT>(x));
#pragma unroll
		for (int i = 0; i < NPL; ++i) ev[i] *= inv;
		if (MODE == 1) inv_keep = m == g ? inv : inv_keep; // one 32-byte store per tile and block instead of four 8-byte ones
		else if (MODE == 0 && p >= lo0 && m == 0) io[idx] = inv;
	}
	tile_step<NPL, LPT>(c, x, hm);
#pragma unroll
	for (int i = 0; i < NPL; ++i) x[i] *= ev[i];
	if (MODE == 1) { if (!CK || (J == 3 && (g & 1))) storeN<NPL>(fo + (int64_t)idx * S, x); }
	else if (MODE == 0 && p >= lo0 && (!CK || (p & 7) == 0 || p == p_last)) storeN<NPL>(fo + (int64_t)idx * S, x);
}
template <int MODE, int NPL, bool CK, int LPT>
__device__ __forceinline__ void fwd_block(const StructParN<NPL> &c, const Half8Masks &hm, const double *lds_e, int k0, int m, const uint4 sv, int base,
                                          int p_first, int p_last, int lo0, int T, FwdCursor &cur, double (&x)[NPL],
                                          double *fo, double *io, double *entry)
{
	double inv_keep = 1.0;
	// four groups of four unrolled steps: 16 fully unrolled steps x 3 modes x 2 directions overflow the
	// instruction cache once the forward, backward and count kernels run side by side
#pragma unroll 1
	for (int g = 0; g < 4; ++g) {
		const unsigned w = sym_word(sv, g);
		const int pb = base + 4 * g;
		fwd_step<MODE, 0, NPL, CK, LPT>(c, hm, lds_e, k0, m, w, pb, p_first, p_last, lo0, T, cur, x, fo, io, entry, g, inv_keep);
		fwd_step<MODE, 1, NPL, CK, LPT>(c, hm, lds_e, k0, m, w, pb, p_first, p_last, lo0, T, cur, x, fo, io, entry, g, inv_keep);
		fwd_step<MODE, 2, NPL, CK, LPT>(c, hm, lds_e, k0, m, w, pb, p_first, p_last, lo0, T, cur, x, fo, io, entry, g, inv_keep);
		fwd_step<MODE, 3, NPL, CK, LPT>(c, hm, lds_e, k0, m, w, pb, p_first, p_last, lo0, T, cur, x, fo, io, entry, g, inv_keep);
	}
	if (MODE == 1 && m < 4) io[base + 4 * m + 3] = inv_keep; // 1/d_p of the block's four normalising positions
}

// items[4*blockIdx.x + row] = work of this row.  REPAIR: the list holds the flagged tiles (count 1); a
// row starts from the neighbour's stored X_{lo-1} and recomputes its whole tile (the verify kernel
// then decides whether the next tile has to follow).  No vector-memory load inside the sweep.
template <bool REPAIR, int NPL, bool CK = false, int LPT = 16>
__device__ __forceinline__ void fwd_struct_body(int block, const double *__restrict__ sp, const double *__restrict__ e,
                                                const double *__restrict__ a0, const uint8_t *__restrict__ obs,
                                                const Chunk *__restrict__ chunks, const SweepItem *__restrict__ items,
                                                int n_items, int W, int T, int flags, double *__restrict__ f,
                                                double *__restrict__ invd, double *__restrict__ entry,
                                                int *__restrict__ touch_f, const FwdCtl ctl = FwdCtl{nullptr, 0.0, nullptr, nullptr, nullptr, nullptr})
{
	const bool walk = (flags & SWEEP_WALK) != 0, from_entry = (flags & SWEEP_FROM_ENTRY) != 0; // SWEEP_CKPT: the CK instantiation
	constexpr int S = LPT * NPL, R = 64 / LPT; // R tiles per wave
	__shared__ double lds_e[4 * S]; // e[0], e[1], 1, 1
	const int lane = threadIdx.x, m = lane & (LPT - 1), k0 = NPL * m;
	const Half8Masks hm = half8_masks(lane);
	fill_lds_e<NPL, LPT>(lds_e, e, lane);
	__syncthreads();
	const int slot = block * R + lane / LPT;
	const bool valid = slot < n_items;
	const SweepItem it = items[valid ? slot : 0];
#ifdef PSMC_TRACE_SWEEP
	const bool tr = !REPAIR && !walk && !from_entry;
	bool tr_first = true;
	if (tr) PSMC_TRACE(g_trace_f, block, 0, wall_clock64());
#endif
	if (REPAIR || (__any(it.count > 1) && !(flags & SWEEP_COARSE))) __builtin_amdgcn_s_setprio(3); // few, latency-critical waves
	else __builtin_amdgcn_s_setprio(1); // ahead of the backward warm-up and the rest of phase 1: once its warm-up is done this sweep
	                                    // is paced by its stores and leaves the vector units to them
	const Chunk c = chunks[it.first];
	// a walk (65..128 states: k_walk_struct) stops where its last tile starts: the step at that position stores the tile's start
	// vector, and nobody reads what a walk computes inside the last tile of its run (round 3; see k_walk1_struct)
	const int p_last = (walk && !REPAIR) ? chunks[it.first + it.count - 1].lo : chunks[it.first + it.count - 1].hi;
	const uint8_t *o = obs + c.off;
	double *fo = f + c.off * S + k0, *io = invd + c.off;
	StructParN<NPL> sc;
	load_struct_par<NPL, LPT>(sp, k0, true, sc);
	double x[NPL];
	int p_first;
	// the fix pass (per row): is this tile's start vector what its neighbour computed?
	bool merging = false;
	if constexpr (REPAIR && !CK) {
		if (flags & SWEEP_MERGE) {
			const bool check = valid && c.lo > 1 && !(c.flags & CHUNK_ANCHOR_F);
			double mm = 0.0, scl = 1.0;
			if (check) { // max_k |u/|u| - v/|v|| / max_k v/|v|, u = the vector the tile was built on, v = the neighbour's X_{lo-1}
				double u[NPL];
				loadN<NPL>(entry + (int64_t)it.first * S + k0, u);
				loadN<NPL>(fo + (int64_t)(c.lo - 2) * S, x);
				const double su = tile_sum<NPL, LPT>(u), sv = tile_sum<NPL, LPT>(x);
				const double iu = 1.0 / su, iv = 1.0 / sv;
				double num = 0.0, den = 0.0;
#pragma unroll
				for (int i = 0; i < NPL; ++i) { num = fmax(num, fabs(u[i] * iu - x[i] * iv)); den = fmax(den, fabs(x[i] * iv)); }
				num = row_max16(num); den = row_max16(den);
				mm = num / den;
				scl = su * iv; // the true vector at the length of the one the tile used: the factor between the two trajectories stays near 1
			}
			if (valid && m == 0 && ctl.mis) ctl.mis[it.first] = check ? mm : -1.0;
			merging = check && !(mm <= ctl.tol) && scl > 0.0 && scl < 1e300; // (a NaN anywhere: left to the verify / repair rounds that follow)
			if (merging) {
#pragma unroll
				for (int i = 0; i < NPL; ++i) x[i] *= scl;
			}
		}
	}
	const bool fix = REPAIR && (flags & SWEEP_MERGE) != 0;
	if (REPAIR && valid && m == 0 && !fix) {
		if (!(flags & SWEEP_NO_TOUCH)) touch_f[it.first] = 1; // X / inv_d of this tile change
		if (ctl.fmerge) ctl.fmerge[it.first] = 0;               // ... all of it: what the fix pass recorded about it no longer applies
	}
	if (fix) { // x holds the start vector of a row that has work
		p_first = c.lo;
	} else if (REPAIR && c.lo > 1) { // from the neighbour's stored X_{lo-1}, or from the boundary vector a walk left
		if (from_entry) loadN<NPL>(entry + (int64_t)it.first * S + k0, x);
		else loadN<NPL>(fo + (int64_t)(c.lo - 2) * S, x);
		p_first = c.lo;
	} else {
		const int ws = max(1, c.lo - chunk_warm_f(c, W));
		if (!REPAIR && ctl.prevx && ws > 1) loadN<NPL>(ctl.prevx + (int64_t)it.first * S + k0, x); // the previous E-step's X at position ws - 1
		else loadN<NPL>(a0 + k0, x);
		if (ws == 1) { // true start: X_1 = a0*e[o_1], d_1 = 1 (khmm.c:171-174 without the division)
			double ev[NPL];
			ev_load<NPL, LPT>(lds_e + ((int)o[0] & 3) * S, k0, ev);
#pragma unroll
			for (int i = 0; i < NPL; ++i) x[i] *= ev[i];
			if (valid && c.lo == 1 && !walk) storeN<NPL>(fo, x);
			p_first = 2;
		} else { // warm-up from the stationary prior
			p_first = ws;
		}
	}
	// tile boundaries ahead: the head's own lo (unless it is position 1, which no step computes), then every T
	FwdCursor cur;
	const int lo_store = walk ? 0x7fffffff : c.lo;
	cur.tile = c.lo >= p_first ? it.first : it.first + 1;
	cur.next_lo = c.lo >= p_first ? c.lo : c.lo + T;
	const int b_first = (p_first - 1) >> 4;
	int nblk = (valid && p_last >= p_first && !(fix && !merging)) ? ((p_last - 1) >> 4) - b_first + 1 : 0;
	// row-uniform scalars (lane 16r speaks for row r)
	int64_t roff[R]; int rbf[R], rnb[R];
	int nb_max = 0;
#pragma unroll
	for (int r = 0; r < R; ++r) {
		roff[r] = readlane_i64(c.off, LPT * r);
		rbf[r] = __builtin_amdgcn_readlane(b_first, LPT * r);
		rnb[r] = __builtin_amdgcn_readlane(nblk, LPT * r);
		nb_max = max(nb_max, rnb[r]);
	}
#ifdef PSMC_TRACE_SWEEP
	if (tr) PSMC_TRACE(g_trace_f, block, 3, trace_hw() | ((unsigned long long)nb_max << 40));
#endif
	for (int bi = 0; bi < nb_max; ++bi) {
		int bb[R];
#pragma unroll
		for (int r = 0; r < R; ++r) bb[r] = rbf[r] + min(bi, max(rnb[r] - 1, 0));
		const uint4 sv = row_symbols(obs, roff, bb, lane / LPT);
		if (bi < nblk) {
			const int base = (b_first + bi) << 4;
			if (cur.next_lo == base + 1 && base + 1 >= p_first && base + 1 <= p_last) { // a tile starts exactly at this block
				storeN<NPL>(entry + (int64_t)cur.tile * S + k0, x);
						cur.tile += 1; cur.next_lo += T;
			}
			const bool full = base + 1 >= p_first && base + 16 <= p_last && !(cur.next_lo >= base + 1 && cur.next_lo <= base + 16);
			const int mode = !full ? 0 : (base + 1 >= lo_store ? 1 : 2);
#ifdef PSMC_TRACE_SWEEP
			if (tr && tr_first && __all(mode == 1)) { tr_first = false; PSMC_TRACE(g_trace_f, block, 1, wall_clock64()); }
#endif
			double old[NPL];
			bool chk = false;
			if constexpr (REPAIR && !CK) { // what the table holds at the block's last position, before this block overwrites it
				chk = merging && mode == 1;
				if (chk) loadN<NPL>(fo + (int64_t)(base + 15) * S, old);
			}
			if (__all(mode == 1)) fwd_block<1, NPL, CK, LPT>(sc, hm, lds_e, k0, m, sv, base, p_first, p_last, lo_store, T, cur, x, fo, io, entry);
			else if (__all(mode == 2)) fwd_block<2, NPL, CK, LPT>(sc, hm, lds_e, k0, m, sv, base, p_first, p_last, lo_store, T, cur, x, fo, io, entry);
			else fwd_block<0, NPL, CK, LPT>(sc, hm, lds_e, k0, m, sv, base, p_first, p_last, lo_store, T, cur, x, fo, io, entry);
			if constexpr (REPAIR && !CK) {
				if (__any(chk)) {
					double num = 0.0, den = 0.0, ratio = 1.0;
					if (chk) { // the two vectors as directions (both divided by their sum), like k_verify
						const double sn = tile_sum<NPL, LPT>(x), so = tile_sum<NPL, LPT>(old);
						const double in = 1.0 / sn, io_ = 1.0 / so;
						ratio = so * in; // stored / rewritten
#pragma unroll
						for (int i = 0; i < NPL; ++i) { num = fmax(num, fabs(x[i] * in - old[i] * io_)); den = fmax(den, fabs(old[i] * io_)); }
					}
					num = row_max16(num); den = row_max16(den);
					const bool hit = chk && num <= ctl.tol * den && ratio > 0.0 && ratio < 1e300; // (false when anything is NaN)
					if (hit) { // the rest of the tile stands as it is; the factor between the two parts goes on record
						nblk = bi + 1; merging = false;
						if (m == 0) {
							if (base + 16 < p_last) { ctl.finv[it.first] = ratio; ctl.fmerge[it.first] = bi + 1; } // (met in the tile's last block: a whole new tile, nothing to stitch)
							if (ctl.mlen) ctl.mlen[it.first] = bi + 1;
						}
					}
				}
			}
		}
		if constexpr (REPAIR) { if (!__any(bi + 1 < nblk)) break; } // every row has met its stored trajectory (or its end)
	}
	if constexpr (REPAIR && !CK) { // a row that ran to the end of its tile: a whole new tile, nothing to stitch
		if (fix && valid && merging && m == 0 && ctl.mlen) ctl.mlen[it.first] = -1;
	}
#ifdef PSMC_TRACE_SWEEP
	if (tr) PSMC_TRACE(g_trace_f, block, 2, wall_clock64());
#endif
}

template <bool REPAIR, int NPL, bool CK = false, int LPT = 16>
__global__ __launch_bounds__(64) void k_fwd_struct(const double *__restrict__ sp, const double *__restrict__ e,
                                                     const double *__restrict__ a0, const uint8_t *__restrict__ obs,
                                                     const Chunk *__restrict__ chunks, const SweepItem *__restrict__ items,
                                                     int n_items, int W, int T, int flags, double *__restrict__ f,
                                                     double *__restrict__ invd, double *__restrict__ entry,
                                                     int *__restrict__ touch_f, int *__restrict__ started, const FwdCtl ctl)
{
	announce_start(started);
	fwd_struct_body<REPAIR, NPL, CK, LPT>(blockIdx.x, sp, e, a0, obs, chunks, items, n_items, W, T, flags, f, invd, entry, touch_f, ctl);
}

// ------------------------------------------------------------------ backward
// bt_p = e[o_p] * (a bt_{p+1}) * sb_p, positions descending.  A tile owns bt[lo+1 .. top+1] and sb[lo..top]
// (top = min(hi, L-1)); bt[lo] is stored by the tile below as its boundary value unless lo == 1.
// The cursor holds the tile the row is in; when p passes its lo the row moves into the tile below.
struct BwdCursor { int lo, top, tile; bool store; }; // store: false for a walk, which only leaves the boundary vectors

// MODE 1: every position of the block is strictly inside (lo, top) of the current tile;
// MODE 2: warm-up above the top tile's top;  MODE 0: general.
template <int MODE, int J, int NPL, int LPT>
__device__ __forceinline__ void bwd_step(const StructParN<NPL> &c, const Half8Masks &hm, const double *lds_e, int k0, int m, unsigned w, int base,
                                         int p_first, int p_low, int T, BwdCursor &cur, double (&x)[NPL], double *bto,
                                         double *sbo, double *bentry, double *bexit)
{
	constexpr int S = LPT * NPL;
	const int p = base + J + 1, idx = base + J;
	if (MODE == 0 && !(p <= p_first && p >= p_low)) return;
	double ev[NPL];
	ev_load<NPL, LPT>(lds_e + sym_of<J>(w) * S, k0, ev);
	if (J == 3) { // sb_p = 1/sum(bt_{p+1}), off the critical path
		const double s = rcp_newton(tile_sum<NPL, LPT>(x));
#pragma unroll
		for (int i = 0; i < NPL; ++i) ev[i] *= s;
		if ((MODE == 1 || (MODE == 0 && p <= cur.top && cur.store)) && m == 0) sbo[idx] = s;
	}
	if (MODE == 0 && p == cur.top) { // the boundary vector this tile builds on
		if (cur.store) storeN<NPL>(bto + (int64_t)cur.top * S, x); // bt[top+1]
		storeN<NPL>(bentry + (int64_t)cur.tile * S + k0, x);
	}
	tile_step<NPL, LPT>(c, x, hm);
#pragma unroll
	for (int i = 0; i < NPL; ++i) x[i] *= ev[i];
	if (MODE == 1) storeN<NPL>(bto + (int64_t)idx * S, x);
	if (MODE == 0 && p <= cur.top) {
		if (cur.store && (p > cur.lo || cur.lo == 1)) storeN<NPL>(bto + (int64_t)idx * S, x);
		if (p == cur.lo) { // leaving the tile: hand over to the one below
			storeN<NPL>(bexit + (int64_t)cur.tile * S + k0, x);
			cur.tile -= 1; cur.top = cur.lo - 1; cur.lo -= T;
		}
	}
}
template <int MODE, int NPL, int LPT>
__device__ __forceinline__ void bwd_block(const StructParN<NPL> &c, const Half8Masks &hm, const double *lds_e, int k0, int m, const uint4 sv, int base,
                                          int p_first, int p_low, int T, BwdCursor &cur, double (&x)[NPL], double *bto,
                                          double *sbo, double *bentry, double *bexit)
{
#pragma unroll 1
	for (int g = 3; g >= 0; --g) {
		const unsigned w = sym_word(sv, g);
		const int pb = base + 4 * g;
		bwd_step<MODE, 3, NPL, LPT>(c, hm, lds_e, k0, m, w, pb, p_first, p_low, T, cur, x, bto, sbo, bentry, bexit);
		bwd_step<MODE, 2, NPL, LPT>(c, hm, lds_e, k0, m, w, pb, p_first, p_low, T, cur, x, bto, sbo, bentry, bexit);
		bwd_step<MODE, 1, NPL, LPT>(c, hm, lds_e, k0, m, w, pb, p_first, p_low, T, cur, x, bto, sbo, bentry, bexit);
		bwd_step<MODE, 0, NPL, LPT>(c, hm, lds_e, k0, m, w, pb, p_first, p_low, T, cur, x, bto, sbo, bentry, bexit);
	}
}

// items: tiles first .. first+count-1, walked from the highest down.
template <bool REPAIR, int NPL, int LPT = 16>
__device__ __forceinline__ void bwd_struct_body(int block, const double *__restrict__ sp, const double *__restrict__ e,
                                                const uint8_t *__restrict__ obs, const Chunk *__restrict__ chunks,
                                                const SweepItem *__restrict__ items, int n_items, int W, int T,
                                                int flags, double *__restrict__ bt, double *__restrict__ sb,
                                                double *__restrict__ bentry, double *__restrict__ bexit,
                                                int *__restrict__ touch_b)
{
	constexpr int S = LPT * NPL, R = 64 / LPT;
	__shared__ double lds_e[4 * S];
	const int lane = threadIdx.x, m = lane & (LPT - 1), k0 = NPL * m;
	const Half8Masks hm = half8_masks(lane);
	fill_lds_e<NPL, LPT>(lds_e, e, lane);
	__syncthreads();
	const int slot = block * R + lane / LPT;
	const SweepItem it = items[slot < n_items ? slot : 0];
#ifdef PSMC_TRACE_SWEEP
	const bool trb = !REPAIR && (flags & (SWEEP_TOP_ONLY | SWEEP_COARSE));
	const unsigned long long trb_c0 = __builtin_readcyclecounter();
	if (trb) PSMC_TRACE(g_trace_b, block, 0, wall_clock64());
#endif
	if (REPAIR || (__any(it.count > 1) && !(flags & SWEEP_COARSE))) __builtin_amdgcn_s_setprio(3); // few, latency-critical waves
	else if ((flags & (SWEEP_TOP_ONLY | SWEEP_COARSE)) && (flags & SWEEP_MERGED)) __builtin_amdgcn_s_setprio(1); // list-order merged grid: second wave on the SIMD of a forward
	                                                                                              // block (priority 1) with three quarters of its steps: ends with it
	const int t_top = it.first + it.count - 1;
	const Chunk c = chunks[t_top];
	const int L = c.L;
	BwdCursor cur;
	cur.lo = c.lo; cur.top = min(c.hi, L - 1); cur.tile = t_top;
	// TOP_ONLY: the warm-up alone; a walk stops where the lowest tile's start vector (bentry) is stored
	const int p_low = (flags & SWEEP_TOP_ONLY) ? cur.top
	                  : ((!REPAIR && (flags & SWEEP_WALK)) ? min(chunks[it.first].hi, chunks[it.first].L - 1) : chunks[it.first].lo);
	const bool valid = slot < n_items && cur.top >= cur.lo; // a tile holding only position L owns no transition
	const bool walk = (flags & (SWEEP_WALK | SWEEP_TOP_ONLY)) != 0, from_entry = (flags & SWEEP_FROM_ENTRY) != 0;
	cur.store = !walk;
	const uint8_t *o = obs + c.off;
	double *bto = bt + c.off * S + k0, *sbo = sb + c.off;
	StructParN<NPL> sc;
	load_struct_par<NPL, LPT>(sp, k0, false, sc);
	double x[NPL]; // bt_{p+1} = e[o_{p+1}] * B_{p+1} (own scaling)
	int p_first;
	if (REPAIR) { // continue from the value the tile above computed at our top boundary
		if (valid && m == 0 && !(flags & SWEEP_NO_TOUCH)) touch_b[t_top] = 1;
		if (from_entry) loadN<NPL>(bentry + (int64_t)t_top * S + k0, x); // the boundary vector a walk left for this tile
		else loadN<NPL>(bexit + (int64_t)(t_top + 1) * S + k0, x);
		p_first = cur.top;
	} else {
		const int q = min(c.hi + chunk_warm_b(c, W) + 1, L); // B_q := 1
		ev_load<NPL, LPT>(lds_e + ((int)o[q - 1] & 3) * S, k0, x);
		p_first = q - 1;
	}
	// highest block.  A tile that holds only position L (not valid: it owns no transition) has p_first = L - 1, which is 0 for a
	// one-bin segment: block -1 would be read 16 bytes BELOW the segment's observations -- for the first segment below the
	// allocation (harmless in value, the row is idle, but a page fault whenever nothing is mapped there; found with
	// profiles/experiments/dbg_flaky_tiling.py, which recycles device memory between contexts)
	const int b_first = max((p_first - 1) >> 4, 0);
	const int nblk = valid ? b_first - ((p_low - 1) >> 4) + 1 : 0;
	int64_t roff[R]; int rbf[R], rnb[R];
	int nb_max = 0;
#pragma unroll
	for (int r = 0; r < R; ++r) {
		roff[r] = readlane_i64(c.off, LPT * r);
		rbf[r] = __builtin_amdgcn_readlane(b_first, LPT * r);
		rnb[r] = __builtin_amdgcn_readlane(nblk, LPT * r);
		nb_max = max(nb_max, rnb[r]);
	}
#ifdef PSMC_TRACE_SWEEP
	if (trb) PSMC_TRACE(g_trace_b, block, 3, trace_hw() | ((unsigned long long)nb_max << 40));
#endif
	for (int bi = 0; bi < nb_max; ++bi) {
		int bb[R];
#pragma unroll
		for (int r = 0; r < R; ++r) bb[r] = rbf[r] - min(bi, max(rnb[r] - 1, 0));
		const uint4 sv = row_symbols(obs, roff, bb, lane / LPT);
		if (bi < nblk) {
			const int base = (b_first - bi) << 4;
			int mode = (base + 1 > cur.lo && base + 16 < cur.top) ? 1 : ((base + 1 > cur.top && base + 16 <= p_first) ? 2 : 0);
			if (walk && mode == 1) mode = 2;
			if (__all(mode == 1)) bwd_block<1, NPL, LPT>(sc, hm, lds_e, k0, m, sv, base, p_first, p_low, T, cur, x, bto, sbo, bentry, bexit);
			else if (__all(mode == 2)) bwd_block<2, NPL, LPT>(sc, hm, lds_e, k0, m, sv, base, p_first, p_low, T, cur, x, bto, sbo, bentry, bexit);
			else bwd_block<0, NPL, LPT>(sc, hm, lds_e, k0, m, sv, base, p_first, p_low, T, cur, x, bto, sbo, bentry, bexit);
		}
	}
#ifdef PSMC_TRACE_SWEEP
	if (trb) { PSMC_TRACE(g_trace_b, block, 2, wall_clock64()); PSMC_TRACE(g_trace_b, block, 1, __builtin_readcyclecounter() - trb_c0); } // [1]: shader-clock cycles
#endif
}

template <bool REPAIR, int NPL, int LPT = 16>
__global__ __launch_bounds__(64) void k_bwd_struct(const double *__restrict__ sp, const double *__restrict__ e,
                                                     const uint8_t *__restrict__ obs, const Chunk *__restrict__ chunks,
                                                     const SweepItem *__restrict__ items, int n_items, int W, int T,
                                                     int flags, double *__restrict__ bt, double *__restrict__ sb,
                                                     double *__restrict__ bentry, double *__restrict__ bexit,
                                                     int *__restrict__ touch_b)
{
	bwd_struct_body<REPAIR, NPL, LPT>(blockIdx.x, sp, e, obs, chunks, items, n_items, W, T, flags, bt, sb, bentry, bexit, touch_b);
}

// Both directions' walks over the glued runs in ONE launch (blocks [0, nbf) forward, the rest backward):
// a process only gets a handful of hardware queues, and a walk that shares one with another stream's
// kernel would wait for it.
template <int NPL>
__global__ __launch_bounds__(64) void k_walk_struct(const double *__restrict__ sp, const double *__restrict__ e,
                                                      const double *__restrict__ a0, const uint8_t *__restrict__ obs,
                                                      const Chunk *__restrict__ chunks, const SweepItem *__restrict__ items_f,
                                                      int n_f, const SweepItem *__restrict__ items_b, int n_b, int W, int T,
                                                      double *__restrict__ entry, double *__restrict__ bentry,
                                                      double *__restrict__ bexit, int *__restrict__ started)
{
	announce_start(started);
	const int nbf = (n_f + 3) / 4;
	if ((int)blockIdx.x < nbf)
		fwd_struct_body<false, NPL>(blockIdx.x, sp, e, a0, obs, chunks, items_f, n_f, W, T, SWEEP_WALK, nullptr, nullptr, entry, nullptr);
	else
		bwd_struct_body<false, NPL>(blockIdx.x - nbf, sp, e, obs, chunks, items_b, n_b, W, T, SWEEP_WALK, nullptr, nullptr, bentry, bexit,
		                       nullptr);
}

// ------------------------------------------------------------------ walks, one state per lane
// A walk is pure latency (one run, ~50 k sequential steps), so it gets a wave of its own with one state per
// lane: 54 instructions per step instead of 85 (struct_step1: two 64-lane scans), no LDS, symbols from
// scalar loads fetched a group ahead.  Blocks [0, n_f) walk the forward runs, the rest the backward runs.
__device__ __forceinline__ double walk_ev(int sym, double e0, double e1) { return sym == 0 ? e0 : (sym == 1 ? e1 : 1.0); }

__global__ __launch_bounds__(64) void k_walk1_struct(const double *__restrict__ sp, const double *__restrict__ e,
                                                       const double *__restrict__ a0, const uint8_t *__restrict__ obs,
                                                       const Chunk *__restrict__ chunks, const SweepItem *__restrict__ items_f,
                                                       int n_f, const SweepItem *__restrict__ items_b, int n_b, int W, int T,
                                                       double *__restrict__ entry, double *__restrict__ bentry,
                                                       double *__restrict__ bexit, int *__restrict__ started)
{
	const int lane = threadIdx.x, block = (int)blockIdx.x;
	announce_start(started);
#ifdef PSMC_TRACE_SWEEP
	PSMC_TRACE(g_trace_w, block, 0, wall_clock64()); PSMC_TRACE(g_trace_w, block, 3, trace_hw());
	struct TraceEnd { int b; __device__ ~TraceEnd() { if (threadIdx.x == 0 && b < 16384) g_trace_w[4 * b + 2] = wall_clock64(); } } trace_end{block};
#endif
	__builtin_amdgcn_s_setprio(3);
	const WaveScanMasks wm = wave_scan_masks(lane);
	const double e0 = e[lane], e1 = e[64 + lane];
	if (block < n_f) { // ---------------- forward
		const SweepItem it = items_f[block];
		const Chunk c = chunks[it.first];
		// count = -k <= 0: walk through k tiles and stop where tile first + k starts -- only its start vector is wanted.  k = 0:
		// the head of a transfer-matrix chain (the warm-up alone; the head tile itself is the first matrix of the chain); k = 1
		// there: a segment's first tile, which has no X_0 for a matrix to start from; and since round 3 every SHORT run of
		// r tiles is the item (first, -(r - 1)): nobody reads what a walk computes inside the LAST tile of its run (the run
		// tiles are recomputed from the boundary vectors), and those T steps were the end of the longest walks
		const bool warm_only = it.count <= 0;
		const int thru = warm_only ? -it.count : it.count;
		const int p_last = thru == 0 ? c.lo - 1 : chunks[it.first + thru - 1].hi;
		const uint8_t *o = obs + c.off;
		StructPar1 s1; s1.mS = sp[lane]; s1.wS = sp[128 + lane]; s1.mP = sp[64 + lane]; s1.wP = sp[192 + lane]; s1.dd = sp[256 + lane];
		double x = a0[lane];
		int p_first;
		const int ws = max(1, c.lo - chunk_warm_f(c, W));
		if (ws == 1) { x *= walk_ev((int)o[0] & 3, e0, e1); p_first = 2; } else p_first = ws;
		int tile = c.lo >= p_first ? it.first : it.first + 1, next_lo = c.lo >= p_first ? c.lo : c.lo + T;
		// groups of four positions aligned to p % 4 == 1 .. 0 (indices 4g .. 4g+3): the last one normalises
		const int g_first = (p_first - 1) >> 2, g_last = (p_last - 1) >> 2;
		unsigned w = *reinterpret_cast<const unsigned *>(o + 4 * g_first);
		for (int g = g_first; g <= g_last; ++g) {
			const unsigned wn = *reinterpret_cast<const unsigned *>(o + 4 * min(g + 1, g_last)); // next group's symbols
#pragma unroll
			for (int j = 0; j < 4; ++j) {
				const int p = 4 * g + j + 1;
				if (p < p_first || p > p_last) continue;
				if (p == next_lo) { entry[(int64_t)tile * 64 + lane] = x; ++tile; next_lo += T; }
				double ev = walk_ev((int)((w >> (8 * j)) & 3u), e0, e1);
				if (j == 3) ev *= rcp_newton(first_lane_f64(wave_sum_nat(x)));
				x = struct_step1(s1, x, wm) * ev;
			}
			w = wn;
		}
		if (warm_only) entry[(int64_t)(it.first + thru) * 64 + lane] = x;
	} else { // ---------------- backward
		const SweepItem it = items_b[block - n_f];
		const bool warm_only = it.count == 0; // the head (top tile) of a chain: only bt_{top+1}
		int tile = it.first + max(it.count, 1) - 1;
		const Chunk c = chunks[tile];
		int lo = c.lo, top = min(c.hi, c.L - 1);
		// a run of r tiles: stop where the lowest tile's start vector (bentry) exists -- its own T steps are for the recomputation
		const int L = c.L, p_low = warm_only ? top + 1 : chunks[it.first + 1].lo;
		if (top < lo) return;
		const uint8_t *o = obs + c.off;
		StructPar1 s1; s1.mS = sp[192 + lane]; s1.wS = sp[64 + lane]; s1.mP = sp[128 + lane]; s1.wP = sp[lane]; s1.dd = sp[256 + lane];
		const int q = min(c.hi + chunk_warm_b(c, W) + 1, L); // B_q := 1
		double x = walk_ev((int)o[q - 1] & 3, e0, e1);
		const int p_first = q - 1;
		const int g_first = (p_first - 1) >> 2, g_last = (p_low - 1) >> 2;
		unsigned w = *reinterpret_cast<const unsigned *>(o + 4 * g_first);
		for (int g = g_first; g >= g_last; --g) {
			const unsigned wn = *reinterpret_cast<const unsigned *>(o + 4 * max(g - 1, g_last));
#pragma unroll
			for (int j = 3; j >= 0; --j) {
				const int p = 4 * g + j + 1;
				if (p > p_first || p < p_low) continue;
				double ev = walk_ev((int)((w >> (8 * j)) & 3u), e0, e1);
				if (j == 3) ev *= rcp_newton(first_lane_f64(wave_sum_nat(x)));
				if (p == top) bentry[(int64_t)tile * 64 + lane] = x; // the boundary vector this tile builds on
				x = struct_step1(s1, x, wm) * ev;
				if (p == lo) { bexit[(int64_t)tile * 64 + lane] = x; --tile; top = lo - 1; lo -= T; }
			}
			w = wn;
		}
		if (warm_only || it.count > 1) bentry[(int64_t)tile * 64 + lane] = x;
	}
}

// ------------------------------------------------------------------ transfer matrices of the long runs
// A walk over a run of r tiles is r*T sequential steps.  For the LONG runs the sequential part is taken out:
// the linear map of every tile (K_t: start vector -> exit vector) is computed column by column -- 64 unit
// vectors swept through the tile, 4 per wave, all tiles of all runs at once -- and a tiny chain kernel then
// applies the K_t one after the other (a 64 x 64 product each).  Columns are kept comparable by scaling
// with powers of two only (exact) and carrying the exponent.
// dir: 0 forward (lo..hi), 1 backward (top..lo)
struct KcTile { int tile, dir; };
__host__ __device__ inline void kc_range(const Chunk &c, int dir, int &lo, int &top) // the positions a KcTile's matrices cover
{
	const bool fwd = dir == 0;
	top = fwd ? c.hi : min(c.hi, c.L - 1);
	lo = fwd ? c.lo : min(((c.lo + 3) & ~3) + 1, top + 1); // backward: the chain kernel takes the tile's last steps itself
}

template <int NPL>
__global__ __launch_bounds__(64) void k_kcol_struct(const double *__restrict__ sp, const double *__restrict__ e,
                                                      const uint8_t *__restrict__ obs, const Chunk *__restrict__ chunks,
                                                      const KcTile *__restrict__ kc, const int *__restrict__ uniq, double *__restrict__ Kcol,
                                                      double *__restrict__ Kexp)
{
	constexpr int S = 16 * NPL, BPT = S / 4; // S unit vectors per tile, four per wave: BPT blocks per tile
	__shared__ double lds_e[4 * S];
	const int lane = threadIdx.x, m = lane & 15, k0 = NPL * m;
	fill_lds_e<NPL, 16>(lds_e, e, lane);
	__syncthreads();
	const int j = blockIdx.x / BPT, col = 4 * (blockIdx.x % BPT) + (lane >> 4);
	// the transfer matrices head the longest dependency chain of the first phase (columns -> chain -> run tiles ->
	// the fused back half may start): ahead of the bulk sweeps, behind the walks
	__builtin_amdgcn_s_setprio(2);
	const KcTile kt = kc[uniq[j]]; // slot j's matrix is computed from this tile (tiles of missing data share one slot per direction)
	const Chunk c = chunks[kt.tile];
	const bool fwd = kt.dir == 0;
	// backward: the map stops above the tile's lowest scaled position p* (the first p >= lo with p % 4 == 0); the
	// chain kernel takes the last steps p* .. lo itself, with the real scale factor, so that the exit vector has
	// the scale a sweep would give it (the counts at a tile boundary mix bt of the two neighbours)
	const int top = fwd ? c.hi : min(c.hi, c.L - 1);
	const int lo = fwd ? c.lo : min(((c.lo + 3) & ~3) + 1, top + 1);
	const uint8_t *o = obs + c.off;
	StructParN<NPL> sc;
	load_struct_par<NPL>(sp, k0, fwd, sc);
	double x[NPL];
#pragma unroll
	for (int i = 0; i < NPL; ++i) x[i] = (k0 + i == col) ? 1.0 : 0.0;
	int E = 0;
	const int n = top - lo + 1;
	for (int q = 0; q < n; ++q) {
		const int p = fwd ? lo + q : top - q;
		const int sym = (int)o[p - 1] & 3; // wave-uniform: a scalar byte load
		double ev[NPL];
		ev_load<NPL>(lds_e + sym * S, k0, ev);
		if ((p & 3) == 0) { // rescale the column by a power of two and remember the exponent
			const double s = row_sum16(lane_sum<NPL>(x));
			const int ex = s > 0.0 ? __builtin_amdgcn_frexp_exp(s) : 0;
#pragma unroll
			for (int i = 0; i < NPL; ++i) x[i] = __builtin_amdgcn_ldexp(x[i], -ex);
			E += ex;
		}
		struct_step<NPL>(sc, x);
#pragma unroll
		for (int i = 0; i < NPL; ++i) x[i] *= ev[i];
	}
	storeN<NPL>(Kcol + ((int64_t)j * S + col) * S + k0, x);
	if (m == 0) Kexp[(int64_t)j * S + col] = (double)E;
}

// The same transfer matrix with one COLUMN per lane (what 64-state models run; 65..128 states keep k_kcol_struct, whose
// chain path ends earlier there: DESIGN.md section 3).  In k_kcol_struct a unit
// vector is a tile of the sweeps: 16 lanes x 4 states, every scan level a DPP round trip -- 85 vector instructions per
// step for four columns (1360 per step and tile).  Here the 64 states of a column sit in ONE lane's registers, so the
// prefix / suffix sums are plain serial FMA chains and every matrix constant is the same for all lanes: one LDS
// broadcast read.  A tile is a work-group of two waves -- wave w holds states 32w .. 32w+31 of all 64 columns -- that
// exchange two totals per step through LDS (the upper half's suffix total, the lower half's prefix total):
// 6 vector instructions per state and step, 2 x 192 + the exchange per step and tile.
//   kcc (host, fill_params): per direction  mS[64] | mP[64] | { wS.e[sym] | wP.e[sym] | dd.e[sym] } for sym = 0, 1, 2
// Scaling: any power of two both halves agree on will do (the chain kernel reads the exponent): every fourth step
// both waves take it from the four exchanged totals, which they both hold.
// One wave group per tile would be 3 712 dependent steps of ~1 000 cycles: longer than the rest of the phase.  The tile's
// steps are therefore cut into `sub` consecutive ranges with a transfer matrix each (matrix index = tile * sub + range,
// ranges in traversal order); the chain kernel applies them one after the other -- `sub` times as many products in the
// chain, `sub` times as many waves here.
// NQ = S / 32 waves per work-group, wave w holds states 32w .. 32w+31 of 64 columns (S / 64 column groups per matrix:
// grid = matrices x column groups).  128 states: four quarters; a wave's suffix sums lack the totals of the quarters
// above it, its prefix sums those of the quarters below it.
template <int NQ>
__global__ __launch_bounds__(64 * NQ) void k_kcol2_struct(const double *__restrict__ kcc, const uint8_t *__restrict__ obs,
                                                            const Chunk *__restrict__ chunks, const KcTile *__restrict__ kc, const int *__restrict__ uniq,
                                                            double *__restrict__ Kcol, double *__restrict__ Kexp, int sub, int prio)
{
	constexpr int S = 32 * NQ, NCG = S / 64, KD = 11 * S; // KD: doubles per direction = mS | mP | 3 x (wS.e | wP.e | dd.e)
	__shared__ double xch[2][NQ][2][64]; // [step parity][wave][S total, P total][column]
	__shared__ double tab[KD];           // this direction's constants: every lane reads the same address (a broadcast read, no
	                                     // bank conflict); as scalar operands they overflowed the SGPR file (85 spills through v_writelane)
	const int lane = threadIdx.x & 63;
	const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	const int j = (int)blockIdx.x / NCG, cg = (int)blockIdx.x % NCG, jt = j / sub, js = j % sub;
	const int col = 64 * cg + lane;
	if (prio >= 2) __builtin_amdgcn_s_setprio(2); else if (prio == 1) __builtin_amdgcn_s_setprio(1); // s_setprio takes an immediate
	const KcTile kt = kc[uniq[jt]]; // slot jt's matrices are computed from this tile
	const Chunk c = chunks[kt.tile];
	const bool fwd = kt.dir == 0;
	for (int i = threadIdx.x; i < KD; i += 64 * NQ) tab[i] = kcc[(fwd ? 0 : KD) + i];
	__syncthreads();
	int top, lo;
	kc_range(c, kt.dir, lo, top);
	const uint8_t *o = obs + c.off;
	const double *cc = tab + 32 * w; // this wave's part of every table
	double x[32];
#pragma unroll
	for (int k = 0; k < 32; ++k) x[k] = (32 * w + k == col) ? 1.0 : 0.0; // column `col` = unit vector e_col
	int E = 0;
	const int n = max(top - lo + 1, 0);
	const int q0 = (int)((int64_t)n * js / sub), q1 = (int)((int64_t)n * (js + 1) / sub); // this block's steps, in traversal order
	for (int q = q0; q < q1; ++q) {
		const int p = fwd ? lo + q : top - q;
		const int sym = min((int)o[p - 1] & 3, 2); // wave-uniform: a scalar byte load
		const double *ce = cc + 2 * S + sym * 3 * S; // wS.e | wP.e | dd.e of this symbol
		double y[32], sS = 0.0, sP = 0.0;
#pragma unroll
		for (int k = 31; k >= 0; --k) { sS = __builtin_fma(x[k], cc[k], sS); y[k] = ce[k] * sS; }          // inclusive suffix of x.mS in this part
#pragma unroll
		for (int k = 0; k < 32; ++k) {
			sP = __builtin_fma(x[k], cc[S + k], sP);                                                           // inclusive prefix of x.mP in this part
			y[k] = __builtin_fma(ce[2 * S + k], x[k], __builtin_fma(ce[S + k], sP, y[k]));
		}
		xch[q & 1][w][0][lane] = sS; xch[q & 1][w][1][lane] = sP;
		__syncthreads();
		if constexpr (NQ == 2) {
			const double S0 = xch[q & 1][0][0][lane], S1 = xch[q & 1][1][0][lane], P0 = xch[q & 1][0][1][lane], P1 = xch[q & 1][1][1][lane];
			// the lower half's suffix sums lack the upper half's total, the upper half's prefix sums the lower half's
			const double T = w == 0 ? S1 : P0;
			const double *cz = ce + (w == 0 ? 0 : S);
			if ((p & 3) == 0) { // rescale by a power of two both waves compute alike, remember the exponent
				const double mag = (S0 + S1) + (P0 + P1);
				const int ex = mag > 0.0 ? __builtin_amdgcn_frexp_exp(mag) : 0;
				const double sc = __builtin_amdgcn_ldexp(1.0, -ex);
#pragma unroll
				for (int k = 0; k < 32; ++k) x[k] = __builtin_fma(cz[k], T, y[k]) * sc;
				E += ex;
			} else {
#pragma unroll
				for (int k = 0; k < 32; ++k) x[k] = __builtin_fma(cz[k], T, y[k]);
			}
		} else {
			double Ts = 0.0, Tp = 0.0, mag = 0.0; // totals of the parts above / below this one, and of everything (same order in every wave)
#pragma unroll
			for (int v = 0; v < NQ; ++v) {
				const double Sv = xch[q & 1][v][0][lane], Pv = xch[q & 1][v][1][lane];
				Ts += v > w ? Sv : 0.0; Tp += v < w ? Pv : 0.0; mag += Sv + Pv;
			}
			double sc = 1.0;
			if ((p & 3) == 0) {
				const int ex = mag > 0.0 ? __builtin_amdgcn_frexp_exp(mag) : 0;
				sc = __builtin_amdgcn_ldexp(1.0, -ex);
				E += ex;
			}
#pragma unroll
			for (int k = 0; k < 32; ++k) x[k] = __builtin_fma(ce[k], Ts, __builtin_fma(ce[S + k], Tp, y[k])) * sc;
		}
	}
	double *out = Kcol + ((int64_t)j * S + col) * S + 32 * w;
#pragma unroll
	for (int k = 0; k < 32; ++k) out[k] = x[k];
	if (w == 0) Kexp[(int64_t)j * S + col] = (double)E;
}

// run r: forward (r < n_f): entry[first+1 .. first+count-1] from entry[first];  backward: bentry[first+count-2 .. first]
// from bentry[first+count-1].  The vectors are left with sum 1 (the verify kernel and k_ll are scale-free).
__device__ __forceinline__ double wave_max_f64(double v) {
#pragma unroll
	for (int m = 32; m >= 1; m >>= 1) v = fmax(v, __shfl_xor(v, m, 64));
	return v;
}
struct KcRun { int first, count, kc0, pad; }; // kc0: index of the run's first KcTile

// one-state-per-lane step for PER * 64 states: lane L holds states L and (PER == 2) L + 64
template <int PER>
__device__ __forceinline__ void struct_step1n(const StructPar1 (&c)[PER], double (&x)[PER], const WaveScanMasks &m)
{
	if constexpr (PER == 1) { x[0] = struct_step1(c[0], x[0], m); }
	else {
		// inclusive suffix / prefix sums over 128 states: each half scans itself, the lower half adds the total of
		// the upper one to its suffix sums, the upper half the total of the lower one to its prefix sums
		const double s1 = wave_suffix_incl(x[1] * c[1].mS, m), s0 = wave_suffix_incl(x[0] * c[0].mS, m);
		const double p0 = wave_prefix_incl_bc(x[0] * c[0].mP), p1 = wave_prefix_incl_bc(x[1] * c[1].mP);
		const double tot1 = readlane_f64(s1, 0), tot0 = readlane_f64(p0, 63);
		const double y0 = __builtin_fma(c[0].wS, s0 + tot1, __builtin_fma(c[0].wP, p0, c[0].dd * x[0]));
		const double y1 = __builtin_fma(c[1].wS, s1, __builtin_fma(c[1].wP, p1 + tot0, c[1].dd * x[1]));
		x[0] = y0; x[1] = y1;
	}
}

// Round 3: the chain is the sequential part of the runs' path, and a product used to cost 4-8 us (7 ms per launch at 128
// states, where it ran beside the bulk sweeps): one wave fetched a matrix 16 rows at a time and waited a full memory
// latency per batch.  Now a work-group of NW = S / RW waves shares a product -- wave w takes the RW = 32 / PER rows
// RW w .. RW w + RW - 1 (the source states it reads with v_readlane) -- and every wave has the NEXT matrix's rows in flight
// while it multiplies the current ones (two register buffers of RW x PER doubles); the partial sums meet in LDS (one
// barrier per product, double-buffered), every wave adds them in wave order and carries the whole vector, so that the
// normalisation, the backward sweep's last steps and the next product need no further exchange.  ~0.3 us per product.
template <int PER>
__global__ __launch_bounds__(64 * (PER == 1 ? 2 : 8)) void k_kchain_struct(const KcRun *__restrict__ runs, int n_f, const int *__restrict__ kslot,
                                                        const double *__restrict__ Kcol,
                                                        const double *__restrict__ Kexp, const double *__restrict__ sp,
                                                        const double *__restrict__ e, const uint8_t *__restrict__ obs,
                                                        const Chunk *__restrict__ chunks, double *__restrict__ entry,
                                                        double *__restrict__ bentry, int sub)
{
	constexpr int S = 64 * PER, RW = 32 / PER, NW = S / RW;
	__shared__ double part[2][NW][S];
	const int lane = threadIdx.x & 63;
	const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	__builtin_amdgcn_s_setprio(3); // the sequential part of the longest dependency chain of phase 1
	const bool fwd = (int)blockIdx.x < n_f;
	const KcRun r = runs[blockIdx.x];
	double *vec = fwd ? entry : bentry;
	int t = fwd ? r.first : r.first + r.count - 1;
	double x[PER];
	const WaveScanMasks wm = wave_scan_masks(lane);
	StructPar1 s1[PER]; // backward roles: mS = c, wS = R, mP = qa, wP = P (sp = P | R | qa | c | dd, S each)
	double e0[PER], e1[PER];
#pragma unroll
	for (int q = 0; q < PER; ++q) {
		const int k = lane + 64 * q;
		x[q] = vec[(int64_t)t * S + k];
		s1[q].mS = sp[3 * S + k]; s1[q].wS = sp[S + k]; s1[q].mP = sp[2 * S + k]; s1[q].wP = sp[k]; s1[q].dd = sp[4 * S + k];
		e0[q] = e[k]; e1[q] = e[S + k];
	}
	const int n_mats = (r.count - 1) * sub; // matrix j = tile j / sub of the run, range j % sub; the tile's matrices sit in slot kslot[kc0 + j / sub]
	const int row0 = RW * w, src_h = row0 >> 6, src_l = row0 & 63; // this wave's rows = source states row0 .. row0 + RW - 1: register src_h, lanes src_l ..
	double ma[RW][PER], mb[RW][PER], ea[PER], eb[PER];
	auto load = [&](int j, double (&m)[RW][PER], double (&ex)[PER]) {
		const int64_t kb = ((int64_t)kslot[r.kc0 + j / sub] * sub + j % sub) * S;
#pragma unroll
		for (int i = 0; i < RW; ++i)
#pragma unroll
			for (int h = 0; h < PER; ++h) m[i][h] = Kcol[(kb + row0 + i) * S + lane + 64 * h];
#pragma unroll
		for (int h = 0; h < PER; ++h) ex[h] = Kexp[kb + lane + 64 * h];
	};
	auto product = [&](int j, const double (&m)[RW][PER], const double (&ex)[PER]) {
		double xs[PER], y[PER], em = -1e300;
#pragma unroll
		for (int h = 0; h < PER; ++h) em = fmax(em, x[h] > 0.0 ? ex[h] : -1e300);
		const double emax = wave_max_f64(em);
#pragma unroll
		for (int h = 0; h < PER; ++h) { xs[h] = x[h] > 0.0 ? __builtin_amdgcn_ldexp(x[h], (int)(ex[h] - emax)) : 0.0; y[h] = 0.0; }
		double src = xs[0];
		if constexpr (PER == 2) src = src_h ? xs[1] : xs[0]; // wave-uniform choice
#pragma unroll
		for (int i = 0; i < RW; ++i) {
			const double v = readlane_f64(src, src_l + i);
#pragma unroll
			for (int h = 0; h < PER; ++h) y[h] = __builtin_fma(v, m[i][h], y[h]);
		}
		double (*pp)[S] = part[j & 1];
#pragma unroll
		for (int h = 0; h < PER; ++h) pp[w][lane + 64 * h] = y[h];
		__syncthreads();
#pragma unroll
		for (int h = 0; h < PER; ++h) {
			double tsum = pp[0][lane + 64 * h];
#pragma unroll
			for (int v = 1; v < NW; ++v) tsum += pp[v][lane + 64 * h];
			y[h] = tsum;
		}
		double tot = y[0];
		if constexpr (PER == 2) tot += y[1];
		const double inv = rcp_newton(first_lane_f64(wave_sum_nat(tot)));
#pragma unroll
		for (int h = 0; h < PER; ++h) x[h] = y[h] * inv;
	};
	auto tile_done = [&]() { // after the last range map of a tile: x = the vector at the tile's far boundary
		double y[PER];
#pragma unroll
		for (int h = 0; h < PER; ++h) y[h] = x[h];
		if (!fwd) { // the tile's last steps p* .. lo with the sweep's own scaling: bt_lo as the sweep leaves it (every wave, redundantly)
			const Chunk c = chunks[t];
			const int top = min(c.hi, c.L - 1), ps = min((c.lo + 3) & ~3, top);
			const uint8_t *o = obs + c.off;
			for (int pq = ps; pq >= c.lo; --pq) {
				const int sym = (int)o[pq - 1] & 3;
				double ev[PER];
#pragma unroll
				for (int h = 0; h < PER; ++h) ev[h] = walk_ev(sym, e0[h], e1[h]);
				if ((pq & 3) == 0) {
					double tot = y[0];
					if constexpr (PER == 2) tot += y[1];
					const double inv = rcp_newton(first_lane_f64(wave_sum_nat(tot)));
#pragma unroll
					for (int h = 0; h < PER; ++h) ev[h] *= inv;
				}
				struct_step1n<PER>(s1, y, wm);
#pragma unroll
				for (int h = 0; h < PER; ++h) y[h] *= ev[h];
			}
		}
		t += fwd ? 1 : -1;
#pragma unroll
		for (int h = 0; h < PER; ++h) { if (w == 0) vec[(int64_t)t * S + lane + 64 * h] = y[h]; x[h] = y[h]; }
	};
	if (n_mats > 0) load(0, ma, ea);
	for (int j = 0; j < n_mats; j += 2) { // two products per trip: the buffers swap roles without moving registers
		if (j + 1 < n_mats) load(j + 1, mb, eb);
		product(j, ma, ea);
		if (j % sub == sub - 1) tile_done();
		if (j + 1 >= n_mats) break;
		if (j + 2 < n_mats) load(j + 2, ma, ea);
		product(j + 1, mb, eb);
		if ((j + 1) % sub == sub - 1) tile_done();
	}
}

// The bulk of both sweeps in ONE grid (block order: see the kernel).  Two uses.  (1) The unfused back half (flags 0 / 0): both table writers share the chip from the first
// to the last wave.  (2) Phase 1 of the fused / factored E-step of a SHARD-SIZED input ("merge1", api_fast.hip plan_fast):
// the forward sweep (flags_f: SWEEP_CKPT or 0) and the warm-up-only backward pass (flags_b = SWEEP_TOP_ONLY).  Such an
// E-step has fewer waves than the device has SIMDs (1024), and the dispatcher places the waves of ONE grid on distinct
// SIMDs but knows nothing about the grids of other streams: two launches of 512 waves side by side leave 268 SIMDs
// with two waves and 268 idle (psmc_hip_place_probe, profiles/r03_place_probe.json), and every step of a
// latency-bound sweep then costs 508 cycles instead of 287.  Per launch: 2 x (8n+9) algorithmic bytes per bin in use
// (1), (8n+9) in use (2).
template <int NPL, bool CK, int LPT = 16>
__global__ __launch_bounds__(64) void k_sweep_struct(const double *__restrict__ sp, const double *__restrict__ e,
                                                       const double *__restrict__ a0, const uint8_t *__restrict__ obs,
                                                       const Chunk *__restrict__ chunks, const SweepItem *__restrict__ items_f,
                                                       int n_f, const SweepItem *__restrict__ items_b, int n_b, int W, int T,
                                                       int flags_f, int flags_b, double *__restrict__ f, double *__restrict__ invd,
                                                       double *__restrict__ entry, double *__restrict__ bt,
                                                       double *__restrict__ sb, double *__restrict__ bentry,
                                                       double *__restrict__ bexit, int *__restrict__ started, const double *__restrict__ prevx)
{
	announce_start(started);
	// Block order.  The dispatcher deals the work-groups of a grid out breadth first -- XCD = index % 8, then shader engine,
	// compute unit, SIMD -- so a short period in the direction pattern aliases with that hierarchy: with even = forward, odd =
	// backward (SWEEP_ALTERNATE: what the unfused back half runs, where both directions do the same work) every XCD holds ONE
	// direction.  For phase 1 of a shard-sized E-step, where the forward blocks (warm-up + tile + stores) outlast the
	// warm-up-only backward ones, that half of the device ended 0.3 ms after the other (scripts/sweep_trace.py); in list
	// order -- all forward blocks, then all backward blocks -- the second wave on a SIMD is of the other direction.
	constexpr int R = 64 / LPT; // tiles per wave
	const int nbf = (n_f + R - 1) / R, nbb = (n_b + R - 1) / R, both = 2 * min(nbf, nbb), b = blockIdx.x;
	bool fwd; int blk;
	if (!(flags_b & SWEEP_ALTERNATE)) { fwd = b < nbf; blk = fwd ? b : b - nbf; }
	else if (b < both) { fwd = (b & 1) == 0; blk = b >> 1; }
	else { fwd = nbf > nbb; blk = b - both + min(nbf, nbb); }
	if (fwd) fwd_struct_body<false, NPL, CK, LPT>(blk, sp, e, a0, obs, chunks, items_f, n_f, W, T, flags_f, f, invd, entry, nullptr, FwdCtl{prevx, 0.0, nullptr, nullptr, nullptr, nullptr});
	else bwd_struct_body<false, NPL, LPT>(blk, sp, e, obs, chunks, items_b, n_b, W, T, flags_b, bt, sb, bentry, bexit, nullptr);
}

// ------------------------------------------------------------------ dirty-tile lists
// deterministic compaction of the verify kernel's flags: out[0..cnt) = flagged tiles in ascending
// order, each as a one-tile sweep item
// host_out / host_cnt are host-mapped pinned memory: the host learns the count (and the list) without a
// copy command, which would be a blit kernel queued behind whatever is running
__global__ __launch_bounds__(64) void k_compact(const int *__restrict__ dirty, int n, SweepItem *__restrict__ out,
                                                  SweepItem *__restrict__ host_out, int *__restrict__ host_cnt)
{
	const int lane = threadIdx.x;
	int base = 0;
	for (int i0 = 0; i0 < n; i0 += 64) {
		const int i = i0 + lane;
		const bool d = i < n && dirty[i] != 0;
		const unsigned long long mask = __ballot(d);
		if (d) {
			SweepItem s; s.first = i; s.count = 1;
			const int at = base + __popcll(mask & ((1ull << lane) - 1ull));
			out[at] = s;
			if (host_out) host_out[at] = s;
		}
		base += __popcll(mask);
	}
	if (lane == 0) *host_cnt = base;
	__threadfence_system();
}

// ------------------------------------------------------------------ launchers
// which: 0 = the sweep items [first, first+n), 1 = flagged tiles of the current repair round,
//        3 = every tile of the glued runs, recomputed from the boundary vector its walk / chain left,
//        4 (backward) = warm-up only: leave the start vector of every item's top tile in bentry (fused backward +
//        counts), 5 (backward) = flagged tiles of the current repair round, boundary vectors only (fused: no bt table
//        to repair)
void launch_fwd_struct(const EstepLaunch &p, hipStream_t st, int which, int first, int n_items)
{
	if (n_items <= 0) return;
	// the throughput-bound bulk launch of a 64-state model: eight tiles per wave ("lanes8": 8 lanes x 8 states, ~13 instead of 17 vector
	// instructions per tile-step; the latency-bound launches -- repairs, run tiles -- keep four, whose step is a third shorter)
	const bool l8 = p.lanes8 && p.ns == 64 && which == 0;
	const dim3 g(l8 ? (n_items + 7) / 8 : (n_items + 3) / 4), b(64);
	const SweepItem *items = (const SweepItem *)(which == 1 ? p.d_ritems_f : (which == 3 || which == 7 ? p.d_members_f : (which == 6 ? p.d_fix_f : p.d_items_f))) + first;
	// which 6 / 7: the fix pass over the tiles outside the glued runs / over the run tiles (one-tile items; launch_fast)
	const bool fixp = which == 6 || which == 7;
	const int flags = (which == 3 ? (SWEEP_FROM_ENTRY | SWEEP_NO_TOUCH) : 0) // run tiles are done before the counts start
	                  | (p.ckpt ? SWEEP_CKPT : 0) | (which == 0 && p.coarse > 1 ? SWEEP_COARSE : 0) | (fixp ? SWEEP_MERGE : 0);
	const FwdCtl ctl = {which == 0 ? p.d_prevx : nullptr, p.tol, fixp ? p.m_mis : nullptr, fixp ? p.m_mlen : nullptr, p.d_fmerge, p.d_finv};
#define PSMC_LF(REP, NPL, CK) hipLaunchKernelGGL((k_fwd_struct<REP, NPL, CK>), g, b, 0, st, p.d_sp, p.d_e, p.d_a0, p.d_obs, p.d_chunks, \
		items, n_items, p.warmup, p.tile_len, flags, p.d_f, p.d_s, p.d_entry, p.d_touch_f, which == 0 && p.d_gate ? p.d_gate + 1 : nullptr, ctl)
	const bool rep = which != 0;
	if (l8) {
#define PSMC_LF8(CK) hipLaunchKernelGGL((k_fwd_struct<false, 8, CK, 8>), g, b, 0, st, p.d_sp, p.d_e, p.d_a0, p.d_obs, p.d_chunks, \
		items, n_items, p.warmup, p.tile_len, flags, p.d_f, p.d_s, p.d_entry, p.d_touch_f, p.d_gate ? p.d_gate + 1 : nullptr, ctl)
		if (p.ckpt) PSMC_LF8(true); else PSMC_LF8(false);
#undef PSMC_LF8
	} else if (p.ns == 128) { if (rep) PSMC_LF(true, 8, false); else PSMC_LF(false, 8, false); }
	else if (p.ckpt) { if (rep) PSMC_LF(true, 4, true); else PSMC_LF(false, 4, true); } // checkpoint stores are a compile-time variant: no per-step branches in the full-table kernels
	else { if (rep) PSMC_LF(true, 4, false); else PSMC_LF(false, 4, false); }
	PSMC_DBG("launch_fwd_struct", which, first, n_items);
#undef PSMC_LF
}
void launch_bwd_struct(const EstepLaunch &p, hipStream_t st, int which, int first, int n_items)
{
	if (n_items <= 0) return;
	const bool l8 = p.lanes8 && p.ns == 64 && which == 4; // the warm-up pass of the fused / factored back half
	const dim3 g(l8 ? (n_items + 7) / 8 : (n_items + 3) / 4), b(64);
	const SweepItem *items = (const SweepItem *)(which == 1 || which == 5 ? p.d_ritems_b : (which == 3 ? p.d_members_b : p.d_items_b)) + first;
	// which == 4 with coarse items: the pass walks every item from the top tile's warm-up down to the lowest tile's top and leaves each tile's start vector
	const int flags = which == 5 ? SWEEP_WALK : (which == 3 ? (SWEEP_FROM_ENTRY | SWEEP_NO_TOUCH) : (which == 4 ? (p.coarse > 1 ? (SWEEP_WALK | SWEEP_COARSE) : SWEEP_TOP_ONLY) : 0));
#define PSMC_LB(REP, NPL) hipLaunchKernelGGL((k_bwd_struct<REP, NPL>), g, b, 0, st, p.d_sp, p.d_e, p.d_obs, p.d_chunks, items, \
		n_items, p.warmup, p.tile_len, flags, p.d_b, p.d_sb, p.d_bentry, p.d_bexit, p.d_touch_b)
	const bool rep = !(which == 0 || which == 4);
	if (l8)
		hipLaunchKernelGGL((k_bwd_struct<false, 8, 8>), g, b, 0, st, p.d_sp, p.d_e, p.d_obs, p.d_chunks, items, n_items, p.warmup, p.tile_len, flags,
		                   p.d_b, p.d_sb, p.d_bentry, p.d_bexit, p.d_touch_b);
	else if (p.ns == 128) { if (rep) PSMC_LB(true, 8); else PSMC_LB(false, 8); }
	else { if (rep) PSMC_LB(true, 4); else PSMC_LB(false, 4); }
	PSMC_DBG("launch_bwd_struct", which, first, n_items);
#undef PSMC_LB
}
void launch_compact(const EstepLaunch &p, hipStream_t st, bool bwd)
{
	hipLaunchKernelGGL(k_compact, dim3(1), dim3(64), 0, st, bwd ? p.d_dirty_b : p.d_dirty, p.n_chunks,
	                   (SweepItem *)(bwd ? p.d_ritems_b : p.d_ritems_f),
	                   (SweepItem *)(p.m_ritems ? p.m_ritems + (size_t)(bwd ? 1 : 0) * 2 * p.n_chunks : nullptr), p.m_cnt + (bwd ? 1 : 0));
	PSMC_DBG("verify + launch_compact", bwd, p.n_chunks, 0);
}
// bulk of both sweeps in one grid: forward items [ff, ff+nf) and backward items [fb, fb+nb); top_only: the backward
// blocks do the warm-up-only pass of the fused / factored back half (phase 1 of a shard-sized E-step)
void launch_sweeps(const EstepLaunch &p, hipStream_t st, int ff, int nf, int fb, int nb, bool top_only)
{
	const bool l8 = p.lanes8 && p.ns == 64 && top_only; // phase 1 of the fused / factored plans (see launch_fwd_struct)
	const int nblk = l8 ? (nf + 7) / 8 + (nb + 7) / 8 : (nf + 3) / 4 + (nb + 3) / 4;
	if (nblk <= 0) return;
	const int co = top_only && p.coarse > 1 ? SWEEP_COARSE : 0; // coarse items: the fused / factored plans only (api_fast.hip enqueue_fast)
	const int flags_f = (p.ckpt ? SWEEP_CKPT : 0) | co,
	          flags_b = (top_only ? (co ? SWEEP_WALK | co : SWEEP_TOP_ONLY) : 0) | (top_only && p.merge_order ? SWEEP_MERGED : SWEEP_ALTERNATE);
#define PSMC_LS(NPL, CK) hipLaunchKernelGGL((k_sweep_struct<NPL, CK>), dim3(nblk), dim3(64), 0, st, p.d_sp, p.d_e, p.d_a0, p.d_obs, p.d_chunks, \
		(const SweepItem *)p.d_items_f + ff, nf, (const SweepItem *)p.d_items_b + fb, nb, p.warmup, p.tile_len, flags_f, flags_b, \
		p.d_f, p.d_s, p.d_entry, p.d_b, p.d_sb, p.d_bentry, p.d_bexit, p.d_gate ? p.d_gate + 1 : nullptr, p.d_prevx)
#define PSMC_LS8(CK) hipLaunchKernelGGL((k_sweep_struct<8, CK, 8>), dim3(nblk), dim3(64), 0, st, p.d_sp, p.d_e, p.d_a0, p.d_obs, p.d_chunks, \
		(const SweepItem *)p.d_items_f + ff, nf, (const SweepItem *)p.d_items_b + fb, nb, p.warmup, p.tile_len, flags_f, flags_b, \
		p.d_f, p.d_s, p.d_entry, p.d_b, p.d_sb, p.d_bentry, p.d_bexit, p.d_gate ? p.d_gate + 1 : nullptr, p.d_prevx)
	if (l8) { if (p.ckpt) PSMC_LS8(true); else PSMC_LS8(false); }
	else if (p.ns == 128) PSMC_LS(8, false); else if (p.ckpt) PSMC_LS(4, true); else PSMC_LS(4, false);
#undef PSMC_LS8
	PSMC_DBG("launch_sweeps", nf, nb, top_only);
#undef PSMC_LS
}
// Start vectors of the forward speculation from the previous E-step's table (FwdCtl::prevx): row lo - wf - 1 of the head tile of every bulk
// item, copied out BEFORE the sweep starts to overwrite the table (a wave reading the row inside the sweep could see it half rewritten by the
// wave of the tile below: a good vector either way, but a result that depends on timing).  ck: the table holds every 8th row only.
__global__ __launch_bounds__(64) void k_gather_prev(const Chunk *__restrict__ chunks, const SweepItem *__restrict__ items, int S, const double *__restrict__ f,
                                                      const double *__restrict__ a0, double *__restrict__ prevx, int ck)
{
	const SweepItem it = items[blockIdx.x];
	const Chunk c = chunks[it.first];
	const int ws = max(1, c.lo - c.wf);
	const bool have = ws > 1 && !(ck && ((ws - 1) & 7)); // (a checkpoint table holds the rows p % 8 == 0)
	for (int k = threadIdx.x; k < S; k += 64) {
		double v = a0[k];
		if (have) { const double t = f[(c.off + ws - 2) * S + k]; v = (t > 0.0 && t < 1e300) ? t : v; } // (a row nobody wrote, a NaN: the stationary value)
		prevx[(int64_t)it.first * S + k] = v;
	}
}
void launch_gather_prev(const EstepLaunch &p, hipStream_t st, int first, int n_items, double *prevx)
{
	if (n_items <= 0) return;
	hipLaunchKernelGGL(k_gather_prev, dim3(n_items), dim3(64), 0, st, p.d_chunks, (const SweepItem *)p.d_items_f + first, p.ns, p.d_f, p.d_a0, prevx, p.ckpt);
}

// Plan time: which tiles consist of missing data only (symbol 2 at every position lo..hi)?  Real .psmcfa files carry runs of 1e4 .. 3e5 `N`
// bins (centromeres, assembly gaps: utils/fq2psmcfa.c:114-127); inside such a run the chain forgets at the rate of the transition matrix's
// second eigenvalue alone -- 0.99995 for a human-like model: 5e5 bins to 1e-12 -- so no speculative warm-up ever works there and every tile
// of the run hangs on the vector at its start.  The planner glues them at once, and their transfer matrix (a^T to the tile length,
// whatever the tile) is computed ONCE per direction (api_fast.hip build_items).
__global__ __launch_bounds__(64) void k_tile_allmiss(const uint8_t *__restrict__ obs, const Chunk *__restrict__ chunks, int *__restrict__ flags)
{
	const Chunk c = chunks[blockIdx.x];
	const uint8_t *o = obs + c.off;
	int bad = 0;
	for (int p = c.lo + (int)threadIdx.x; p <= c.hi; p += 64) bad |= ((int)o[p - 1] & 3) != 2;
	const int any_bad = __any(bad);
	if (threadIdx.x == 0) flags[blockIdx.x] = any_bad ? 0 : 1;
}
int launch_tile_allmiss(hipStream_t st, const uint8_t *d_obs, const Chunk *d_chunks, int n_chunks, int *d_flags)
{
	if (n_chunks <= 0) return 0;
	hipLaunchKernelGGL(k_tile_allmiss, dim3(n_chunks), dim3(64), 0, st, d_obs, d_chunks, d_flags);
	return (int)hipGetLastError();
}

void launch_kchain(const EstepLaunch &p, hipStream_t st_cols, hipStream_t st_chain, hipEvent_t ev_cols)
{
	if (p.n_kc <= 0) return;
	if (p.ns == 128) // 65..128 states: unit vectors as sweep tiles (the column-per-lane kernel's chain path ends later there)
		hipLaunchKernelGGL(k_kcol_struct<8>, dim3(p.n_kuniq * 32), dim3(64), 0, st_cols, p.d_sp, p.d_e, p.d_obs, p.d_chunks,
		                   (const KcTile *)p.d_kc, p.d_kuniq, p.d_Kcol, p.d_Kexp);
	else
		hipLaunchKernelGGL(k_kcol2_struct<2>, dim3(p.n_kuniq * p.kc_sub), dim3(128), 0, st_cols, p.d_kcc, p.d_obs, p.d_chunks, (const KcTile *)p.d_kc, p.d_kuniq,
		                   p.d_Kcol, p.d_Kexp, p.kc_sub, p.kcol_prio);
	if (st_cols != st_chain) { (void)hipEventRecord(ev_cols, st_cols); (void)hipStreamWaitEvent(st_chain, ev_cols, 0); }
	if (p.ns == 128)
		hipLaunchKernelGGL(k_kchain_struct<2>, dim3(p.n_chain_f + p.n_chain_b), dim3(512), 0, st_chain, (const KcRun *)p.d_kruns, p.n_chain_f, p.d_kslot,
		                   p.d_Kcol, p.d_Kexp, p.d_sp, p.d_e, p.d_obs, p.d_chunks, p.d_entry, p.d_bentry, p.kc_sub);
	else
		hipLaunchKernelGGL(k_kchain_struct<1>, dim3(p.n_chain_f + p.n_chain_b), dim3(128), 0, st_chain, (const KcRun *)p.d_kruns, p.n_chain_f, p.d_kslot,
		                   p.d_Kcol, p.d_Kexp, p.d_sp, p.d_e, p.d_obs, p.d_chunks, p.d_entry, p.d_bentry, p.kc_sub);
	PSMC_DBG("launch_kchain", p.n_kc, p.n_chain_f, p.n_chain_b);
}
int walk_blocks(const EstepLaunch &p) { return p.ns == 64 ? p.n_wl_f + p.n_wl_b : (p.n_wl_f + 3) / 4 + (p.n_wl_b + 3) / 4; }
// walks over the glued runs: boundary vectors only.  64 states: one wave per run, one state per lane; 65..128: four runs per wave
void launch_walks(const EstepLaunch &p, hipStream_t st)
{
	if (p.n_wl_f + p.n_wl_b <= 0) return;
	if (p.ns == 64) {
		hipLaunchKernelGGL(k_walk1_struct, dim3(p.n_wl_f + p.n_wl_b), dim3(64), 0, st, p.d_sp, p.d_e, p.d_a0, p.d_obs, p.d_chunks,
		                   (const SweepItem *)p.d_wl_f, p.n_wl_f, (const SweepItem *)p.d_wl_b, p.n_wl_b, p.warmup,
		                   p.tile_len, p.d_entry, p.d_bentry, p.d_bexit, p.d_gate);
		PSMC_DBG("launch_walks", p.n_wl_f, p.n_wl_b, 0);
		return;
	}
	hipLaunchKernelGGL(k_walk_struct<8>, dim3((p.n_wl_f + 3) / 4 + (p.n_wl_b + 3) / 4), dim3(64), 0, st, p.d_sp, p.d_e, p.d_a0, p.d_obs, p.d_chunks,
	                   (const SweepItem *)p.d_wl_f, p.n_wl_f, (const SweepItem *)p.d_wl_b, p.n_wl_b, p.warmup, p.tile_len,
	                   p.d_entry, p.d_bentry, p.d_bexit, p.d_gate);
	PSMC_DBG("launch_walks (four per wave)", p.n_wl_f, p.n_wl_b, 0);
}

} // namespace psmc

#ifdef PSMC_TRACE_SWEEP
// debug build only: which = 0 forward bulk sweep, 1 backward warm-up pass, 2 walks; out = 4 * n stamps
extern "C" int psmc_hip_debug_trace(int which, unsigned long long *out, int n)
{
	if (n > 16384) n = 16384;
	return (int)(which == 0 ? hipMemcpyFromSymbol(out, HIP_SYMBOL(psmc::g_trace_f), sizeof(unsigned long long) * 4 * n)
	             : which == 1 ? hipMemcpyFromSymbol(out, HIP_SYMBOL(psmc::g_trace_b), sizeof(unsigned long long) * 4 * n)
	                          : hipMemcpyFromSymbol(out, HIP_SYMBOL(psmc::g_trace_w), sizeof(unsigned long long) * 4 * n));
}
#endif
