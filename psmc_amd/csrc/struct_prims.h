// struct_prims.h -- row-local (16-lane) scans and the O(N) structured matrix-vector step
// shared by estep_struct.hip and the device self-test.  See estep_struct.hip for the algebra.
#pragma once
#include <hip/hip_runtime.h>
#include "wave_prims.h"

namespace psmc {

typedef double d2v_t __attribute__((ext_vector_type(2)));

// v_mov_b32_dpp pair; lanes without a source lane in their row read 0
template <int CTRL> __device__ __forceinline__ double dpp_z(double v) {
	long long s = __builtin_bit_cast(long long, v);
	long long r = __builtin_amdgcn_update_dpp(0LL, s, CTRL, 0xf, 0xf, true);
	return __builtin_bit_cast(double, r);
}
// sum over the lanes m' > m of the same 16-lane row
__device__ __forceinline__ double row_excl_suffix(double t) {
	t = t + dpp_z<0x101>(t); // row_shl:1
	t = t + dpp_z<0x102>(t);
	t = t + dpp_z<0x104>(t);
	t = t + dpp_z<0x108>(t);
	return dpp_z<0x101>(t);
}
// sum over the lanes m' < m of the same 16-lane row
__device__ __forceinline__ double row_excl_prefix(double t) {
	t = t + dpp_z<0x111>(t); // row_shr:1
	t = t + dpp_z<0x112>(t);
	t = t + dpp_z<0x114>(t);
	t = t + dpp_z<0x118>(t);
	return dpp_z<0x111>(t);
}
// sum / max over the 16 lanes of a row, identical in every lane of the row
__device__ __forceinline__ double row_sum16(double t) {
	t = t + dpp_mov<0xB1>(t);  // quad_perm:[1,0,3,2]
	t = t + dpp_mov<0x4E>(t);  // quad_perm:[2,3,0,1]
	t = t + dpp_mov<0x124>(t); // row_ror:4
	t = t + dpp_mov<0x128>(t); // row_ror:8
	return bcast16<0>(t);       // butterfly orders differ by an ulp between lanes: take lane 0's
}
__device__ __forceinline__ double row_max16(double t) {
	t = fmax(t, dpp_mov<0xB1>(t));
	t = fmax(t, dpp_mov<0x4E>(t));
	t = fmax(t, dpp_mov<0x124>(t));
	t = fmax(t, dpp_mov<0x128>(t));
	return t;
}
// inclusive prefix / suffix sums over all 64 lanes (one state per lane): the row scans above plus the
// totals of the other rows, read with v_readlane and added under per-lane 0/1 masks
// (mlo[r] = 1 for lanes of rows > r, mhi[r] = 1 for lanes of rows < r+1 ... see wave_scan_masks)
struct WaveScanMasks { double pre[3], suf[3]; };
__device__ __forceinline__ WaveScanMasks wave_scan_masks(int lane) {
	const int row = lane >> 4;
	WaveScanMasks m;
	m.pre[0] = row > 0 ? 1.0 : 0.0; m.pre[1] = row > 1 ? 1.0 : 0.0; m.pre[2] = row > 2 ? 1.0 : 0.0; // add total of row 0 / 1 / 2
	m.suf[0] = row < 1 ? 1.0 : 0.0; m.suf[1] = row < 2 ? 1.0 : 0.0; m.suf[2] = row < 3 ? 1.0 : 0.0; // add total of row 1 / 2 / 3
	return m;
}
__device__ __forceinline__ double wave_prefix_incl(double t, const WaveScanMasks &m) {
	t = t + dpp_z<0x111>(t); t = t + dpp_z<0x112>(t); t = t + dpp_z<0x114>(t); t = t + dpp_z<0x118>(t);
	const double r0 = readlane_f64(t, 15), r1 = readlane_f64(t, 31), r2 = readlane_f64(t, 47); // row totals
	return __builtin_fma(m.pre[2], r2, __builtin_fma(m.pre[1], r1, __builtin_fma(m.pre[0], r0, t)));
}
// the same with the gfx9 row_bcast DPP modes: lane 15 of a row into the next row (rows 1,3), then lane 31
// into rows 2,3 -- no scalar round trip
template <int CTRL, int ROWMASK> __device__ __forceinline__ double dpp_zm(double v) {
	long long s = __builtin_bit_cast(long long, v);
	long long r = __builtin_amdgcn_update_dpp(0LL, s, CTRL, ROWMASK, 0xf, false);
	return __builtin_bit_cast(double, r);
}
__device__ __forceinline__ double wave_prefix_incl_bc(double t) {
	t = t + dpp_z<0x111>(t); t = t + dpp_z<0x112>(t); t = t + dpp_z<0x114>(t); t = t + dpp_z<0x118>(t);
	t = t + dpp_zm<0x142, 0xA>(t); // row_bcast:15
	t = t + dpp_zm<0x143, 0xC>(t); // row_bcast:31
	return t;
}
__device__ __forceinline__ double wave_suffix_incl(double t, const WaveScanMasks &m) {
	t = t + dpp_z<0x101>(t); t = t + dpp_z<0x102>(t); t = t + dpp_z<0x104>(t); t = t + dpp_z<0x108>(t);
	const double r1 = readlane_f64(t, 16), r2 = readlane_f64(t, 32), r3 = readlane_f64(t, 48);
	return __builtin_fma(m.suf[0], r1, __builtin_fma(m.suf[1], r2, __builtin_fma(m.suf[2], r3, t)));
}
// one-state-per-lane form of struct_step: x <- wS.SUF(x.mS) + wP.PRE(x.mP) + dd.x
struct StructPar1 { double mS, wS, mP, wP, dd; };
__device__ __forceinline__ double struct_step1(const StructPar1 &c, double x, const WaveScanMasks &m) {
	const double SI = wave_suffix_incl(x * c.mS, m), PI = wave_prefix_incl_bc(x * c.mP);
	return __builtin_fma(c.wS, SI, __builtin_fma(c.wP, PI, c.dd * x));
}

// 2^-E for x = 1.m * 2^E (x positive, normal): x * pow2_rcp(x) lies in [1, 2).  The forward sweep normalises with THIS
// instead of 1/sum (estep_struct.hip fwd_step): multiplying by a power of two is exact, so the lag-normalised X and
// everything computed from it are the same numbers whatever the scale factors were -- which makes the back halves
// that take the forward scale factors (estep_fused.hip, estep_factored.hip) reproducible bit for bit even when they
// read a scale factor a forward repair is just rewriting (old or new: the two differ by a power of two).
__device__ __forceinline__ double pow2_rcp(double x) {
	const unsigned hi = (unsigned)((unsigned long long)__builtin_bit_cast(long long, x) >> 32);
	const unsigned r = 0x7FE00000u - (hi & 0x7FF00000u);
	return __builtin_bit_cast(double, (long long)((unsigned long long)r << 32));
}
__device__ __forceinline__ double rcp_newton(double x) {
	double r = __builtin_amdgcn_rcp(x);
	double t = __builtin_fma(-x, r, 1.0);
	r = __builtin_fma(r, t, r);
	t = __builtin_fma(-x, r, 1.0);
	r = __builtin_fma(r, t, r);
	return r;
}

// per-lane constants of one sweep direction (NPL adjacent states per lane; 16 lanes = one tile:
// NPL = 4 for up to 64 states, 8 for up to 128)
template <int NPL> struct StructParN { double mS[NPL], wS[NPL], mP[NPL], wP[NPL], dd[NPL]; };
typedef StructParN<4> StructPar;

// x <- M x for the structured M: wS.SUF(x.mS) + wP.PRE(x.mP) + dd.x
template <int NPL>
__device__ __forceinline__ void struct_step(const StructParN<NPL> &c, double (&x)[NPL])
{
	double su[NPL], pv[NPL]; // lane-local inclusive suffix sums of x.mS / prefix sums of x.mP
	su[NPL - 1] = x[NPL - 1] * c.mS[NPL - 1];
#pragma unroll
	for (int i = NPL - 2; i >= 0; --i) su[i] = __builtin_fma(x[i], c.mS[i], su[i + 1]);
	pv[0] = x[0] * c.mP[0];
#pragma unroll
	for (int i = 1; i < NPL; ++i) pv[i] = __builtin_fma(x[i], c.mP[i], pv[i - 1]);
	const double ES = row_excl_suffix(su[0]), EP = row_excl_prefix(pv[NPL - 1]);
#pragma unroll
	for (int i = 0; i < NPL; ++i) { // the lane-local part does not wait for the row scans
		const double t = __builtin_fma(c.wS[i], su[i], __builtin_fma(c.wP[i], pv[i], c.dd[i] * x[i]));
		x[i] = __builtin_fma(c.wS[i], ES, __builtin_fma(c.wP[i], EP, t));
	}
}

// ---- 8-lane groups (two per DPP row, eight tiles per wave): 64 states as 8 lanes x 8 states.  The cross-lane part of a scan
// shrinks from four levels to three and is shared by twice as many states per lane: ~13 vector instructions per tile-step instead
// of 17.  row_shr / row_shl by 1 and 2 would cross the group boundary inside a row: those levels are masked (an FMA instead of an
// add); the shift by 4 uses the DPP bank mask instead.  (Round 2 built this, round 3 removed it -- inside the noise while phase 1
// ended with the runs' path -- round 4 brought it back for the bulk sweeps, "lanes8": DESIGN.md.)
template <int CTRL, int BANKMASK> __device__ __forceinline__ double dpp_zb(double v) {
	long long s = __builtin_bit_cast(long long, v);
	long long r = __builtin_amdgcn_update_dpp(0LL, s, CTRL, 0xf, BANKMASK, false);
	return __builtin_bit_cast(double, r);
}
struct Half8Masks { double p1, p2, s1, s2; };
__device__ __forceinline__ Half8Masks half8_masks(int lane) {
	const int m = lane & 7;
	Half8Masks k; k.p1 = m >= 1 ? 1.0 : 0.0; k.p2 = m >= 2 ? 1.0 : 0.0; k.s1 = m <= 6 ? 1.0 : 0.0; k.s2 = m <= 5 ? 1.0 : 0.0;
	return k;
}
__device__ __forceinline__ double half8_excl_prefix(double t, const Half8Masks &k) {
	t = __builtin_fma(k.p1, dpp_z<0x111>(t), t);
	t = __builtin_fma(k.p2, dpp_z<0x112>(t), t);
	t = t + dpp_zb<0x114, 0xA>(t); // lanes 4..7 and 12..15 of the row only
	return k.p1 * dpp_z<0x111>(t);
}
__device__ __forceinline__ double half8_excl_suffix(double t, const Half8Masks &k) {
	t = __builtin_fma(k.s1, dpp_z<0x101>(t), t);
	t = __builtin_fma(k.s2, dpp_z<0x102>(t), t);
	t = t + dpp_zb<0x104, 0x5>(t); // lanes 0..3 and 8..11 of the row only
	return k.s1 * dpp_z<0x101>(t);
}
// sum over the 8 lanes of a group, bit-identical in all of them (every level adds the same two numbers)
__device__ __forceinline__ double half8_sum(double t) {
	t = t + dpp_mov<0xB1>(t);  // quad_perm:[1,0,3,2]
	t = t + dpp_mov<0x4E>(t);  // quad_perm:[2,3,0,1]
	t = t + dpp_mov<0x141>(t); // row_half_mirror
	return t;
}
__device__ __forceinline__ void struct_step_h8(const StructParN<8> &c, double (&x)[8], const Half8Masks &k)
{
	double su[8], pv[8];
	su[7] = x[7] * c.mS[7];
#pragma unroll
	for (int i = 6; i >= 0; --i) su[i] = __builtin_fma(x[i], c.mS[i], su[i + 1]);
	pv[0] = x[0] * c.mP[0];
#pragma unroll
	for (int i = 1; i < 8; ++i) pv[i] = __builtin_fma(x[i], c.mP[i], pv[i - 1]);
	const double ES = half8_excl_suffix(su[0], k), EP = half8_excl_prefix(pv[7], k);
#pragma unroll
	for (int i = 0; i < 8; ++i) {
		const double t = __builtin_fma(c.wS[i], su[i], __builtin_fma(c.wP[i], pv[i], c.dd[i] * x[i]));
		x[i] = __builtin_fma(c.wS[i], ES, __builtin_fma(c.wP[i], EP, t));
	}
}

template <int NPL> __device__ __forceinline__ void loadN(const double *p, double (&v)[NPL]) {
#pragma unroll
	for (int i = 0; i < NPL / 2; ++i) { const d2v_t a = reinterpret_cast<const d2v_t *>(p)[i]; v[2 * i] = a.x; v[2 * i + 1] = a.y; }
}
template <int NPL> __device__ __forceinline__ void storeN(double *p, const double (&v)[NPL]) {
#pragma unroll
	for (int i = 0; i < NPL / 2; ++i) { d2v_t a; a.x = v[2 * i]; a.y = v[2 * i + 1]; reinterpret_cast<d2v_t *>(p)[i] = a; }
}
// Per-lane vectors in LDS (emission rows, constants): lane m of a 16-lane row wants its NPL adjacent states k = NPL m + i.  Stored in
// the natural order that is a ds_read_b128 with 8 NPL bytes between lanes -- a two-way (NPL = 4) or four-way (NPL = 8) bank conflict
// on every read (round 4: the 128-state counts kernel lost 15 % to it).  So a row of 16 NPL doubles is kept as cells of 16 bytes,
// cell 16 p + m = states NPL m + 2p, NPL m + 2p + 1: the sixteen lanes of a row read consecutive cells, the four rows the same ones.
#ifdef PSMC_EV_NATURAL // A/B build: the natural order of rounds 1-3
template <int NPL, int LPT = 16> __device__ __forceinline__ int ev_slot(int k) { return k; }
template <int NPL, int LPT = 16> __device__ __forceinline__ void ev_load(const double *row, int k0, double (&v)[NPL]) {
#pragma unroll
	for (int i = 0; i < NPL / 2; ++i) { const d2v_t a = reinterpret_cast<const d2v_t *>(row + k0)[i]; v[2 * i] = a.x; v[2 * i + 1] = a.y; }
}
#else
// (LPT lanes per tile: 16, or 8 in the eight-tiles-per-wave sweeps -- then a row of LPT NPL doubles is LPT cells per state pair)
template <int NPL, int LPT = 16> __device__ __forceinline__ int ev_slot(int k) { return 2 * (LPT * ((k % NPL) >> 1) + k / NPL) + (k & 1); } // index of state k in its row
template <int NPL, int LPT = 16> __device__ __forceinline__ void ev_load(const double *row, int k0, double (&v)[NPL]) {
	const d2v_t *c = reinterpret_cast<const d2v_t *>(row) + k0 / NPL;
#pragma unroll
	for (int p = 0; p < NPL / 2; ++p) { const d2v_t t = c[LPT * p]; v[2 * p] = t.x; v[2 * p + 1] = t.y; }
}
#endif
__device__ __forceinline__ void load4(const double *p, double (&v)[4]) { loadN<4>(p, v); }
__device__ __forceinline__ void store4(double *p, const double (&v)[4]) { storeN<4>(p, v); }

} // namespace psmc
