// wave_prims.h -- gfx950 (CDNA4) wave64 cross-lane primitives used by the
// PSMC E-step kernels.  Device-only, header-only.
//
// Layout vocabulary: a wave is 64 lanes = 4 DPP "rows" of 16 lanes.
//   natural     : lane i holds x[i]                          (state k = lane)
//   replicated  : r[j], j=0..3; lane i holds x[16*j + (i&15)] in EVERY row
// The 64-term dot products of the forward/backward recursions are evaluated
// with `row_newbcast` DPP (the only DPP mode CDNA4 allows on 64-bit ALU ops):
// lane i of row R reads lane 16*R+N of the source register, so with the state
// vector held in replicated layout every lane can read x[16*j+N] for free as an
// operand of v_fmac_f64_dpp (fast mode) or v_mov_b64_dpp (exact mode).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace psmc {

typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned f64_lo(double x) { return (unsigned)__builtin_bit_cast(unsigned long long, x); }
__device__ __forceinline__ unsigned f64_hi(double x) { return (unsigned)(__builtin_bit_cast(unsigned long long, x) >> 32); }
__device__ __forceinline__ double f64_mk(unsigned lo, unsigned hi) {
	return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | (unsigned long long)lo);
}

// lane i <- v[(i & ~15) + N]   (compiler-visible: hazards handled by hipcc)
template <int N> __device__ __forceinline__ double bcast16(double v) {
	long long s = __builtin_bit_cast(long long, v);
	long long r = __builtin_amdgcn_update_dpp(s, s, 0x150 + N, 0xf, 0xf, true); // row_newbcast:N
	return __builtin_bit_cast(double, r);
}

// generic 64-bit DPP move (split into two v_mov_b32_dpp by the compiler)
template <int CTRL> __device__ __forceinline__ double dpp_mov(double v) {
	long long s = __builtin_bit_cast(long long, v);
	long long r = __builtin_amdgcn_update_dpp(s, s, CTRL, 0xf, 0xf, true);
	return __builtin_bit_cast(double, r);
}

// acc = fma(bcast16<N>(r), m, acc) in ONE instruction (fast mode only).
// hipcc does not pad hazards inside asm: callers must have passed r through
// dpp_guard() after the last VALU write of r (VALU write -> DPP read: 2 states).
template <int N> __device__ __forceinline__ void fmac_bcast(double &acc, double r, double m) {
	asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
	    : "+v"(acc) : "v"(r), "v"(m), "n"(N));
}
__device__ __forceinline__ void dpp_guard(double (&r)[4]) {
	asm volatile("s_nop 1" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]));
}

// natural -> replicated, via the LDS crossbar (no LDS memory).  Reference
// implementation: semantics of ds_bpermute are unambiguous.
__device__ __forceinline__ void rep_rows_bperm(double x, double (&r)[4]) {
	const int lane = (int)(threadIdx.x & 63);
	const unsigned lo = f64_lo(x), hi = f64_hi(x);
#pragma unroll
	for (int j = 0; j < 4; ++j) {
		const int addr = (16 * j + (lane & 15)) * 4;
		r[j] = f64_mk((unsigned)__builtin_amdgcn_ds_bpermute(addr, (int)lo),
		              (unsigned)__builtin_amdgcn_ds_bpermute(addr, (int)hi));
	}
}

// natural -> replicated with the gfx950 v_permlane16_swap / v_permlane32_swap
// pair (register-only, ~14 VALU issues, no LDS latency).
//   permlane16_swap(A,B): odd rows of A <-> even rows of B
//   permlane32_swap(A,B): rows 2,3 of A <-> rows 0,1 of B
// (x0,x1,x2,x3 = the four rows of x)
//   (P,Q)   = swap16(x,x)  -> P=(x0,x0,x2,x2)  Q=(x1,x1,x3,x3)
//   (P0,P2) = swap32(P,P)  -> (x0 x4 rows), (x2 x4 rows);  same for Q -> x1, x3
__device__ __forceinline__ void rep_rows_swap(double x, double (&r)[4]) {
	const unsigned lo = f64_lo(x), hi = f64_hi(x);
	const u32x2_t pl = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
	const u32x2_t ph = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
	const u32x2_t al = __builtin_amdgcn_permlane32_swap(pl.x, pl.x, false, false);
	const u32x2_t ah = __builtin_amdgcn_permlane32_swap(ph.x, ph.x, false, false);
	const u32x2_t bl = __builtin_amdgcn_permlane32_swap(pl.y, pl.y, false, false);
	const u32x2_t bh = __builtin_amdgcn_permlane32_swap(ph.y, ph.y, false, false);
	r[0] = f64_mk(al.x, ah.x); r[2] = f64_mk(al.y, ah.y);
	r[1] = f64_mk(bl.x, bh.x); r[3] = f64_mk(bl.y, bh.y);
}

template <int IMPL> __device__ __forceinline__ void rep_rows(double x, double (&r)[4]) {
	if (IMPL == 0) rep_rows_bperm(x, r); else rep_rows_swap(x, r);
}

// Sum over all 64 states given the replicated form; every lane gets the total.
// Tree order (fast mode only): 3 adds across the row groups, then a 4-level
// butterfly inside the 16-lane row with 32-bit DPP moves.
__device__ __forceinline__ double wave_sum_rep(const double (&r)[4]) {
	double t = (r[0] + r[1]) + (r[2] + r[3]);
	t = t + dpp_mov<0xB1>(t);  // quad_perm:[1,0,3,2]
	t = t + dpp_mov<0x4E>(t);  // quad_perm:[2,3,0,1]
	t = t + dpp_mov<0x124>(t); // row_ror:4
	t = t + dpp_mov<0x128>(t); // row_ror:8
	return t;
}

// Strict left-to-right sum x[0]+x[1]+...+x[63] starting from 0.0 (exact mode:
// reproduces `for (k...) sum += v[k]` of the reference bit for bit); every
// lane gets the total.
// One instruction per term: v_fmac_f64_dpp with the multiplier 1.0.  fma(x, 1.0, s) rounds the exact value x*1.0 + s = x + s
// once -- it IS the IEEE sum s + x, bit for bit, for every input (the one place an fma appears in the exact kernels; the
// products of the recursions keep their own rounding).  v_add_f64 has no DPP form on gfx950, so the alternative is a
// v_mov_b64_dpp per term: 64 more vector instructions per position of a sweep that is bound by their issue.
// the sixteen lanes of one row group in order, as ONE asm block (the compiler pads a wait state around every block)
#define PSMC_FB(N) "v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #N " row_mask:0xf bank_mask:0xf\n\t"
__device__ __forceinline__ void add_bcast16(double &s, double r, double one) {
	asm(PSMC_FB(0) PSMC_FB(1) PSMC_FB(2) PSMC_FB(3) PSMC_FB(4) PSMC_FB(5) PSMC_FB(6) PSMC_FB(7)
	    PSMC_FB(8) PSMC_FB(9) PSMC_FB(10) PSMC_FB(11) PSMC_FB(12) PSMC_FB(13) PSMC_FB(14) PSMC_FB(15)
	    : "+v"(s) : "v"(r), "v"(one));
}
#define PSMC_SEQ16(R) add_bcast16(s, R, one);
__device__ __forceinline__ double seq_sum_rep(const double (&r)[4]) {
	double q[4] = {r[0], r[1], r[2], r[3]};
	dpp_guard(q); // the asm below is invisible to the compiler's hazard padding (VALU write -> DPP read)
	double s = 0.0, one = 1.0;
	PSMC_SEQ16(q[0]) PSMC_SEQ16(q[1]) PSMC_SEQ16(q[2]) PSMC_SEQ16(q[3])
	return s;
}

// exact-order dot product: ((x0*m0 + x1*m1) + x2*m2) + ... , products rounded
// separately (v_mul_f64 then v_add_f64, never an fma), first term 0.0 + p0.
// Sixteen terms of one row group as ONE asm block: a wave alone on its SIMD issues one instruction of ANY kind every four
// cycles, so the exact sweeps pay for every s_nop as for a multiply.  Left to itself hipcc funnels the sixteen broadcasts
// through one temporary and pads a wait state before each (a DPP move reads its destination as `old`: VALU write -> DPP
// read); here two temporaries alternate, so the register a v_mov_b64_dpp overwrites was written four instructions earlier.
// Callers pass R through dpp_guard() after its last VALU write.
// The three instructions of a term are spread over three steps -- step n broadcasts x_n, multiplies term n-1 and adds term n-2 --
// so that no instruction reads what the one before it wrote; three temporaries rotate (term n uses temporary n % 3: the
// register a v_mov_b64_dpp overwrites was last written seven instructions earlier).
#define PSMC_XMOV(N, T) "v_mov_b64_dpp " T ", %4 row_newbcast:" #N " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define PSMC_XMUL(T, M) "v_mul_f64 " T ", " T ", " M "\n\t"
#define PSMC_XADD(T) "v_add_f64 %0, %0, " T "\n\t"
#define PSMC_XSTEP(N, TN, TM, MM, TA) PSMC_XMOV(N, TN) PSMC_XMUL(TM, MM) PSMC_XADD(TA)
__device__ __forceinline__ void xdot16(double &acc, double R, double m0, double m1, double m2, double m3, double m4, double m5,
                                       double m6, double m7, double m8, double m9, double m10, double m11, double m12,
                                       double m13, double m14, double m15) {
	double t0, t1, t2;
	asm(PSMC_XMOV(0, "%1") PSMC_XMOV(1, "%2") PSMC_XMUL("%1", "%5")
	    PSMC_XSTEP(2, "%3", "%2", "%6", "%1") PSMC_XSTEP(3, "%1", "%3", "%7", "%2") PSMC_XSTEP(4, "%2", "%1", "%8", "%3")
	    PSMC_XSTEP(5, "%3", "%2", "%9", "%1") PSMC_XSTEP(6, "%1", "%3", "%10", "%2") PSMC_XSTEP(7, "%2", "%1", "%11", "%3")
	    PSMC_XSTEP(8, "%3", "%2", "%12", "%1") PSMC_XSTEP(9, "%1", "%3", "%13", "%2") PSMC_XSTEP(10, "%2", "%1", "%14", "%3")
	    PSMC_XSTEP(11, "%3", "%2", "%15", "%1") PSMC_XSTEP(12, "%1", "%3", "%16", "%2") PSMC_XSTEP(13, "%2", "%1", "%17", "%3")
	    PSMC_XSTEP(14, "%3", "%2", "%18", "%1") PSMC_XSTEP(15, "%1", "%3", "%19", "%2")
	    PSMC_XMUL("%1", "%20") PSMC_XADD("%3") PSMC_XADD("%1")
	    : "+v"(acc), "=&v"(t0), "=&v"(t1), "=&v"(t2)
	    : "v"(R), "v"(m0), "v"(m1), "v"(m2), "v"(m3), "v"(m4), "v"(m5), "v"(m6), "v"(m7), "v"(m8), "v"(m9), "v"(m10), "v"(m11),
	      "v"(m12), "v"(m13), "v"(m14), "v"(m15));
}
#define PSMC_XDOT16(R, M, O)                                                                                                   \
	xdot16(acc, R, M[(O) + 0], M[(O) + 1], M[(O) + 2], M[(O) + 3], M[(O) + 4], M[(O) + 5], M[(O) + 6], M[(O) + 7], M[(O) + 8], \
	       M[(O) + 9], M[(O) + 10], M[(O) + 11], M[(O) + 12], M[(O) + 13], M[(O) + 14], M[(O) + 15]);
__device__ __forceinline__ double xdot64_guarded(const double (&q)[4], const double (&m)[64]) { // q has passed dpp_guard()
	double acc = 0.0;
	PSMC_XDOT16(q[0], m, 0) PSMC_XDOT16(q[1], m, 16) PSMC_XDOT16(q[2], m, 32) PSMC_XDOT16(q[3], m, 48)
	return acc;
}
__device__ __forceinline__ double xdot64(const double (&r)[4], const double (&m)[64]) {
	double q[4] = {r[0], r[1], r[2], r[3]};
	dpp_guard(q);
	return xdot64_guarded(q, m);
}

// fast dot product: 4 independent FMA chains (one per row group) interleaved so
// that consecutive FMAs of one chain are 4 issues apart, then a 2-level add.
#define PSMC_FDOT4(N)                                                          \
	fmac_bcast<N>(c0, r[0], m[N]);      fmac_bcast<N>(c1, r[1], m[16 + N]);       \
	fmac_bcast<N>(c2, r[2], m[32 + N]); fmac_bcast<N>(c3, r[3], m[48 + N]);
__device__ __forceinline__ double fdot64(const double (&r)[4], const double (&m)[64]) {
	double c0 = 0.0, c1 = 0.0, c2 = 0.0, c3 = 0.0;
	PSMC_FDOT4(0) PSMC_FDOT4(1) PSMC_FDOT4(2) PSMC_FDOT4(3) PSMC_FDOT4(4) PSMC_FDOT4(5)
	PSMC_FDOT4(6) PSMC_FDOT4(7) PSMC_FDOT4(8) PSMC_FDOT4(9) PSMC_FDOT4(10) PSMC_FDOT4(11)
	PSMC_FDOT4(12) PSMC_FDOT4(13) PSMC_FDOT4(14) PSMC_FDOT4(15)
	return (c0 + c1) + (c2 + c3);
}

// ---- natural-layout matvec (fast mode) -----------------------------------------------
// y[k] = sum_l x[l] * M[l][k] with x AND y in natural layout and no replication step:
// lane (r,m) (row r = lane>>4, m = lane&15) accumulates, for each of the four output groups
// j, the partial sum over ITS OWN input block r:  P_j = sum_N x[16r+N] * A[16j+N],
// A[16j+N] = M[16r+N][16j+m]  (row_newbcast:N reads x[16r+N] straight from the natural
// register).  The four partials are then transposed-and-reduced across rows with the
// gfx950 permlane swaps:
//   swap16(P0,P1): P0=(P0.r0,P1.r0,P0.r2,P1.r2) P1=(P0.r1,P1.r1,P0.r3,P1.r3); S01=P0+P1
//   swap16(P2,P3) likewise -> S23;  swap32(S01,S23): rows 2,3 of S01 <-> rows 0,1 of S23;
//   y = S01+S23 has row j = sum_r P_j.r = y[16j+m]: natural layout again.
__device__ __forceinline__ void swap16_f64(double &a, double &b) {
	const u32x2_t l = __builtin_amdgcn_permlane16_swap(f64_lo(a), f64_lo(b), false, false);
	const u32x2_t h = __builtin_amdgcn_permlane16_swap(f64_hi(a), f64_hi(b), false, false);
	a = f64_mk(l.x, h.x); b = f64_mk(l.y, h.y);
}
__device__ __forceinline__ void swap32_f64(double &a, double &b) {
	const u32x2_t l = __builtin_amdgcn_permlane32_swap(f64_lo(a), f64_lo(b), false, false);
	const u32x2_t h = __builtin_amdgcn_permlane32_swap(f64_hi(a), f64_hi(b), false, false);
	a = f64_mk(l.x, h.x); b = f64_mk(l.y, h.y);
}
#define PSMC_NDOT4(N)                                                  \
	fmac_bcast<N>(p0, x, A[N]);      fmac_bcast<N>(p1, x, A[16 + N]);      \
	fmac_bcast<N>(p2, x, A[32 + N]); fmac_bcast<N>(p3, x, A[48 + N]);
__device__ __forceinline__ double matvec64_nat(double x, const double (&A)[64]) {
	double p0 = 0.0, p1 = 0.0, p2 = 0.0, p3 = 0.0;
	asm volatile("s_nop 1" : "+v"(x)); // VALU write of x -> DPP read: 2 wait states
	PSMC_NDOT4(0) PSMC_NDOT4(1) PSMC_NDOT4(2) PSMC_NDOT4(3) PSMC_NDOT4(4) PSMC_NDOT4(5)
	PSMC_NDOT4(6) PSMC_NDOT4(7) PSMC_NDOT4(8) PSMC_NDOT4(9) PSMC_NDOT4(10) PSMC_NDOT4(11)
	PSMC_NDOT4(12) PSMC_NDOT4(13) PSMC_NDOT4(14) PSMC_NDOT4(15)
	swap16_f64(p0, p1);
	swap16_f64(p2, p3);
	double s01 = p0 + p1, s23 = p2 + p3;
	swap32_f64(s01, s23);
	return s01 + s23;
}
// register image of a 64x64 matrix M (row-major, M[l*64+k]) for matvec64_nat
__device__ __forceinline__ void load_nat_matrix(const double *__restrict__ M, int lane, double (&A)[64]) {
	const int r = lane >> 4, m = lane & 15;
#pragma unroll
	for (int j = 0; j < 4; ++j)
#pragma unroll
		for (int N = 0; N < 16; ++N) A[16 * j + N] = M[(16 * r + N) * 64 + 16 * j + m];
}
// sum over all 64 lanes of a natural-layout value; every lane gets the total (tree order)
__device__ __forceinline__ double wave_sum_nat(double x) {
	double t = x;
	t = t + dpp_mov<0xB1>(t);  // quad_perm:[1,0,3,2]
	t = t + dpp_mov<0x4E>(t);  // quad_perm:[2,3,0,1]
	t = t + dpp_mov<0x124>(t); // row_ror:4
	t = t + dpp_mov<0x128>(t); // row_ror:8   -> row sums
	double c = t;
	swap16_f64(t, c);
	t = t + c;                 // (t0+t1, t0+t1, t2+t3, t2+t3)
	c = t;
	swap32_f64(t, c);
	return t + c;
}

// lane 0's value in every lane (tree reductions agree between lanes only to an ulp)
__device__ __forceinline__ double first_lane_f64(double v) {
	return f64_mk((unsigned)__builtin_amdgcn_readfirstlane((int)f64_lo(v)), (unsigned)__builtin_amdgcn_readfirstlane((int)f64_hi(v)));
}

// wave-uniform double out of a VGPR lane (lane index may be a runtime scalar)
__device__ __forceinline__ double readlane_f64(double v, int lane) {
	return f64_mk((unsigned)__builtin_amdgcn_readlane((int)f64_lo(v), lane),
	              (unsigned)__builtin_amdgcn_readlane((int)f64_hi(v), lane));
}

} // namespace psmc
