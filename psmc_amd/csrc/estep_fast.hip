// estep_fast.hip -- FAST mode of the PSMC E-step for gfx950.
//
// Same mathematics as khmm.c's hmm_forward / hmm_backward / hmm_expect
// (lh3/psmc khmm.c:145-190, 210-241, 297-324) but re-associated for the GPU:
//   * every segment is cut into tiles of `chunk` bins; a tile's sweep starts
//     `warmup` bins outside the tile from an arbitrary vector (the chain forgets
//     its start), so all tiles of all segments run concurrently, one wavefront
//     per tile, lane = hidden state;
//   * the 64-term dot products are 64 v_fmac_f64_dpp (row_newbcast operand
//     broadcast, transition column/row held in 128 VGPRs per lane);
//   * lagged normalisation: X_p = e[o_p]*(a^T X_{p-1}) / d_p with
//     d_p = sum(X_{p-1}), which takes the cross-lane reduction off the
//     sequential critical path; sum(X_p) is then the reference's s_p and
//     LL = sum_p log(sum X_p);
//   * backward uses the same divisors, B_p = a(e[o_{p+1}]*B_{p+1}) / d_p, and is
//     renormalised once per tile so that the posterior sums to one;
//   * A = a .* sum_p X_p (x) (e[o_{p+1}]*B_{p+1}) is a K=bins GEMM: it runs on
//     the FP64 matrix cores (v_mfma_f64_16x16x4_f64) or, as a cross-check, on
//     the VALU; per-wave partials are reduced in a fixed order (deterministic,
//     no atomics).
// tests/fastmodel.py is the executable numpy specification of this file.
// HBM layout: X[g*64+k] (d_f), bt[g*64+k] = e[o_p]*B_p (d_b), inv_d[g] (d_s),
// g = seg_off + p - 1.
#include <hip/hip_runtime.h>
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

// ------------------------------------------------------------------ forward
template <int REP>
__global__ __launch_bounds__(64) void k_fwd_fast(const double *__restrict__ a, const double *__restrict__ e,
                                                   const double *__restrict__ a0, const uint8_t *__restrict__ obs,
                                                   const Chunk *__restrict__ chunks, int W, double *__restrict__ f,
                                                   double *__restrict__ invd, double *__restrict__ entry,
                                                   double *__restrict__ LLpart)
{
	const int lane = threadIdx.x;
	const Chunk c = chunks[blockIdx.x];
	const uint8_t *o = obs + c.off;
	double *fo = f + c.off * 64, *io = invd + c.off;
	double col[64];
#pragma unroll
	for (int l = 0; l < 64; ++l) col[l] = a[l * 64 + lane];
	const double e0 = e[lane], e1 = e[64 + lane];
	const int ws = max(1, c.lo - W);
	double x, prod = 1.0, ll = 0.0;
	int p;
	if (ws == 1) { // true start: X_1 = a0*e[o_1], d_1 = 1 (khmm.c:171-174 without the division)
		x = a0[lane] * pick_ef((int)o[0], e0, e1);
		if (c.lo == 1) { fo[lane] = x; if (lane == 0) io[0] = 1.0; }
		p = 2;
	} else { // warm-up from the stationary prior
		x = a0[lane];
		p = ws;
	}
	int blk = (p - 1) >> 6;
	int symv = o[(blk << 6) + lane], symn = o[((blk + 1) << 6) + lane];
	for (; p <= c.hi; ++p) {
		const int idx = p - 1;
		if ((idx >> 6) != blk) { blk = idx >> 6; symv = symn; symn = o[((blk + 1) << 6) + lane]; }
		const int sym = __builtin_amdgcn_readlane(symv, idx & 63);
		double r[4];
		rep_rows<REP>(x, r);
		dpp_guard(r);
		if (p == c.lo) entry[(int64_t)blockIdx.x * 64 + lane] = x; // X_{lo-1} as warmed up here
		const double sig = wave_sum_rep(r); // = s_{p-1} of the reference
		const double inv = fast_rcp(sig);
		if (p - 1 >= c.lo) {
			prod *= sig;
			if (prod < 1e-280) { ll += log(prod); prod = 1.0; }
		}
		const double acc = fdot64(r, col);
		x = acc * (pick_ef(sym, e0, e1) * inv);
		if (p >= c.lo) {
			fo[(int64_t)idx * 64 + lane] = x;
			if (lane == 0) io[idx] = inv;
		}
	}
	{ // s_hi
		double r[4];
		rep_rows<REP>(x, r);
		prod *= wave_sum_rep(r);
		ll += log(prod);
	}
	if (lane == 0) LLpart[blockIdx.x] = ll * (double)c.mult;
}

// ------------------------------------------------------------------ backward
// cached (symbol, inv_d) for descending bin indices, one coalesced load per 64
struct DownStream {
	const uint8_t *o; const double *io; int lane, blk, symv, symn; double invv, invn;
	__device__ __forceinline__ void fetch(int b, int &sv, double &iv) const {
		const int i = (max(b, 0) << 6) + lane;
		sv = o[i]; iv = io[i];
	}
	__device__ __forceinline__ void init(const uint8_t *o_, const double *io_, int lane_, int idx) {
		o = o_; io = io_; lane = lane_; blk = idx >> 6;
		fetch(blk, symv, invv); fetch(blk - 1, symn, invn);
	}
	__device__ __forceinline__ void seek(int idx) {
		if ((idx >> 6) != blk) { blk = idx >> 6; symv = symn; invv = invn; fetch(blk - 1, symn, invn); }
	}
	__device__ __forceinline__ int sym(int idx) const { return __builtin_amdgcn_readlane(symv, idx & 63); }
	__device__ __forceinline__ double inv(int idx) const { return readlane_f64(invv, idx & 63); }
};

template <int REP>
__global__ __launch_bounds__(64) void k_bwd_fast(const double *__restrict__ aT, const double *__restrict__ e,
                                                   const uint8_t *__restrict__ obs, const Chunk *__restrict__ chunks,
                                                   int W, const double *__restrict__ f, const double *__restrict__ invd,
                                                   double *__restrict__ bt, double *__restrict__ bexit,
                                                   double *__restrict__ Epart)
{
	const int lane = threadIdx.x;
	const Chunk c = chunks[blockIdx.x];
	const int L = c.L, lo = c.lo, top = min(c.hi, L - 1);
	double E0 = 0.0, E1 = 0.0, E2 = 0.0;
	if (top >= lo) {
		const uint8_t *o = obs + c.off;
		const double *fo = f + c.off * 64, *io = invd + c.off;
		double *bto = bt + c.off * 64;
		double row[64]; // a[k][l], k = lane
#pragma unroll
		for (int l = 0; l < 64; ++l) row[l] = aT[l * 64 + lane];
		const double e0 = e[lane], e1 = e[64 + lane];
		const int q = min(c.hi + W + 1, L);                // B_q := 1
		double btn = pick_ef((int)o[q - 1], e0, e1);       // e[o_q] * B_q, natural layout
		DownStream ds;
		ds.init(o, io, lane, q - 2);

		// X rows of the owned positions, prefetched four steps ahead through x0..x3
		auto ldx = [&](int pp) -> double {
			return (pp <= top && pp >= lo) ? fo[(int64_t)(pp - 1) * 64 + lane] : 0.0;
		};
		int p = q - 1;
		double x0 = ldx(p), x1 = ldx(p - 1), x2 = ldx(p - 2), x3 = ldx(p - 3);
		// one step: consumes btn = e[o_{p+1}]*B_{p+1}, produces btn = e[o_p]*B_p
		for (; p >= lo; --p) {
			const double Xp = x0;
			x0 = x1; x1 = x2; x2 = x3; x3 = ldx(p - 4);
			const int idx = p - 1;
			ds.seek(idx);
			const int sym = ds.sym(idx);
			const double inv = ds.inv(idx);
			double r[4];
			rep_rows<REP>(btn, r);
			dpp_guard(r);
			double bnew = fdot64(r, row); // (a . e*B_{p+1})[k] = B_p[k] * d_p
			if (p <= top) {
				if (p == top) { // normalise the tile: posterior at `top` sums to one
					double rr[4];
					rep_rows<REP>(Xp * bnew, rr);
					const double kappa = 1.0 / wave_sum_rep(rr);
					bnew *= kappa; btn *= kappa;
					bto[(int64_t)top * 64 + lane] = btn; // bt[top+1]
				}
				const double v = Xp * bnew; // gamma_p(k)  (khmm.c:317 up to scaling)
				if (sym == 0) E0 += v; else if (sym == 1) E1 += v; else E2 += v;
			}
			btn = bnew * (pick_ef(sym, e0, e1) * inv);
			if (p <= top && p > lo) bto[(int64_t)idx * 64 + lane] = btn; // bt[p]
			if (p == lo) bexit[(int64_t)blockIdx.x * 64 + lane] = btn;
		}
	}
	const double m = (double)c.mult;
	double *oe = Epart + (int64_t)blockIdx.x * 192;
	oe[lane] = E0 * m; oe[64 + lane] = E1 * m; oe[128 + lane] = E2 * m;
}

// ------------------------------------------------------------------ expect
// C[k][l] += sum_p X_p[k] * bt_{p+1}[l] over the tile's positions lo..min(hi,L-1),
// split over n_sub waves.  FP64 matrix cores: D(16x16) += A(16x4) B(4x16) with
//   A[i][t] = X_{p+t}[16m+i]   (lane = 16t+i),  B[t][j] = bt_{p+t+1}[16n+j] (lane = 16t+j)
//   D[(lane>>4)+4r][lane&15] = acc[r]
__global__ __launch_bounds__(64, 2) void k_expect_mfma(const Chunk *__restrict__ chunks, int n_sub,
                                                      const double *__restrict__ f, const double *__restrict__ bt,
                                                      double *__restrict__ Cpart)
{
	const int lane = threadIdx.x, t = lane >> 4, i = lane & 15;
	const Chunk c = chunks[blockIdx.x / n_sub];
	const int sub = blockIdx.x % n_sub;
	const int top = min(c.hi, c.L - 1), n = top - c.lo + 1;
	const int per = n > 0 ? ((((n + n_sub - 1) / n_sub) + 3) & ~3) : 0;
	const int p0 = c.lo + sub * per, p1 = min(top, p0 + per - 1);
	const double *fo = f + c.off * 64 + i, *bo = bt + c.off * 64 + i;
	d4_t acc[4][4];
#pragma unroll
	for (int m = 0; m < 4; ++m)
#pragma unroll
		for (int nn = 0; nn < 4; ++nn) acc[m][nn] = (d4_t){0.0, 0.0, 0.0, 0.0};
	auto load = [&](int p, double (&FA)[4], double (&BM)[4]) {
		const int pp = p + t;
		const bool ok = pp <= p1;
		const int64_t idx = (int64_t)(ok ? pp : p1) - 1;
		const double *fr = fo + idx * 64, *br = bo + (idx + 1) * 64;
#pragma unroll
		for (int m = 0; m < 4; ++m) { FA[m] = ok ? fr[16 * m] : 0.0; BM[m] = br[16 * m]; }
	};
	if (p0 <= p1) {
		double FA[4], BM[4];
		load(p0, FA, BM);
		for (int p = p0; p <= p1; p += 4) {
			double FN[4] = {0, 0, 0, 0}, BN[4] = {0, 0, 0, 0};
			if (p + 4 <= p1) load(p + 4, FN, BN);
#pragma unroll
			for (int m = 0; m < 4; ++m)
#pragma unroll
				for (int nn = 0; nn < 4; ++nn)
					acc[m][nn] = __builtin_amdgcn_mfma_f64_16x16x4f64(FA[m], BM[nn], acc[m][nn], 0, 0, 0);
#pragma unroll
			for (int m = 0; m < 4; ++m) { FA[m] = FN[m]; BM[m] = BN[m]; }
		}
	}
	const double mult = (double)c.mult;
	double *out = Cpart + (int64_t)blockIdx.x * 4096;
#pragma unroll
	for (int m = 0; m < 4; ++m)
#pragma unroll
		for (int nn = 0; nn < 4; ++nn)
#pragma unroll
			for (int r = 0; r < 4; ++r) out[(16 * m + t + 4 * r) * 64 + 16 * nn + i] = acc[m][nn][r] * mult;
}

// VALU cross-check of the above: lane = k, 64 accumulators C[k][0..63] per lane.
#define PSMC_OUTER4(N)                                                    \
	fmac_bcast<N>(C[N], r[0], X);      fmac_bcast<N>(C[16 + N], r[1], X);    \
	fmac_bcast<N>(C[32 + N], r[2], X); fmac_bcast<N>(C[48 + N], r[3], X);
__global__ __launch_bounds__(64) void k_expect_valu(const Chunk *__restrict__ chunks, int n_sub,
                                                      const double *__restrict__ f, const double *__restrict__ bt,
                                                      double *__restrict__ Cpart)
{
	const int lane = threadIdx.x;
	const Chunk c = chunks[blockIdx.x / n_sub];
	const int sub = blockIdx.x % n_sub;
	const int top = min(c.hi, c.L - 1), n = top - c.lo + 1;
	const int per = n > 0 ? ((((n + n_sub - 1) / n_sub) + 3) & ~3) : 0;
	const int p0 = c.lo + sub * per, p1 = min(top, p0 + per - 1);
	const double *fo = f + c.off * 64, *bo = bt + c.off * 64;
	double C[64];
#pragma unroll
	for (int l = 0; l < 64; ++l) C[l] = 0.0;
	for (int p = p0; p <= p1; ++p) {
		const double X = fo[(int64_t)(p - 1) * 64 + lane];
		double r[4];
#pragma unroll
		for (int j = 0; j < 4; ++j) r[j] = bo[(int64_t)p * 64 + 16 * j + (lane & 15)]; // replicated load of bt[p+1]
		dpp_guard(r);
		PSMC_OUTER4(0) PSMC_OUTER4(1) PSMC_OUTER4(2) PSMC_OUTER4(3) PSMC_OUTER4(4) PSMC_OUTER4(5)
		PSMC_OUTER4(6) PSMC_OUTER4(7) PSMC_OUTER4(8) PSMC_OUTER4(9) PSMC_OUTER4(10) PSMC_OUTER4(11)
		PSMC_OUTER4(12) PSMC_OUTER4(13) PSMC_OUTER4(14) PSMC_OUTER4(15)
	}
	const double mult = (double)c.mult;
	double *out = Cpart + (int64_t)blockIdx.x * 4096 + lane * 64;
#pragma unroll
	for (int l = 0; l < 64; ++l) out[l] = C[l] * mult;
}

// ------------------------------------------------------------------ checks
// Largest relative mismatch between a tile's warmed-up entry vector and the
// value its neighbour computed with a full tile of history behind it.
__global__ __launch_bounds__(64) void k_warm_check(const Chunk *__restrict__ chunks, const double *__restrict__ f,
                                                     const double *__restrict__ bt, const double *__restrict__ entry,
                                                     const double *__restrict__ bexit,
                                                     unsigned long long *__restrict__ warm)
{
	const int lane = threadIdx.x;
	const Chunk c = chunks[blockIdx.x];
	if (c.lo <= 1) return;
	{
		const double w = entry[(int64_t)blockIdx.x * 64 + lane];
		const double tr = f[(c.off + c.lo - 2) * 64 + lane];
		const double num = wave_max(fabs(w - tr)), den = wave_max(fabs(tr));
		if (lane == 0) atomicMax(&warm[0], (unsigned long long)__double_as_longlong(num / den));
	}
	if (min(c.hi, c.L - 1) >= c.lo) {
		const double w = bexit[(int64_t)blockIdx.x * 64 + lane];
		const double tr = bt[(c.off + c.lo - 1) * 64 + lane];
		const double num = wave_max(fabs(w - tr)), den = wave_max(fabs(tr));
		if (lane == 0) atomicMax(&warm[1], (unsigned long long)__double_as_longlong(num / den));
	}
}

// ------------------------------------------------------------------ reduce
// Fixed-order two-stage reduction of the per-wave partials (deterministic).
__global__ __launch_bounds__(256) void k_reduce1(const double *__restrict__ Cpart, int nC,
                                                   const double *__restrict__ Epart, const double *__restrict__ LLpart,
                                                   int nchunks, double *__restrict__ stage)
{
	const int y = blockIdx.y, tid = threadIdx.x;
	double *st = stage + (int64_t)y * STATS_LEN;
	if (blockIdx.x < 16) {
		const int i = blockIdx.x * 256 + tid;
		double s = 0.0;
		for (int j = y; j < nC; j += RED_ROWS) s += Cpart[(int64_t)j * 4096 + i];
		st[i] = s;
	} else if (tid < 192) {
		double s = 0.0;
		for (int j = y; j < nchunks; j += RED_ROWS) s += Epart[(int64_t)j * 192 + tid];
		st[4096 + tid] = s;
	} else if (tid == 192) {
		double s = 0.0;
		for (int j = y; j < nchunks; j += RED_ROWS) s += LLpart[j];
		st[4096 + 192] = s;
	}
}
// Writes the final statistics UNPADDED: out = [A n*n | E 2*n | LL].
__global__ __launch_bounds__(256) void k_reduce2(const double *__restrict__ stage, const double *__restrict__ a,
                                                   double tiny_total, int n, double *__restrict__ out)
{
	const int i = blockIdx.x * 256 + threadIdx.x;
	if (i >= STATS_LEN) return;
	double s = 0.0;
	for (int y = 0; y < RED_ROWS; ++y) s += stage[(int64_t)y * STATS_LEN + i];
	if (i < 4096) { // A = a .* C + n_seg*HMM_TINY (khmm.c:305-306,316)
		const int k = i >> 6, l = i & 63;
		if (k < n && l < n) out[k * n + l] = a[i] * s + tiny_total;
	} else if (i < 4096 + 192) { // khmm.c:307-308; the missing-symbol row is dropped (khmm.c:355)
		const int b = (i - 4096) >> 6, k = (i - 4096) & 63;
		if (b < 2 && k < n) out[n * n + b * n + k] = s + tiny_total;
	} else {
		out[n * n + 2 * n] = s;
	}
}

// ------------------------------------------------------------------ launcher
int launch_fast(const EstepLaunch &p)
{
	if (p.n_chunks <= 0) return 0;
	const dim3 g(p.n_chunks), b(64);
	hipMemsetAsync(p.d_warm, 0, 2 * sizeof(unsigned long long), p.stream);
	if (p.ev[0]) hipEventRecord(p.ev[0], p.stream);
	if (p.rep_impl == 0)
		hipLaunchKernelGGL(k_fwd_fast<0>, g, b, 0, p.stream, p.d_a, p.d_e, p.d_a0, p.d_obs, p.d_chunks, p.warmup, p.d_f,
		                   p.d_s, p.d_entry, p.d_LLpart);
	else
		hipLaunchKernelGGL(k_fwd_fast<1>, g, b, 0, p.stream, p.d_a, p.d_e, p.d_a0, p.d_obs, p.d_chunks, p.warmup, p.d_f,
		                   p.d_s, p.d_entry, p.d_LLpart);
	if (p.ev[1]) hipEventRecord(p.ev[1], p.stream);
	const double *aT = p.d_aeT + 2 * 4096;
	if (p.rep_impl == 0)
		hipLaunchKernelGGL(k_bwd_fast<0>, g, b, 0, p.stream, aT, p.d_e, p.d_obs, p.d_chunks, p.warmup, p.d_f, p.d_s,
		                   p.d_b, p.d_bexit, p.d_Epart);
	else
		hipLaunchKernelGGL(k_bwd_fast<1>, g, b, 0, p.stream, aT, p.d_e, p.d_obs, p.d_chunks, p.warmup, p.d_f, p.d_s,
		                   p.d_b, p.d_bexit, p.d_Epart);
	if (p.ev[2]) hipEventRecord(p.ev[2], p.stream);
	const int nC = p.n_chunks * p.n_sub;
	if (p.expect_impl == 0)
		hipLaunchKernelGGL(k_expect_valu, dim3(nC), b, 0, p.stream, p.d_chunks, p.n_sub, p.d_f, p.d_b, p.d_Cpart);
	else
		hipLaunchKernelGGL(k_expect_mfma, dim3(nC), b, 0, p.stream, p.d_chunks, p.n_sub, p.d_f, p.d_b, p.d_Cpart);
	if (p.ev[3]) hipEventRecord(p.ev[3], p.stream);
	hipLaunchKernelGGL(k_warm_check, g, b, 0, p.stream, p.d_chunks, p.d_f, p.d_b, p.d_entry, p.d_bexit, p.d_warm);
	hipLaunchKernelGGL(k_reduce1, dim3(17, RED_ROWS), dim3(256), 0, p.stream, p.d_Cpart, nC, p.d_Epart, p.d_LLpart,
	                   p.n_chunks, p.d_stage);
	hipLaunchKernelGGL(k_reduce2, dim3((STATS_LEN + 255) / 256), dim3(256), 0, p.stream, p.d_stage, p.d_a,
	                   p.tiny_total, p.n_states, p.d_stats);
	if (p.ev[4]) hipEventRecord(p.ev[4], p.stream);
	return (int)hipGetLastError();
}

} // namespace psmc
