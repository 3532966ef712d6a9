// microbench.hip -- instruction latency / issue-interval probes for the FP64
// building blocks of the E-step kernels (one wave, s_memtime shader cycles).
// Diagnostic only: psmc_hip_microbench() fills out[] with cycles per operation.
#include <hip/hip_runtime.h>
#include "wave_prims.h"
#include "struct_prims.h"
#include "psmc_hip_internal.h"

namespace psmc {

#define REPEAT16(X) X X X X X X X X X X X X X X X X
#define N_ITER 64

template <int WHICH>
__device__ __forceinline__ double probe(double x, double m)
{
	double a0 = x, a1 = x + 1, a2 = x + 2, a3 = x + 3, a4 = x + 4, a5 = x + 5, a6 = x + 6, a7 = x + 7;
	double r[4] = {x, x + 0.5, x + 0.25, x + 0.125};
	dpp_guard(r);
	for (int it = 0; it < N_ITER; ++it) {
		if (WHICH == 0) { REPEAT16(fmac_bcast<3>(a0, r[0], m);) }                       // dependent fmac_dpp
		else if (WHICH == 1) { REPEAT16(a0 = __builtin_fma(a0, m, x);) }                  // dependent v_fma_f64
		else if (WHICH == 2) { REPEAT16(a0 = a0 + m;) }                                   // dependent v_add_f64
		else if (WHICH == 3) {                                                            // 4 chains fmac_dpp
			REPEAT16(fmac_bcast<3>(a0, r[0], m); fmac_bcast<3>(a1, r[1], m); fmac_bcast<3>(a2, r[2], m); fmac_bcast<3>(a3, r[3], m);)
		} else if (WHICH == 4) {                                                          // 8 chains fmac_dpp
			REPEAT16(fmac_bcast<3>(a0, r[0], m); fmac_bcast<3>(a1, r[1], m); fmac_bcast<3>(a2, r[2], m); fmac_bcast<3>(a3, r[3], m);
			         fmac_bcast<5>(a4, r[0], m); fmac_bcast<5>(a5, r[1], m); fmac_bcast<5>(a6, r[2], m); fmac_bcast<5>(a7, r[3], m);)
		} else if (WHICH == 5) {                                                          // dependent rep_rows_swap
			REPEAT16(rep_rows_swap(a0, r); a0 = r[1];)
		} else if (WHICH == 6) {                                                          // dependent rep_rows_bperm
			REPEAT16(rep_rows_bperm(a0, r); a0 = r[1];)
		} else if (WHICH == 7) { REPEAT16(a0 = a0 + dpp_mov<0xB1>(a0);) }                 // dpp_mov(2x32) + add
		else if (WHICH == 8) { REPEAT16(a0 = __builtin_amdgcn_rcp(a0);) }                 // dependent v_rcp_f64
		else if (WHICH == 9) { REPEAT16(a0 = bcast16<3>(a0) * m;) }                       // mov_b64_dpp + mul (exact mode term)
		else if (WHICH == 10) { REPEAT16(a0 = a0 * m; a1 = a1 * m; a2 = a2 * m; a3 = a3 * m; a4 = a4 * m; a5 = a5 * m; a6 = a6 * m; a7 = a7 * m;) } // 8 indep v_mul_f64
		else if (WHICH == 11) { REPEAT16(a0 = a0 / m;) }                                  // dependent IEEE division
	}
	return a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + r[0] + r[1] + r[2] + r[3];
}

typedef double d4m_t __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(64) void k_microbench(double *out, double seed)
{
	const int lane = threadIdx.x;
	double sink = 0.0;
	const double x = seed + 1e-3 * lane, m = 1.0 + 1e-9 * lane;
#define RUN(W, OPS)                                                        \
	{                                                                      \
		const unsigned long long t0 = __builtin_readcyclecounter();         \
		const double v = probe<W>(x, m);                                    \
		const unsigned long long t1 = __builtin_readcyclecounter();         \
		sink += v;                                                          \
		if (lane == 0) out[W] = (double)(t1 - t0) / (double)(N_ITER * 16 * (OPS)); \
	}
	RUN(0, 1) RUN(1, 1) RUN(2, 1) RUN(3, 4) RUN(4, 8) RUN(5, 1) RUN(6, 1) RUN(7, 1) RUN(8, 1) RUN(9, 1) RUN(10, 8) RUN(11, 1)
	{ // f64 MFMA: dependent accumulate chain, and 4 independent accumulators
		d4m_t c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
		unsigned long long t0 = __builtin_readcyclecounter();
		for (int it = 0; it < N_ITER; ++it) { REPEAT16(c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, m, c0, 0, 0, 0);) }
		unsigned long long t1 = __builtin_readcyclecounter();
		if (lane == 0) out[12] = (double)(t1 - t0) / (N_ITER * 16);
		t0 = __builtin_readcyclecounter();
		for (int it = 0; it < N_ITER; ++it) {
			REPEAT16(c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, m, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, m, c1, 0, 0, 0);
			         c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, m, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, m, c3, 0, 0, 0);)
		}
		t1 = __builtin_readcyclecounter();
		if (lane == 0) out[13] = (double)(t1 - t0) / (N_ITER * 16 * 4);
		sink += c0[0] + c1[1] + c2[2] + c3[3];
		// do FP64 vector instructions of the same wave overlap with its matrix instructions?  Per group: 4 MFMA
		// (independent accumulators) followed by 32 independent v_fma_f64 (8 chains) / 32 v_mov_b32_dpp
		double f0 = x, f1 = x + 1, f2 = x + 2, f3 = x + 3, f4 = x + 4, f5 = x + 5, f6 = x + 6, f7 = x + 7;
#define MB_MFMA4 c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, m, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, m, c1, 0, 0, 0); \
		 c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, m, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, m, c3, 0, 0, 0);
#define MB_FMA8 f0 = __builtin_fma(f0, m, x); f1 = __builtin_fma(f1, m, x); f2 = __builtin_fma(f2, m, x); f3 = __builtin_fma(f3, m, x); \
		f4 = __builtin_fma(f4, m, x); f5 = __builtin_fma(f5, m, x); f6 = __builtin_fma(f6, m, x); f7 = __builtin_fma(f7, m, x);
		t0 = __builtin_readcyclecounter();
		for (int it = 0; it < N_ITER; ++it) { REPEAT16(MB_MFMA4 __builtin_amdgcn_sched_barrier(0); MB_FMA8 MB_FMA8 MB_FMA8 MB_FMA8 __builtin_amdgcn_sched_barrier(0);) }
		t1 = __builtin_readcyclecounter();
		if (lane == 0) out[18] = (double)(t1 - t0) / (N_ITER * 16);
		sink += c0[0] + c1[1] + c2[2] + c3[3] + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;
		int i0 = lane, i1 = lane + 1, i2 = lane + 2, i3 = lane + 3;
#define MB_DPP8 i0 = __builtin_amdgcn_update_dpp(0, i0, 0x111, 0xf, 0xf, true); i1 = __builtin_amdgcn_update_dpp(0, i1, 0x111, 0xf, 0xf, true); \
		i2 = __builtin_amdgcn_update_dpp(0, i2, 0x111, 0xf, 0xf, true); i3 = __builtin_amdgcn_update_dpp(0, i3, 0x111, 0xf, 0xf, true); \
		i0 = __builtin_amdgcn_update_dpp(0, i0, 0x101, 0xf, 0xf, true); i1 = __builtin_amdgcn_update_dpp(0, i1, 0x101, 0xf, 0xf, true); \
		i2 = __builtin_amdgcn_update_dpp(0, i2, 0x101, 0xf, 0xf, true); i3 = __builtin_amdgcn_update_dpp(0, i3, 0x101, 0xf, 0xf, true);
		t0 = __builtin_readcyclecounter();
		for (int it = 0; it < N_ITER; ++it) { REPEAT16(MB_MFMA4 __builtin_amdgcn_sched_barrier(0); MB_DPP8 MB_DPP8 MB_DPP8 MB_DPP8 __builtin_amdgcn_sched_barrier(0);) }
		t1 = __builtin_readcyclecounter();
		if (lane == 0) out[19] = (double)(t1 - t0) / (N_ITER * 16);
		sink += c0[0] + c1[1] + c2[2] + c3[3] + (double)(i0 + i1 + i2 + i3);
#undef MB_MFMA4
#undef MB_FMA8
#undef MB_DPP8
	}
	{ // structured O(N) sweep step (estep_struct.hip): dependent chain, 4 tiles per wave
		StructPar c;
		double xv[4];
		for (int i = 0; i < 4; ++i) {
			c.mS[i] = 0.01 + 1e-4 * (lane + i); c.wS[i] = 0.3; c.mP[i] = 0.02; c.wP[i] = 0.2 + 1e-3 * i; c.dd[i] = 0.97;
			xv[i] = x + i;
		}
		unsigned long long t0 = __builtin_readcyclecounter();
		for (int it = 0; it < N_ITER; ++it) { REPEAT16(struct_step(c, xv);) }
		unsigned long long t1 = __builtin_readcyclecounter();
		if (lane == 0) out[14] = (double)(t1 - t0) / (N_ITER * 16);
		sink += xv[0] + xv[1] + xv[2] + xv[3];
		// the same with the emission multiply and the every-4th-step normalisation
		t0 = __builtin_readcyclecounter();
		for (int it = 0; it < N_ITER * 4; ++it) {
			double ev[4] = {m, m, m, m};
			struct_step(c, xv); xv[0] *= ev[0]; xv[1] *= ev[1]; xv[2] *= ev[2]; xv[3] *= ev[3];
			struct_step(c, xv); xv[0] *= ev[0]; xv[1] *= ev[1]; xv[2] *= ev[2]; xv[3] *= ev[3];
			struct_step(c, xv); xv[0] *= ev[0]; xv[1] *= ev[1]; xv[2] *= ev[2]; xv[3] *= ev[3];
			const double inv = rcp_newton(row_sum16((xv[0] + xv[1]) + (xv[2] + xv[3])));
			ev[0] *= inv; ev[1] *= inv; ev[2] *= inv; ev[3] *= inv;
			struct_step(c, xv); xv[0] *= ev[0]; xv[1] *= ev[1]; xv[2] *= ev[2]; xv[3] *= ev[3];
		}
		t1 = __builtin_readcyclecounter();
		if (lane == 0) out[15] = (double)(t1 - t0) / (N_ITER * 16);
		sink += xv[0] + xv[1] + xv[2] + xv[3];
		// shader clock in MHz: cycle counter against the 100 MHz constant clock
		const unsigned long long w0 = wall_clock64();
		t0 = __builtin_readcyclecounter();
		for (int it = 0; it < N_ITER * 4; ++it) { REPEAT16(struct_step(c, xv);) }
		t1 = __builtin_readcyclecounter();
		const unsigned long long w1 = wall_clock64();
		if (lane == 0) out[16] = (double)(t1 - t0) / (double)(w1 - w0) * 100.0;
		sink += xv[0];
		// one-state-per-lane step (walks, fused kernel): two 64-lane scans, emission, norm every 4th step
		const WaveScanMasks wm = wave_scan_masks(lane);
		StructPar1 c1; c1.mS = c.mS[0]; c1.wS = c.wS[0]; c1.mP = c.mP[0]; c1.wP = c.wP[1]; c1.dd = c.dd[0];
		double y = x;
		t0 = __builtin_readcyclecounter();
		for (int it = 0; it < N_ITER * 4; ++it) {
			y = struct_step1(c1, y, wm) * m; y = struct_step1(c1, y, wm) * m; y = struct_step1(c1, y, wm) * m;
			const double inv = rcp_newton(first_lane_f64(wave_sum_nat(y)));
			y = struct_step1(c1, y, wm) * (m * inv);
		}
		t1 = __builtin_readcyclecounter();
		if (lane == 0) out[17] = (double)(t1 - t0) / (N_ITER * 16);
		sink += y;
	}
	if (sink == 123.456) out[63] = sink;
}

// 8-byte-per-lane streaming copy of a known size: calibrates the FETCH_SIZE / WRITE_SIZE
// PMC counters for the access width the E-step kernels use (MI355X_MICROARCH.md, HBM section).
__global__ __launch_bounds__(256) void k_stream_copy8(const double *__restrict__ src, double *__restrict__ dst, size_t n)
{
	for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i] + 1.0;
}
int run_stream_probe(hipStream_t stream, const double *src, double *dst, size_t n)
{
	hipLaunchKernelGGL(k_stream_copy8, dim3(8192), dim3(256), 0, stream, src, dst, n);
	return (int)hipGetLastError();
}

// Achievable-HBM probes, 16 bytes per lane like the table traffic of the E-step kernels: [0] fill, [1] read,
// [2] copy, [3] the store pattern of the structured sweeps (a wave appends 512 B per step to each of four
// streams, two 16-byte stores per lane 32 bytes apart).  The roofline of DESIGN.md quotes the nominal 8 TB/s;
// these are what a plain streaming kernel reaches on the same box.
typedef double pd2_t __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void k_hbm_fill(pd2_t *__restrict__ dst, size_t n)
{
	const pd2_t v = {1.0, 2.0};
	for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = v;
}
__global__ __launch_bounds__(256) void k_hbm_read(const pd2_t *__restrict__ src, size_t n, double *__restrict__ sink)
{
	double s = 0.0;
	for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { const pd2_t v = src[i]; s += v.x + v.y; }
	if (s == 123.456) sink[0] = s;
}
__global__ __launch_bounds__(256) void k_hbm_copy(const pd2_t *__restrict__ src, pd2_t *__restrict__ dst, size_t n)
{
	for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
__global__ __launch_bounds__(64) void k_hbm_sweepstore(double *__restrict__ dst, size_t steps_per_stream)
{
	const int lane = threadIdx.x, row = lane >> 4, m = lane & 15;
	double *o = dst + ((size_t)blockIdx.x * 4 + row) * steps_per_stream * 64 + 4 * m;
	pd2_t a = {1.0, 2.0}, b = {3.0, 4.0};
	for (size_t t = 0; t < steps_per_stream; ++t) {
		reinterpret_cast<pd2_t *>(o)[0] = a; reinterpret_cast<pd2_t *>(o)[1] = b;
		o += 64; a.x += 1.0; b.y += 1.0;
	}
}
int run_hbm_probe(hipStream_t stream, int which, double *a, double *b, size_t bytes)
{
	const size_t n16 = bytes / 16;
	if (which == 0) hipLaunchKernelGGL(k_hbm_fill, dim3(16384), dim3(256), 0, stream, (pd2_t *)a, n16);
	else if (which == 1) hipLaunchKernelGGL(k_hbm_read, dim3(16384), dim3(256), 0, stream, (const pd2_t *)a, n16, b);
	else if (which == 2) hipLaunchKernelGGL(k_hbm_copy, dim3(16384), dim3(256), 0, stream, (const pd2_t *)a, (pd2_t *)b, n16);
	else { // 4096-step streams (2 MB each)
		const size_t steps = 4096, streams = bytes / (steps * 512);
		hipLaunchKernelGGL(k_hbm_sweepstore, dim3((unsigned)(streams / 4)), dim3(64), 0, stream, a, steps);
	}
	return (int)hipGetLastError();
}

// The structured step (4 tiles per wave) on n_waves wavefronts at once: per wave the cycles per step and the
// shader clock it saw (cycle counter against the 100 MHz constant clock).  One wave alone runs at the peak
// clock; a full device of FP64 work may not.
__global__ __launch_bounds__(64) void k_load_probe(double *__restrict__ out, int steps, double seed)
{
	const int lane = threadIdx.x;
	StructPar c;
	double xv[4];
#pragma unroll
	for (int i = 0; i < 4; ++i) {
		c.mS[i] = 0.01 + 1e-4 * (lane + i) + seed * 1e-9; c.wS[i] = 0.012 - 1e-5 * lane; c.mP[i] = 0.009 + 1e-5 * i;
		c.wP[i] = 0.011; c.dd[i] = 0.3; xv[i] = 1.0 + 0.01 * (lane & 15) + 0.001 * i;
	}
	const unsigned long long w0 = wall_clock64(), t0 = __builtin_readcyclecounter();
	for (int it = 0; it < steps; it += 4) {
		struct_step(c, xv); struct_step(c, xv); struct_step(c, xv);
		const double inv = rcp_newton(row_sum16((xv[0] + xv[1]) + (xv[2] + xv[3])));
		struct_step(c, xv);
#pragma unroll
		for (int i = 0; i < 4; ++i) xv[i] *= inv;
	}
	const unsigned long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();
	if (lane == 0) {
		out[2 * blockIdx.x] = (double)(t1 - t0) / steps;
		out[2 * blockIdx.x + 1] = (double)(t1 - t0) / (double)(w1 - w0) * 100.0;
	}
	if (xv[0] + xv[1] + xv[2] + xv[3] == 123.456) out[0] = 0.0;
}
// the 16x4 step with the table stores of a sweep: every wave appends 512 B per step to each of four streams
// (mode 1: two 16-byte stores per lane and step;  mode 2: the same bytes, written as 2 KB per stream every 4th step)
template <int MODE>
__global__ __launch_bounds__(64) void k_load_probe_st(double *__restrict__ out, int steps, double seed, double *__restrict__ tbl,
                                                        int store_steps)
{
	const int lane = threadIdx.x, row = lane >> 4, m = lane & 15;
	StructPar c;
	double xv[4];
#pragma unroll
	for (int i = 0; i < 4; ++i) {
		c.mS[i] = 0.01 + 1e-4 * (lane + i) + seed * 1e-9; c.wS[i] = 0.012 - 1e-5 * lane; c.mP[i] = 0.009 + 1e-5 * i;
		c.wP[i] = 0.011; c.dd[i] = 0.3; xv[i] = 1.0 + 0.01 * (lane & 15) + 0.001 * i;
	}
	double *o = tbl + ((size_t)blockIdx.x * 4 + row) * (size_t)store_steps * 64 + 4 * m;
	const unsigned long long w0 = wall_clock64(), t0 = __builtin_readcyclecounter();
	double keep[3][4];
	for (int it = 0; it < steps; it += 4) {
		const bool st = it >= steps - store_steps;
#pragma unroll
		for (int j = 0; j < 4; ++j) {
			if (j == 3) {
				const double inv = rcp_newton(row_sum16((xv[0] + xv[1]) + (xv[2] + xv[3])));
#pragma unroll
				for (int i = 0; i < 4; ++i) xv[i] *= inv;
			}
			struct_step(c, xv);
			if (MODE == 1) { if (st) { store4(o, xv); o += 64; } }
			else if (j < 3) {
#pragma unroll
				for (int i = 0; i < 4; ++i) keep[j][i] = xv[i];
			} else if (st) { store4(o, keep[0]); store4(o + 64, keep[1]); store4(o + 128, keep[2]); store4(o + 192, xv); o += 256; }
		}
	}
	const unsigned long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();
	if (lane == 0) {
		out[2 * blockIdx.x] = (double)(t1 - t0) / steps;
		out[2 * blockIdx.x + 1] = (double)(t1 - t0) / (double)(w1 - w0) * 100.0;
	}
	if (xv[0] + xv[1] + xv[2] + xv[3] == 123.456) out[0] = 0.0;
}
int run_load_probe_st(hipStream_t stream, double *d_out, int n_waves, int steps, double *tbl, int store_steps, int mode)
{
	if (mode == 2) hipLaunchKernelGGL(k_load_probe_st<2>, dim3(n_waves), dim3(64), 0, stream, d_out, steps, 0.37, tbl, store_steps);
	else hipLaunchKernelGGL(k_load_probe_st<1>, dim3(n_waves), dim3(64), 0, stream, d_out, steps, 0.37, tbl, store_steps);
	return (int)hipGetLastError();
}

int run_load_probe(hipStream_t stream, double *d_out, int n_waves, int steps)
{
	hipLaunchKernelGGL(k_load_probe, dim3(n_waves), dim3(64), 0, stream, d_out, steps, 0.37);
	return (int)hipGetLastError();
}

// Do f64 matrix instructions of ONE wave overlap with f64 vector instructions of ANOTHER wave on the same SIMD?
// (Within a wave they do not: out[18] of k_microbench.)  One work-group of n_waves waves on one CU; bit w of `mask`
// makes wave w a matrix wave (64 independent-accumulator v_mfma_f64_16x16x4 per round), the others vector waves
// (1024 v_fma_f64 in 8 chains per round) -- about 4100 cycles of issue each when alone on a SIMD.  out[w] = shader
// cycles per round of wave w.  Waves of a work-group go to the CU's four SIMDs round robin (w % 4).
__global__ __launch_bounds__(512) void k_pipe_probe(double *out, double seed, unsigned mask, int rounds)
{
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	const double x = seed + 1e-3 * lane, m = 1.0 + 1e-9 * lane;
	const bool matrix = (mask >> w) & 1u;
	d4m_t c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
	double f0 = x, f1 = x + 1, f2 = x + 2, f3 = x + 3, f4 = x + 4, f5 = x + 5, f6 = x + 6, f7 = x + 7;
	__syncthreads();
	const unsigned long long t0 = __builtin_readcyclecounter();
	if (matrix) {
		for (int it = 0; it < rounds; ++it) {
			REPEAT16(c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, m, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, m, c1, 0, 0, 0);
			         c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, m, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, m, c3, 0, 0, 0);)
		}
	} else {
		for (int it = 0; it < rounds * 8; ++it) {
			REPEAT16(f0 = __builtin_fma(f0, m, x); f1 = __builtin_fma(f1, m, x); f2 = __builtin_fma(f2, m, x); f3 = __builtin_fma(f3, m, x);
			         f4 = __builtin_fma(f4, m, x); f5 = __builtin_fma(f5, m, x); f6 = __builtin_fma(f6, m, x); f7 = __builtin_fma(f7, m, x);)
		}
	}
	const unsigned long long t1 = __builtin_readcyclecounter();
	if (lane == 0) out[w] = (double)(t1 - t0) / rounds;
	if (c0[0] + c1[1] + c2[2] + c3[3] + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 == 123.456) out[0] = 0.0;
}
int run_pipe_probe(hipStream_t stream, double *d_out, int n_waves, unsigned mask, int rounds)
{
	hipLaunchKernelGGL(k_pipe_probe, dim3(1), dim3(64 * n_waves), 0, stream, d_out, 0.37, mask, rounds);
	return (int)hipGetLastError();
}

int run_microbench(hipStream_t stream, double *d_out)
{
	hipLaunchKernelGGL(k_microbench, dim3(1), dim3(64), 0, stream, d_out, 0.37);
	return (int)hipGetLastError();
}


// ---- pipe probe, second edition (round 3): what DOES overlap with another wave's v_mfma_f64 on the same SIMD?
// The first probe paired matrix waves with v_fma_f64 waves only and found one shared FP64 pipe.  The fused step's other
// instructions are mostly not f64 arithmetic: scan levels (two v_mov_b32_dpp + one v_add_f64), LDS reads of the
// emission rows, scalar loads of the symbols, v_readlane.  One work-group of up to 8 waves on one CU (wave w -> SIMD
// w % 4); kinds[w] chooses what wave w issues, ~4 k cycles of issue per round when alone:
//   0 idle (exits)            1 v_mfma_f64_16x16x4 x 64           2 v_fma_f64 x 1024 (8 chains)
//   3 v_mov_b32_dpp x 1024    4 scan levels: (2 v_mov_b32_dpp + v_add_f64) x 341, as row_excl_prefix emits them
//   5 ds_read_b128 x 512      6 s_load_dwordx4 x 256 + v_readlane_b32 x 512
//   7 v_add_u32 x 1024        8 v_fma_f32 x 1024                  9 v_add_f64 x 1024
// out[w] = shader cycles per round of wave w.
struct PipeKinds { int k[8]; };
__global__ __launch_bounds__(512) void k_pipe_probe2(double *out, double seed, PipeKinds kinds, int rounds, const uint4 *__restrict__ gsrc)
{
	__shared__ uint4 lds[1024];
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	for (int i = threadIdx.x; i < 1024; i += blockDim.x) lds[i] = make_uint4(i, i + 1, i + 2, i + 3);
	const int kind = kinds.k[w & 7];
	const double x = seed + 1e-3 * lane, m = 1.0 + 1e-9 * lane;
	d4m_t c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
	double f0 = x, f1 = x + 1, f2 = x + 2, f3 = x + 3, f4 = x + 4, f5 = x + 5, f6 = x + 6, f7 = x + 7;
	float g0 = (float)x, g1 = g0 + 1, g2 = g0 + 2, g3 = g0 + 3, g4 = g0 + 4, g5 = g0 + 5, g6 = g0 + 6, g7 = g0 + 7;
	const float gm = 1.0f + 1e-6f * lane;
	int i0 = lane, i1 = lane + 1, i2 = lane + 2, i3 = lane + 3, i4 = lane + 4, i5 = lane + 5, i6 = lane + 6, i7 = lane + 7;
	unsigned acc = 0;
	__syncthreads();
	const unsigned long long t0 = __builtin_readcyclecounter();
	if (kind == 1) {
		for (int it = 0; it < rounds; ++it) {
			REPEAT16(c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, m, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, m, c1, 0, 0, 0);
			         c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, m, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, m, c3, 0, 0, 0);)
		}
	} else if (kind == 2) {
		for (int it = 0; it < rounds * 8; ++it) {
			REPEAT16(f0 = __builtin_fma(f0, m, x); f1 = __builtin_fma(f1, m, x); f2 = __builtin_fma(f2, m, x); f3 = __builtin_fma(f3, m, x);
			         f4 = __builtin_fma(f4, m, x); f5 = __builtin_fma(f5, m, x); f6 = __builtin_fma(f6, m, x); f7 = __builtin_fma(f7, m, x);)
		}
	} else if (kind == 3) {
#define PSMC_DPP8 asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n" \
		"v_mov_b32_dpp %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n" \
		"v_mov_b32_dpp %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n" \
		"v_mov_b32_dpp %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n" \
		: "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7));
		for (int it = 0; it < rounds * 8; ++it) { REPEAT16(PSMC_DPP8) }
#undef PSMC_DPP8
	} else if (kind == 4) {
		// 4 chains x (4 scan levels = 12 instructions) x 21 per round ~ 1008 instructions, two thirds of them DPP moves
		for (int it = 0; it < rounds * 21; ++it) {
			f0 = f0 + dpp_z<0x111>(f0); f1 = f1 + dpp_z<0x111>(f1); f2 = f2 + dpp_z<0x111>(f2); f3 = f3 + dpp_z<0x111>(f3);
			f0 = f0 + dpp_z<0x112>(f0); f1 = f1 + dpp_z<0x112>(f1); f2 = f2 + dpp_z<0x112>(f2); f3 = f3 + dpp_z<0x112>(f3);
			f0 = f0 + dpp_z<0x114>(f0); f1 = f1 + dpp_z<0x114>(f1); f2 = f2 + dpp_z<0x114>(f2); f3 = f3 + dpp_z<0x114>(f3);
			f0 = f0 + dpp_z<0x118>(f0); f1 = f1 + dpp_z<0x118>(f1); f2 = f2 + dpp_z<0x118>(f2); f3 = f3 + dpp_z<0x118>(f3);
			f0 *= 0.03125; f1 *= 0.03125; f2 *= 0.03125; f3 *= 0.03125; // keep the sums finite (4 more v_mul_f64: 1024 in all)
		}
	} else if (kind == 5) {
		const unsigned a = (unsigned)(size_t)lds + 16u * (unsigned)lane; // LDS byte address
		for (int it = 0; it < rounds * 32; ++it) {
			uint4 r0, r1, r2, r3, r4, r5, r6, r7, r8, r9, ra, rb, rc, rd, re, rf;
			asm volatile("ds_read_b128 %0, %16\n ds_read_b128 %1, %16 offset:1024\n ds_read_b128 %2, %16 offset:2048\n ds_read_b128 %3, %16 offset:3072\n"
			             "ds_read_b128 %4, %16 offset:4096\n ds_read_b128 %5, %16 offset:5120\n ds_read_b128 %6, %16 offset:6144\n ds_read_b128 %7, %16 offset:7168\n"
			             "ds_read_b128 %8, %16 offset:8192\n ds_read_b128 %9, %16 offset:9216\n ds_read_b128 %10, %16 offset:10240\n ds_read_b128 %11, %16 offset:11264\n"
			             "ds_read_b128 %12, %16 offset:12288\n ds_read_b128 %13, %16 offset:13312\n ds_read_b128 %14, %16 offset:14336\n ds_read_b128 %15, %16 offset:15360\n"
			             "s_waitcnt lgkmcnt(0)\n"
			             : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(r4), "=v"(r5), "=v"(r6), "=v"(r7), "=v"(r8), "=v"(r9), "=v"(ra), "=v"(rb), "=v"(rc), "=v"(rd), "=v"(re), "=v"(rf)
			             : "v"(a) : "memory");
			acc ^= r0.x ^ rf.w;
		}
	} else if (kind == 6) {
		for (int it = 0; it < rounds * 64; ++it) {
			uint4 s0, s1, s2, s3; int q0, q1, q2, q3, q4, q5, q6, q7;
			asm volatile("s_load_dwordx4 %0, %12, 0x0\n s_load_dwordx4 %1, %12, 0x10\n s_load_dwordx4 %2, %12, 0x20\n s_load_dwordx4 %3, %12, 0x30\n"
			             "v_readlane_b32 %4, %13, 0\n v_readlane_b32 %5, %13, 16\n v_readlane_b32 %6, %13, 32\n v_readlane_b32 %7, %13, 48\n"
			             "v_readlane_b32 %8, %13, 1\n v_readlane_b32 %9, %13, 17\n v_readlane_b32 %10, %13, 33\n v_readlane_b32 %11, %13, 49\n"
			             "s_waitcnt lgkmcnt(0)\n"
			             : "=s"(s0), "=s"(s1), "=s"(s2), "=s"(s3), "=s"(q0), "=s"(q1), "=s"(q2), "=s"(q3), "=s"(q4), "=s"(q5), "=s"(q6), "=s"(q7)
			             : "s"(gsrc), "v"(i0) : "memory");
			acc ^= s0.x ^ s3.w ^ (unsigned)q0 ^ (unsigned)q7;
		}
	} else if (kind == 7) {
		for (int it = 0; it < rounds * 8; ++it) {
			REPEAT16(asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n"
			         : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(lane));)
		}
	} else if (kind == 8) {
		for (int it = 0; it < rounds * 8; ++it) {
			REPEAT16(g0 = __builtin_fmaf(g0, gm, gm); g1 = __builtin_fmaf(g1, gm, gm); g2 = __builtin_fmaf(g2, gm, gm); g3 = __builtin_fmaf(g3, gm, gm);
			         g4 = __builtin_fmaf(g4, gm, gm); g5 = __builtin_fmaf(g5, gm, gm); g6 = __builtin_fmaf(g6, gm, gm); g7 = __builtin_fmaf(g7, gm, gm);)
		}
	} else if (kind == 9) {
		for (int it = 0; it < rounds * 8; ++it) {
			REPEAT16(f0 = f0 + m; f1 = f1 + m; f2 = f2 + m; f3 = f3 + m; f4 = f4 + m; f5 = f5 + m; f6 = f6 + m; f7 = f7 + m;)
		}
	}
	const unsigned long long t1 = __builtin_readcyclecounter();
	if (lane == 0 && w < 8) out[w] = kind == 0 ? 0.0 : (double)(t1 - t0) / rounds;
	if (c0[0] + c1[1] + c2[2] + c3[3] + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 + g0 + g1 + g2 + g3 + g4 + g5 + g6 + g7 + (double)(i0 + i1 + i2 + i3 + i4 + i5 + i6 + i7) + (double)acc == 123.456) out[0] = 0.0;
}
int run_pipe_probe2(hipStream_t stream, double *d_out, const int *kinds8, int rounds, const void *gsrc)
{
	PipeKinds k;
	int n = 0;
	for (int i = 0; i < 8; ++i) { k.k[i] = kinds8[i]; if (kinds8[i]) n = i + 1; }
	if (n == 0) return 0;
	hipLaunchKernelGGL(k_pipe_probe2, dim3(1), dim3(64 * n), 0, stream, d_out, 0.37, k, rounds, (const uint4 *)gsrc);
	return (int)hipGetLastError();
}

// ---- placement probe (round 3): where do the waves of a small launch land?  n_waves waves of the structured step in
// work-groups of wpb waves; every wave records XCC_ID and HW_ID (SE / CU / SIMD) and its cycles per step.  A shard-sized
// E-step has fewer waves than the device has SIMDs (1024): if the dispatcher stacks them, a step costs twice its latency.
__global__ __launch_bounds__(256) void k_place_probe(double *__restrict__ out, int steps, double seed)
{
	const int lane = threadIdx.x & 63, w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
	StructPar c;
	double xv[4];
#pragma unroll
	for (int i = 0; i < 4; ++i) {
		c.mS[i] = 0.01 + 1e-4 * (lane + i) + seed * 1e-9; c.wS[i] = 0.012 - 1e-5 * lane; c.mP[i] = 0.009 + 1e-5 * i;
		c.wP[i] = 0.011; c.dd[i] = 0.3; xv[i] = 1.0 + 0.01 * (lane & 15) + 0.001 * i;
	}
	// seed < 0 (PSMC_HIP_PROBE_PRIO=1): the first half of the grid -- the first wave the dispatcher puts on every SIMD -- runs at wave
	// priority 0, the second half at 3: does s_setprio decide who issues, or the age of the wave?
	if (seed < 0.0) { if (2 * blockIdx.x < gridDim.x) __builtin_amdgcn_s_setprio(0); else __builtin_amdgcn_s_setprio(3); } // the YOUNGER wave of every SIMD at the higher priority
	const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);   // HW_REG_HW_ID
	const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20); // HW_REG_XCC_ID
	const unsigned long long t0 = __builtin_readcyclecounter();
	for (int it = 0; it < steps; it += 4) {
		struct_step(c, xv); struct_step(c, xv); struct_step(c, xv);
		const double inv = rcp_newton(row_sum16((xv[0] + xv[1]) + (xv[2] + xv[3])));
		struct_step(c, xv);
#pragma unroll
		for (int i = 0; i < 4; ++i) xv[i] *= inv;
	}
	const unsigned long long t1 = __builtin_readcyclecounter();
	if (lane == 0) { out[3 * w] = (double)(t1 - t0) / steps; out[3 * w + 1] = (double)hw; out[3 * w + 2] = (double)xcc; }
	if (xv[0] + xv[1] + xv[2] + xv[3] == 123.456) out[0] = 0.0;
}
int run_place_probe(hipStream_t stream, double *d_out, int n_waves, int wpb, int steps)
{
	hipLaunchKernelGGL(k_place_probe, dim3((n_waves + wpb - 1) / wpb), dim3(64 * wpb), 0, stream, d_out, steps, getenv("PSMC_HIP_PROBE_PRIO") ? -0.37 : 0.37);
	return (int)hipGetLastError();
}

} // namespace psmc
