// estep_fused.hip -- FAST mode, structured matrices: backward sweep and expected counts in ONE kernel.
//
// The counts kernel of estep_fast.hip reads X and bt back from HBM (1 KB per bin) after the backward
// sweep wrote bt (0.5 KB per bin).  With O(N) sweep steps (estep_struct.hip) the E-step is HBM-bound,
// so here the wave that walks a tile backwards feeds its bt vectors straight into the f64 matrix
// cores: bt never goes to memory, HBM traffic per bin drops from 2.1 KB to 1.0 KB (write X, read X).
//
// One wave per tile, lane = state for the sweep (struct_step1: two 64-lane scans), and the MFMA
// layout of k_expect_mfma for the counts: a group is four consecutive positions 4j..4j+3, row group
// t = lane>>4 holds position 4j+t; the sweep writes bt_p into a small LDS ring (slot p & 7) and the
// count stage reads bt_P, bt_{P+1} back in operand layout.  Only position 4j of a group carries a
// scale factor sb (lagged, sparse normalisation as everywhere else).
//   g_P[k] = X_P[k] bt_P[k] / e[o_P][k],  G_P = sum_k g_P[k]
//   S[o_P][k] += g_P[k] / G_P,   C[k][l] += (sb_P / G_P) X_P[k] bt_{P+1}[l],   A = a .* C  (k_reduce2)
// The tile's start vector bt_{top+1} comes from `bentry` (left there by the warm-up-only pass of
// k_bwd_struct or by a walk) or, for a repair, from the exit vector of the tile above; the exit
// vector bt_lo goes to `bexit` for the verify kernel.  Speculate / verify / repair as in estep_fast.hip.
#include <hip/hip_runtime.h>
#include "wave_prims.h"
#include "struct_prims.h"
#include "psmc_hip_internal.h"

namespace psmc {

typedef double d4f_t __attribute__((ext_vector_type(4)));
struct SweepItemF { int first, count; };

// mode 0: tiles items[0..n) from bentry;  mode 1: flagged tiles items[0..n) from the exit vector of the
// tile above (which also becomes their bentry);  mode 2: every tile b < n whose X a forward repair
// rewrote (touch_f), from bentry.
__global__ __launch_bounds__(64, 2) void k_bwd_count_struct(const double *__restrict__ sp, const double *__restrict__ e,
                                                              const double *__restrict__ re, const uint8_t *__restrict__ obs,
                                                              const Chunk *__restrict__ chunks, const SweepItemF *__restrict__ items,
                                                              int n, int mode, const double *__restrict__ f,
                                                              double *__restrict__ bentry, double *__restrict__ bexit,
                                                              double *__restrict__ Cpart, double *__restrict__ Spart,
                                                              const int *__restrict__ touch_f, int *__restrict__ touch_b)
{
	__shared__ double lds_bt[8 * 64];
	__shared__ double lds_sc[8];
	const int lane = threadIdx.x, t = lane >> 4, i = lane & 15;
	int tile;
	if (mode == 2) { tile = blockIdx.x; if (!touch_f[tile]) return; }
	else tile = items[blockIdx.x].first;
	const Chunk c = chunks[tile];
	const int L = c.L, lo = c.lo, top = min(c.hi, L - 1);
	d4f_t acc[4][4];
	double S[3][4];
#pragma unroll
	for (int m = 0; m < 4; ++m) {
#pragma unroll
		for (int nn = 0; nn < 4; ++nn) acc[m][nn] = (d4f_t){0.0, 0.0, 0.0, 0.0};
		S[0][m] = S[1][m] = S[2][m] = 0.0;
	}
	if (top >= lo) { // a tile holding only position L owns no transition: its partials are zero
		if (mode == 1) __builtin_amdgcn_s_setprio(3);
		const uint8_t *o = obs + c.off;
		const double *fo = f + c.off * 64 + i;
		StructPar1 sc1; // backward: SUF over z.c weighted by R, PRE over z.qa weighted by P
		sc1.mS = sp[192 + lane]; sc1.wS = sp[64 + lane]; sc1.mP = sp[128 + lane]; sc1.wP = sp[lane]; sc1.dd = sp[256 + lane];
		const WaveScanMasks wm = wave_scan_masks(lane);
		const double e0 = e[lane], e1 = e[64 + lane];
		double re0[4], re1[4]; // 1/e[b][16m+i]
#pragma unroll
		for (int m = 0; m < 4; ++m) { re0[m] = re[16 * m + i]; re1[m] = re[64 + 16 * m + i]; }
		double x; // bt_{p+1}, natural layout
		if (mode == 1) {
			x = bexit[(int64_t)(tile + 1) * 64 + lane];
			bentry[(int64_t)tile * 64 + lane] = x;
			if (lane == 0) touch_b[tile] = 1;
		} else {
			x = bentry[(int64_t)tile * 64 + lane];
		}
		lds_bt[((top + 1) & 7) * 64 + lane] = x;
		// operands of a group: X rows and symbols of the positions 4j+t (clamped into the tile)
		auto load = [&](int j, double (&FA)[4], int &sym, bool &ok) {
			const int P = 4 * j + t;
			ok = P >= lo && P <= top;
			const int64_t idx = (int64_t)min(max(P, lo), top) - 1;
			const double *fr = fo + idx * 64;
#pragma unroll
			for (int m = 0; m < 4; ++m) FA[m] = fr[16 * m];
			sym = o[idx];
		};
		const int j_top = top >> 2, j_lo = lo >> 2;
		double FA[4]; int sym; bool ok;
		load(j_top, FA, sym, ok);
		for (int j = j_top; j >= j_lo; --j) {
			double FN[4] = {0, 0, 0, 0}; int symn = 2; bool okn = false;
			if (j > j_lo) load(j - 1, FN, symn, okn);
			// ---- sweep through the group's positions, highest first
#pragma unroll
			for (int tt = 3; tt >= 0; --tt) {
				const int p = 4 * j + tt;
				if (p > top || p < lo) continue; // wave-uniform
				const int s_p = __builtin_amdgcn_readlane(sym, 16 * tt);
				double ev = s_p == 0 ? e0 : (s_p == 1 ? e1 : 1.0);
				if (tt == 0) { // p % NORM_EVERY == 0: sb_p = 1/sum(bt_{p+1}), off the critical path
					const double s = rcp_newton(first_lane_f64(wave_sum_nat(x)));
					ev *= s;
					if (lane == 0) lds_sc[j & 7] = s;
				}
				x = struct_step1(sc1, x, wm) * ev;
				lds_bt[(p & 7) * 64 + lane] = x;
				if (p == lo) bexit[(int64_t)tile * 64 + lane] = x;
			}
			// ---- counts of the group (row group t = position 4j+t)
			const int P = 4 * j + t;
			double BP[4], BM[4];
#pragma unroll
			for (int m = 0; m < 4; ++m) {
				BP[m] = lds_bt[(P & 7) * 64 + 16 * m + i];
				BM[m] = lds_bt[((P + 1) & 7) * 64 + 16 * m + i];
			}
			const double sc = (t == 0 && 4 * j >= lo) ? lds_sc[j & 7] : 1.0;
			double g[4], G = 0.0;
#pragma unroll
			for (int m = 0; m < 4; ++m) {
				const double r = sym == 0 ? re0[m] : (sym == 1 ? re1[m] : 1.0);
				g[m] = ok ? FA[m] * BP[m] * r : 0.0;
				G += g[m];
			}
			G = G + dpp_mov<0xB1>(G);  // quad_perm:[1,0,3,2]
			G = G + dpp_mov<0x4E>(G);  // quad_perm:[2,3,0,1]
			G = G + dpp_mov<0x124>(G); // row_ror:4
			G = G + dpp_mov<0x128>(G); // row_ror:8
			const double iG = ok ? rcp_newton(G) : 0.0; // rows outside the tile contribute nothing
			const double w0 = sym == 0 ? iG : 0.0, w1 = sym == 1 ? iG : 0.0, w2 = sym == 2 ? iG : 0.0, wa = sc * iG;
#pragma unroll
			for (int m = 0; m < 4; ++m) {
				S[0][m] = __builtin_fma(g[m], w0, S[0][m]);
				S[1][m] = __builtin_fma(g[m], w1, S[1][m]);
				S[2][m] = __builtin_fma(g[m], w2, S[2][m]);
				FA[m] = ok ? FA[m] * wa : 0.0;
				if (!ok) BM[m] = 0.0; // never multiply stale LDS contents (0 * NaN)
			}
#pragma unroll
			for (int m = 0; m < 4; ++m)
#pragma unroll
				for (int nn = 0; nn < 4; ++nn)
					acc[m][nn] = __builtin_amdgcn_mfma_f64_16x16x4f64(FA[m], BM[nn], acc[m][nn], 0, 0, 0);
#pragma unroll
			for (int m = 0; m < 4; ++m) FA[m] = FN[m];
			sym = symn; ok = okn;
		}
	}
	const double mult = (double)c.mult;
	double *out = Cpart + (int64_t)tile * 4096;
#pragma unroll
	for (int m = 0; m < 4; ++m)
#pragma unroll
		for (int nn = 0; nn < 4; ++nn)
#pragma unroll
			for (int r = 0; r < 4; ++r) out[(16 * m + t + 4 * r) * 64 + 16 * nn + i] = acc[m][nn][r] * mult;
	double *os = Spart + (int64_t)tile * 192;
#pragma unroll
	for (int b = 0; b < 3; ++b)
#pragma unroll
		for (int m = 0; m < 4; ++m) {
			double v = S[b][m];
			v += __shfl_xor(v, 16, 64);
			v += __shfl_xor(v, 32, 64);
			if (t == 0) os[b * 64 + 16 * m + i] = v * mult;
		}
}

// which: 0 = bulk single tiles of the backward item list, 1 = flagged tiles of the current repair round,
//        2 = every tile a forward repair touched, 3 = the tiles of the glued runs (boundary vectors from the walk)
void launch_bwd_count(const EstepLaunch &p, hipStream_t st, int which, int first, int n)
{
	if (n <= 0) return;
	const SweepItemF *items = (const SweepItemF *)(which == 1 ? p.d_ritems_b : (which == 3 ? p.d_members_b : p.d_items_b)) + first;
	hipLaunchKernelGGL(k_bwd_count_struct, dim3(n), dim3(64), 0, st, p.d_sp, p.d_e, p.d_re, p.d_obs, p.d_chunks, items, n,
	                   which == 1 ? 1 : (which == 2 ? 2 : 0), p.d_f, p.d_bentry, p.d_bexit, p.d_Cpart, p.d_Epart, p.d_touch_f,
	                   p.d_touch_b);
}

} // namespace psmc
