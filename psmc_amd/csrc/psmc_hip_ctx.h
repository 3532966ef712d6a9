// psmc_hip_ctx.h -- the context behind the C-ABI of include/psmc_hip.h and the helpers its translation units share:
//   api.hip        context, options, segments, parameter staging, tables, exact mode, table readers / decoding
//   api_fast.hip   fast mode: tile plan, sweep items, learning, the launch of one fast E-step and its entry points
//   api_batch.hip  psmc_hip_estep_batch (bootstrap replicates): exact launch groups, fast per-replicate plans
//   api_probes.hip device self-test, microbenchmarks and probes (diagnostics)
// Everything is built with -ffp-contract=off: no host or device expression is ever fused.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <string>
#include <thread>
#include <vector>
#include "psmc_hip.h"
#include "psmc_hip_internal.h"

using namespace psmc;

#define HMM_TINY_H 1e-25

struct psmc_hip_ctx {
	int n = 0, ns = 64, device = 0, mode = PSMC_HIP_MODE_EXACT; // ns: states padded to 64 or 128
	std::string err;
	// options
	int chunk = 0, warmup = 3072, max_rounds = 4096, rep_impl = -1, n_sub = 6, target_waves = 1536, overlap = 1;
	double warm_tol = 1e-12;
	int struct_opt = 1;        // "structured": 1 = use the O(N) sweeps when a[][] factors (auto), 0 = always dense
	bool struct_tiles_set = false; // the caller chose struct_tiles: no adaptation to small inputs
	int struct_tiles = 8192;   // "struct_tiles": tiles aimed at when the structured sweeps are used (4 per wave)
	bool use_struct = false, planned_struct = false;
	int last_fused = 0, last_ckpt = 0; // what the last fast E-step ran: EstepLaunch::fused / ckpt
	bool want_factored = false; // this call asked for the factored statistics (psmc_hip_estep_factored)
	int kc_sub = 4;            // "kc_sub": k_kcol2_struct cuts a tile's steps into this many ranges, one matrix (and one pair of waves) each; default: by the plan
	int kcol_prio = 2;         // "kcol_prio": wave priority of k_kcol2_struct
	int two_phase = -1;        // "two_phase": 2 = the fused back half runs as two launches and the tiles of the second list with an odd index start from
	                           // the exit vector of the tile above instead of speculating; 0 = every tile speculates; -1 = by the plan
	int two_phase_used = 2;    // what plan_fast chose
	int merge1 = -1;           // "merge1": bulk forward sweep + backward warm-up pass in one grid (k_sweep_struct); -1 = by the plan (shard-sized inputs)
	int merge1_used = 0;
	int merge_order = -1;      // "merge_order": block order of that grid: 1 = forward blocks, then backward blocks; 0 = alternating (every XCD gets one
	                           // direction); -1 = by the plan: 1 while a tile is shorter than its warm-up (measured: 3.75 M bins 3.00 vs 3.25 ms, 7.5 M equal, 15 M 7.9 vs 7.6)
	int lanes8 = -1;           // "lanes8": 64 states, fused / factored plans: the bulk sweeps of phase 1 run eight tiles per wave (8 lanes x 8 states: a quarter
	                           // fewer vector instructions per tile-step, half the waves).  -1 = with the factored statistics of a genome-sized input only (more than one round of tiles) -- measured (round 4, genome):
	                           // factored 10.26 -> 9.91 ms; full counts 12.09 -> 12.73 (its forward sweep is paced by 15.6 GB of stores and half as many waves hide less)
	int gate = -1;             // "gate": order the dispatch of phase 1's grids walks -> bulk -> transfer matrices (estep_struct.hip k_gate); -1 = with coarse
	                           // items (measured: without them the bulk grid is the critical path and walks that land late, stacked on few SIMDs, slow fewer of its waves)
	int *d_gate = nullptr;
	int coarse = -1;           // "coarse": a bulk sweep item spans up to this many consecutive tiles of a segment: ONE speculative warm-up per item and
	                           // direction, the backward pass walks the item and leaves every tile's start vector (build_items); -1 = by the plan
	int coarse_used = 1, items_coarse = -1;
	bool warm_shift_set = false, kc_sub_set = false;
	int warm_shift_used = 1, kc_sub_used = 4;
	int kc_div = 16;           // "kc_div": at most n_tiles / kc_div tiles per direction get a transfer matrix (16 tile sweeps of work each)
	int kc_min = -1;           // "kc_min": runs of at least this many tiles get the transfer-matrix chain instead of a walk (0: never; -1: 4 / 5 with
	                           // 64 states (one round of tiles / two), 8 / 12 with 65..128 -- measured, build_items)
	int n_wl_f = 0, n_wl_b = 0, n_kc = 0, n_chain_f = 0, n_chain_b = 0, n_singles_b = 0;
	double *d_Kcol = nullptr; size_t kcol_cap = 0;
	hipStream_t stream5 = nullptr;
	int ckpt = 1;              // "ckpt": factored statistics recompute X from checkpoints every 8 positions instead of reading the table
	int fuse = 1;              // "fuse": backward sweep and counts in one kernel (estep_fused.hip): structured matrices, up to 64 states
	int fuse128 = 2;           // "fuse128": the same with 65..128 states: 2 = k_bwd_count8x_struct (sixteen tiles per work-group, one sweep per tile, operands
	                           // exchanged through LDS), 0 = unfused (1, round 3's kernel -- four waves redo the sweep of four tiles -- was removed in round 6)
	int count_group = 4;       // tiles per work-group of the fused back half, what the tile lists are padded to (build_items)
	// Round 6: mis-speculation made cheap, then warm-ups sized per tile -- built, measured, and OFF: halving every warm-up at no
	// cost at all is worth 0.35 ms of 12.25 (profiles/r06_warmup_sensitivity.txt), and the three together cost more than that (profiles/r06_fix_pass_ab.txt).
	int merge = 0;             // "merge": the forward fix pass between the forward sweep and the back half (estep_struct.hip FwdCtl): a mis-speculated tile is rewritten
	                           // until it meets its stored trajectory, nothing is counted twice (0: verify afterwards, whole tiles and their groups of the counts again)
	int adapt = 0;             // "adapt": with "merge", every speculating tile's forward warm-up follows the mismatch its speculation left at the last E-step (0: "warmup" for all)
	int prev_start = 0;        // "prev_start": forward warm-ups start from the previous E-step's X at that position instead of the stationary vector
	int *h_mlen = nullptr, *m_mlen = nullptr;        // pinned + device-mapped [n_chunks]: blocks each merging repair rewrote / -1
	double *h_mis = nullptr, *m_mis = nullptr;       // pinned + device-mapped [2][n_chunks]: first-verify mismatch of every tile, forward | backward
	double *d_prevx = nullptr, *d_finv = nullptr;    // [n_chunks][ns] gathered start vectors; [n_chunks] scale factors of the merging repairs
	std::vector<int> order_f;                        // head tiles of the bulk forward items in launch order (build_items): adapt_warmups looks at how well the rows of a wave still match
	int n_fix_f = 0;
	bool merge_used = false;                         // what the last E-step decided about "merge"; n_fix_f: tiles of its first fix launch
	Chunk *h_chunks = nullptr; size_t h_chunks_cap = 0; // pinned staging copy of the tiles (their warm-ups change between E-steps)
	bool prev_ok = false;                            // the X table holds a complete forward sweep of THIS plan ...
	unsigned long long prev_serial = 0, tab_serial = 0; // ... and nobody has used the tables since (tab_serial: on the root context, ensure_tables)
	const double *prev_f = nullptr; int prev_ckpt = 0;
	int learn = 1;             // "learn": glue tiles that needed a repair to their neighbour for the following E-steps
	int warm_shift = 1;        // "warm_shift": before that, once, give such a tile a warm-up of warmup << warm_shift bins (0: glue at once)
	bool chunks_dirty = false; // a tile's warm-up changed: d_chunks is stale
	int group_cap = 131072;    // "group_cap": longest run of glued tiles, in bins
	int *d_items = nullptr;    // items_f | items_b | ritems_f | ritems_b, 2*n_chunks ints each
	int *h_ritems = nullptr;   // pinned + device-mapped, 2 * 2*n_chunks ints
	int *m_ritems = nullptr, *m_cnt = nullptr; // device views of h_ritems / h_cnt
	std::vector<uint8_t> glue_f, glue_b; // glue_f[b]: tile b continues the forward item of b-1; glue_b[b]: b continues b+1's backward item
	std::vector<uint8_t> gap;            // gap[b]: tile b is a FULL tile of missing data inside its segment (plan_fast: k_tile_allmiss): glued at once,
	                                     // free of "group_cap", one shared transfer matrix per direction (build_items)
	int gap_tiles = 1;                   // "gap_tiles": 0 = treat tiles of missing data like any other (rounds 1-4)
	int n_kuniq = 0;                     // transfer-matrix slots of the current item lists
	std::vector<int> flagged_f, flagged_b;
	bool items_dirty = true;
	int n_items_f = 0, n_items_b = 0;
	int n_sub_used = 6;
	// segments
	int n_seg = 0;
	std::vector<int32_t> L;
	std::vector<int64_t> off;
	int64_t total = 0; // padded bins
	bool obs_borrowed = false;
	uint8_t *d_obs = nullptr;
	int64_t *d_seg_off = nullptr;
	int32_t *d_seg_len = nullptr;
	// selection
	std::vector<int32_t> sel;      // as given
	std::vector<int32_t> work;     // unique selected ids
	std::vector<int32_t> sel2work; // sel[i] -> index into work
	std::vector<int32_t> mult;     // per work item
	int32_t *d_work = nullptr;
	bool plan_dirty = true;
	// parameters
	double *h_par = nullptr, *d_par = nullptr; // a | aeT(3) | e(3) | a0 | re(3)
	size_t par_len = 0;        // doubles in one parameter block: PAR_LEN up to 128 states, 2 S^2 + 4 S beyond (a | aT | e(3) | a0, estep_wide.hip)
	static constexpr size_t PAR_LEN = 2 * 16384 + 3 * 128 + 128 + 3 * 128 + 5 * 128 + 2 * 11 * 128; // ns=64: ... | re(3) | sp(5) (17152) | kcc; ns=128: a | aT | e(3) | a0 | re(3) | sp(5) | kcc
	static constexpr size_t RE128_OFF = 2 * 16384 + 3 * 128 + 128, SP128_OFF = RE128_OFF + 3 * 128, KCC128_OFF = SP128_OFF + 5 * 128;
	static constexpr size_t SP_OFF = 4 * 4096 + 192 + 64 + 192; // structured vectors P | R | qa | c | dd
	static constexpr size_t KCC_OFF = SP_OFF + 5 * 64;          // 64 states: constant tables of k_kcol2_struct, 2 x (2*64 + 9*64) doubles (the 128-state layout's space, unused here)
	// tables
	double *d_f = nullptr, *d_b = nullptr, *d_s = nullptr, *d_sb = nullptr;
	int64_t tab_bins = 0; bool have_b = false;
	// exact outputs
	double *d_segA = nullptr, *d_segE = nullptr, *d_segA0 = nullptr, *d_chk = nullptr;
	int seg_cap = 0;
	std::vector<double> h_segA, h_segE, h_segA0, h_chk, h_s;
	// fast
	std::vector<Chunk> chunks;
	Chunk *d_chunks = nullptr;
	int chunk_cap = 0, chunk_used = 0;
	double *d_entry = nullptr, *d_bentry = nullptr, *d_bexit = nullptr, *d_Cpart = nullptr, *d_Epart = nullptr,
	       *d_LLpart = nullptr;
	size_t cpart_cap = 0; // doubles in d_Cpart (api_fast.hip enqueue_fast)
	int *d_dirty = nullptr, *d_cnt = nullptr, *h_cnt = nullptr, *d_touch = nullptr;
	hipStream_t stream2 = nullptr, stream3 = nullptr, stream4 = nullptr;
	hipEvent_t evx[14] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
	bool timing_two_launches = false; // the fused back half ran as two launches (lists A and B): evx[11] / evx[12] sit between them
	int n_long_f = 0, n_long_b = 0, n_mem_f = 0, n_mem_b = 0;
	int n_B_b = 0, n_list_a = 0, n_list_b = 0; // two-phase plan: trailing backward items that start from above; tile lists of the fused back half
	int items_two_phase = -1;  // what the current item lists were built for
	int runs_late = 1;         // "runs_late": two-phase plan, 1 = run tiles go to the second launch of the fused back half
	bool runs_in_b = false;    // two-phase plan: every tile of a glued run is in the second list of the fused back half (build_items)
	int *d_ftiles = nullptr;   // [2 * (n_tiles + 4)] tile lists A | B of the fused back half (bit 30: start from the tile above)
	FastReport report = {0, 0, 0, 0, 1, 0, 0};
	double *d_stage = nullptr, *d_stats = nullptr;
	unsigned long long *d_warm = nullptr;
	double warm_err[2] = {0, 0};
	// runtime
	hipStream_t stream = nullptr;
	hipEvent_t ev[10] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
	double last_ms[7] = {0, 0, 0, 0, 0, 0, 0};
	bool timing_valid = false;
	// batch (psmc_hip_estep_batch)
	int64_t batch_bins = 0;            // "batch_bins": table bins per launch group of the exact batch (0 = from free memory)
	int exact_refwd = -1;              // "exact_refwd": exact batch, 64 states: 1 = no f table, the expect pass recomputes the forward sweep (twice the replicates
	                                   // per group at +0.4 us per bin of the longest segment); 0 = tables for f and b; -1 = only when the tables of all
	                                   // replicates do not fit one launch group (api_batch.hip batch_refwd)
	int32_t *d_bw_seg = nullptr, *d_bw_par = nullptr; int64_t *d_bw_tab = nullptr; size_t bw_cap = 0; // work list of a group
	double *d_bpar = nullptr; size_t bpar_cap = 0; // [n_rep][PAR_LEN] parameter blocks of a batch call
	int *d_cu_mask = nullptr;          // k_expect_exact_rf2: one word per compute unit (which role order its resident work-groups took), 4096 words
	double *d_lkp = nullptr; size_t lkp_cap = 0; std::vector<double> h_lkp; // hmm_lk's logged products of a launch's entries (k_lk_products)
	int64_t *d_lkoff = nullptr; size_t lkoff_cap = 0;                     // ... and where each entry's begin
	double *d_s_all = nullptr; size_t s_all_cap = 0; // exact batch without the f table, several groups: the scale factors of ALL replicates (one forward pass)
	int batch_sort = 1;                // "batch_sort": the exact batch deals ENTRIES to its launches longest first (api_batch.hip); 0 = replicate-major order
	int batch_tailfill = 1;            // "batch_tailfill": the exact batch puts the shortest entries into the spare slots of the memory-bound launches when that saves a launch
	int batch_major = 1;               // "batch_major": blocks no longer than the dominant trunk length keep replicate order (replicates complete launch by launch)
	int batch_first = 0;               // "batch_first": the next batch calls' replicate 0 is replicate batch_first of the context (fast mode: which kept plan it uses)
	double *d_b2 = nullptr; int64_t b2_bins = 0, b2_alloc = 0; // exact batch without the f table: a second chunk of b table, allocated when memory became free later (api_batch.hip)
	int64_t reserved_cap = 0;          // ... and the per-launch capacity it sized the tables for
	int reserved_refwd = -1;           // what psmc_hip_reserve_batch_tables decided about the f table (-1: nothing reserved): the batches that follow keep it
	int cu_first = 0, cu_count = 0;    // psmc_hip_set_cu_range: the streams of this context are masked to these compute units (0: the whole device)
	int last_batch_groups = 0;
	bool tables_batch = false;         // the tables hold the slots of a batch group, not the segments at their own offsets
	// fast batch: one plan-holding child per replicate; children share the parent's streams, events, parameter
	// staging, observations and TABLES (they run one after the other)
	psmc_hip_ctx *x_twin = nullptr;    // fast batch: an exact-mode context over the same observations, made when a replicate's fast E-step does not converge (api_batch.hip batch_exact_once)
	int n_exact_fallbacks = 0;
	psmc_hip_ctx *parent = nullptr;
	std::vector<psmc_hip_ctx *> kids;
	// ... and what the replicates of one parent have LEARNED is shared through the parent (round 4): they tile every segment with the
	// same tile length share_T, so a tile is (segment, index) in all of them, and a replicate that plans starts from the glue flags and
	// warm-ups its predecessors needed (the slow regions belong to the data: the first E-step of replicate 2..R no longer repeats the
	// repair rounds of replicate 1).  Filled by learn_groups of the children, read by their plan_fast (api_fast.hip).
	int share_learn = 1;       // "share_learn": 0 = every replicate of a fast batch learns for itself (and tiles its own selection)
	int share_T = 0;
	std::vector<std::vector<uint8_t>> sh_glue_f, sh_glue_b;   // [segment][tile index]
	std::vector<std::vector<int32_t>> sh_wf, sh_wb;
	std::vector<int32_t> chunk_seg, chunk_idx;                // of every tile of the current plan
	// PSMC_HIP_DEBUG_TIMES: host seconds spent in the pieces of the fast E-steps of this context and its batch children since the last
	// report -- [0] replicate context creation, [1] select, [2] plan_fast (tiling + per-plan allocations), [3] build_items,
	// [4] launch_fast (kernels + verify / repair rounds, synchronous), [5] result read-back; and [6] repair rounds, [7] repaired tiles
	double dbg_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
};
inline double dbg_now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
inline psmc_hip_ctx *dbg_root(psmc_hip_ctx *c) { return c->parent ? c->parent : c; }

inline int fail(psmc_hip_ctx *c, int code, const char *what, hipError_t e = hipSuccess)
{
	if (c) {
		c->err = what;
		if (e != hipSuccess) { c->err += ": "; c->err += hipGetErrorString(e); }
	}
	return code;
}
#define HIPCHK(c, call)                                                     \
	do {                                                                    \
		hipError_t e__ = (call);                                            \
		if (e__ != hipSuccess) return fail((c), PSMC_HIP_EDEVICE, #call, e__); \
	} while (0)

template <class T> inline int dev_alloc(psmc_hip_ctx *c, T **p, size_t n)
{
	if (*p) { (void)hipFree(*p); *p = nullptr; }
	if (n == 0) n = 1;
	hipError_t e = hipMalloc((void **)p, n * sizeof(T));
	if (e != hipSuccess) { *p = nullptr; return fail(c, PSMC_HIP_ENOMEM, "hipMalloc", e); }
	// PSMC_HIP_POISON=1 (tests): fresh device memory is usually zero, recycled memory is not -- fill every allocation with
	// 0xFF bytes (NaN as double, -1 as int) so that anything that depends on memory nobody wrote shows up at once.
	// PSMC_HIP_POISON=vary: a different finite garbage value per allocation (bytes 0x3B..0x42: doubles from 1e-23 to 1e5),
	// so that two contexts with the same call history disagree if anything reads memory nobody wrote
	static const char *poison = getenv("PSMC_HIP_POISON");
	static int poison_count = 0;
	if (poison) { // the fill runs on the null stream, the kernels on non-blocking streams: finish it before anybody writes results there
		(void)hipMemset(*p, strcmp(poison, "vary") == 0 ? 0x3B + (poison_count++ % 8) : 0xFF, n * sizeof(T));
		(void)hipDeviceSynchronize();
	}
	return 0;
}

// ---- shared between the translation units (defined in the file named)
int  set_segments_common(psmc_hip_ctx *c, int n_seg, const int32_t *L);                                     // api.hip
bool fill_params(const psmc_hip_ctx *c, const double *a, const double *e, const double *a0, double *dst);   // api.hip
int  stage_params(psmc_hip_ctx *c, const double *a, const double *e, const double *a0, hipStream_t st);     // api.hip
int  ensure_tables(psmc_hip_ctx *c, bool need_b, int64_t want_bins = 0, bool need_f = true);                // api.hip
bool fused_counts(const psmc_hip_ctx *c);                                                                   // api.hip
bool kcol2_on(const psmc_hip_ctx *c);                                                                       // api.hip
void fill_common(psmc_hip_ctx *c, EstepLaunch &p, hipStream_t st, const double *par_base = nullptr);        // api.hip
void collect_timing(psmc_hip_ctx *c);                                                                       // api.hip
double host_lk(const double *s, int L);                                                                     // api.hip
int  ensure_seg_outputs(psmc_hip_ctx *c, int nw);                                                           // api.hip
int  estep_exact(psmc_hip_ctx *c, const double *a, const double *e, const double *a0, double *A, double *E, double *A0, double *LL, double *chk); // api.hip
int  ensure_fast_buffers(psmc_hip_ctx *c);                                                                  // api_fast.hip
int  auto_tile_len(const psmc_hip_ctx *c, int64_t bins, size_t n_work, bool structured);                    // api_fast.hip
int  enqueue_fast(psmc_hip_ctx *c, const double *a, const double *e, const double *a0, double *d_out, hipStream_t st); // api_fast.hip
int  read_warm(psmc_hip_ctx *c, hipStream_t st);                                                            // api_fast.hip
int  estep_fast(psmc_hip_ctx *c, const double *a, const double *e, const double *a0, double *A, double *E, double *A0, double *LL, double *chk); // api_fast.hip
