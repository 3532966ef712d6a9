// api.hip -- C-ABI of libpsmc_hip.so (see include/psmc_hip.h): context,
// segment upload, tile planning, launch orchestration and the host-side
// ordered reductions of the exact mode.  Built with -ffp-contract=off so that
// no host or device expression in this file is ever fused.
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
	int chunk = 0, warmup = 3072, max_rounds = 4096, rep_impl = 1, expect_impl = 1, n_sub = 6, target_waves = 1536, overlap = 1;
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
	                           // fewer vector instructions per tile-step, half the waves).  -1 = with the factored statistics only -- measured (round 4, genome):
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
	                           // exchanged through LDS), 1 = k_bwd_count8_struct (four waves redo the sweep of four tiles), 0 = unfused
	int count_group = 4;       // tiles per work-group of the fused back half, what the tile lists are padded to (build_items)
	int learn = 1;             // "learn": glue tiles that needed a repair to their neighbour for the following E-steps
	int warm_shift = 1;        // "warm_shift": before that, once, give such a tile a warm-up of warmup << warm_shift bins (0: glue at once)
	bool chunks_dirty = false; // a tile's warm-up changed: d_chunks is stale
	int group_cap = 131072;    // "group_cap": longest run of glued tiles, in bins
	int *d_items = nullptr;    // items_f | items_b | ritems_f | ritems_b, 2*n_chunks ints each
	int *h_ritems = nullptr;   // pinned + device-mapped, 2 * 2*n_chunks ints
	int *m_ritems = nullptr, *m_cnt = nullptr; // device views of h_ritems / h_cnt
	std::vector<uint8_t> glue_f, glue_b; // glue_f[b]: tile b continues the forward item of b-1; glue_b[b]: b continues b+1's backward item
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
	FastReport report = {0, 0, 0, 0, 1};
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
	                                   // per group at +0.4 us per bin of the longest segment); 0 = tables for f and b; -1 = 1
	int32_t *d_bw_seg = nullptr, *d_bw_par = nullptr; int64_t *d_bw_tab = nullptr; size_t bw_cap = 0; // work list of a group
	double *d_bpar = nullptr; size_t bpar_cap = 0; // [n_par][PAR_LEN] parameter blocks of a group
	int last_batch_groups = 0;
	bool tables_batch = false;         // the tables hold the slots of a batch group, not the segments at their own offsets
	// fast batch: one plan-holding child per replicate; children share the parent's streams, events, parameter
	// staging, observations and TABLES (they run one after the other)
	psmc_hip_ctx *parent = nullptr;
	std::vector<psmc_hip_ctx *> kids;
};

static int fail(psmc_hip_ctx *c, int code, const char *what, hipError_t e = hipSuccess)
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

template <class T> static int dev_alloc(psmc_hip_ctx *c, T **p, size_t n)
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

// The fast E-step keeps four streams busy (forward chain, backward chain, counts, walks) beside the
// caller's.  HIP hands a process 4 hardware queues by default and streams beyond that share one -- a
// kernel then waits for an unrelated one.  Ask for 8 before the runtime starts (no effect, and no harm,
// if the host program initialised HIP earlier or set the variable itself).
__attribute__((constructor)) static void psmc_hip_more_queues() { setenv("GPU_MAX_HW_QUEUES", "8", 0); }

static void destroy_kids(psmc_hip_ctx *c);

extern "C" int psmc_hip_device_count(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) return 0;
	return n;
}

extern "C" const char *psmc_hip_strerror(int err)
{
	switch (err) {
	case PSMC_HIP_OK: return "ok";
	case PSMC_HIP_EINVAL: return "invalid argument";
	case PSMC_HIP_ENOMEM: return "out of memory";
	case PSMC_HIP_EDEVICE: return "HIP runtime error";
	case PSMC_HIP_ENOTSUP: return "not supported in this build";
	case PSMC_HIP_ESTATE: return "call order violated";
	case PSMC_HIP_ECONVERGE: return "fast-mode tile boundaries did not converge";
	default: return "unknown error";
	}
}
extern "C" const char *psmc_hip_last_error(const psmc_hip_ctx *ctx) { return ctx ? ctx->err.c_str() : ""; }

extern "C" int psmc_hip_create(psmc_hip_ctx **out, int n_states, int device, int mode)
{
	if (!out) return PSMC_HIP_EINVAL;
	*out = nullptr;
	if (n_states < 1) return PSMC_HIP_EINVAL;
	if (mode != PSMC_HIP_MODE_EXACT && mode != PSMC_HIP_MODE_FAST) return PSMC_HIP_EINVAL;
	if (n_states > 128) return PSMC_HIP_ENOTSUP; // fast mode beyond 64 states: structured matrices only (checked per E-step)
	int nd = psmc_hip_device_count();
	if (nd <= 0 || device < 0 || device >= nd) return PSMC_HIP_EDEVICE;
	if (hipSetDevice(device) != hipSuccess) return PSMC_HIP_EDEVICE;
	psmc_hip_ctx *c = new (std::nothrow) psmc_hip_ctx();
	if (!c) return PSMC_HIP_ENOMEM;
	c->n = n_states; c->ns = n_states > 64 ? 128 : 64; c->device = device; c->mode = mode;
	if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return PSMC_HIP_EDEVICE; }
	if (hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking) != hipSuccess) { delete c; return PSMC_HIP_EDEVICE; }
	if (hipStreamCreateWithFlags(&c->stream3, hipStreamNonBlocking) != hipSuccess) { delete c; return PSMC_HIP_EDEVICE; }
	if (hipStreamCreateWithFlags(&c->stream4, hipStreamNonBlocking) != hipSuccess) { delete c; return PSMC_HIP_EDEVICE; }
	if (hipStreamCreateWithFlags(&c->stream5, hipStreamNonBlocking) != hipSuccess) { delete c; return PSMC_HIP_EDEVICE; }
	for (int i = 0; i < 14; ++i)
		if (hipEventCreate(&c->evx[i]) != hipSuccess) { delete c; return PSMC_HIP_EDEVICE; }
	for (int i = 0; i < 10; ++i)
		if (hipEventCreate(&c->ev[i]) != hipSuccess) { delete c; return PSMC_HIP_EDEVICE; }
	if (hipHostMalloc((void **)&c->h_par, psmc_hip_ctx::PAR_LEN * sizeof(double), hipHostMallocDefault) != hipSuccess ||
	    hipMalloc((void **)&c->d_par, psmc_hip_ctx::PAR_LEN * sizeof(double)) != hipSuccess) {
		psmc_hip_destroy(c);
		return PSMC_HIP_ENOMEM;
	}
	// PSMC_HIP_OPTIONS="key=value,key=value": options for every context of the process (A/B of a whole program or test
	// run under another plan without touching its code; unknown keys are an error so that a typo cannot pass for a result)
	if (const char *env = getenv("PSMC_HIP_OPTIONS")) {
		std::string all(env);
		size_t at = 0;
		while (at < all.size()) {
			size_t end = all.find(',', at);
			if (end == std::string::npos) end = all.size();
			const std::string kv = all.substr(at, end - at);
			const size_t eq = kv.find('=');
			at = end + 1;
			if (kv.empty()) continue;
			// the value must be a whole number token ("chunk=abc" or "kc_min=" must not pass for 0), and the message names the pair
			char *endp = nullptr;
			const double val = eq == std::string::npos ? 0.0 : strtod(kv.c_str() + eq + 1, &endp);
			const bool parsed = eq != std::string::npos && eq + 1 < kv.size() && endp && *endp == '\0';
			if (parsed && kv.compare(0, eq, "rccl") == 0) continue; // a group-level key (psmc_hip_group_set_option): not this context's business
			if (!parsed || psmc_hip_set_option(c, kv.substr(0, eq).c_str(), val) != PSMC_HIP_OK) {
				fprintf(stderr, "[psmc_hip] PSMC_HIP_OPTIONS: bad entry \"%s\" (unknown key, value out of range, or not a number)\n", kv.c_str());
				psmc_hip_destroy(c);
				return PSMC_HIP_EINVAL;
			}
		}
	}
	*out = c;
	return PSMC_HIP_OK;
}

static void destroy_kids(psmc_hip_ctx *c)
{
	for (psmc_hip_ctx *k : c->kids) psmc_hip_destroy(k);
	c->kids.clear();
}

extern "C" void psmc_hip_destroy(psmc_hip_ctx *c)
{
	if (!c) return;
	(void)hipSetDevice(c->device);
	destroy_kids(c);
	if (c->parent) { // a batch child owns its plan only: streams, events, staging, observations and tables are the parent's
		// (d_seg*: a child that took the exact fallback of psmc_hip_estep -- 65..128 states, a matrix without the PSMC form)
		void *mine[] = {c->d_seg_off, c->d_seg_len, c->d_work, c->d_chunks, c->d_entry, c->d_bexit, c->d_Cpart, c->d_Epart, c->d_LLpart,
		                c->d_stage, c->d_stats, c->d_warm, c->d_bentry, c->d_dirty, c->d_cnt, c->d_gate, c->d_touch, c->d_items, c->d_ftiles, c->d_Kcol,
		                c->d_segA, c->d_segE, c->d_segA0, c->d_chk};
		for (void *p : mine) if (p) (void)hipFree(p);
		if (c->h_cnt) (void)hipHostFree(c->h_cnt);
		if (c->h_ritems) (void)hipHostFree(c->h_ritems);
		delete c;
		return;
	}
	if (c->stream) (void)hipStreamSynchronize(c->stream);
	if (c->stream2) (void)hipStreamSynchronize(c->stream2);
	if (c->stream3) (void)hipStreamSynchronize(c->stream3);
	if (c->stream4) (void)hipStreamSynchronize(c->stream4);
	if (c->stream5) (void)hipStreamSynchronize(c->stream5);
	if (!c->obs_borrowed && c->d_obs) (void)hipFree(c->d_obs);
	void *ptrs[] = {c->d_seg_off, c->d_seg_len, c->d_work, c->d_par, c->d_f, c->d_b, c->d_s, c->d_segA, c->d_segE,
	                c->d_segA0, c->d_chk, c->d_chunks, c->d_entry, c->d_bexit, c->d_Cpart, c->d_Epart, c->d_LLpart,
	                c->d_stage, c->d_stats, c->d_warm, c->d_bentry, c->d_dirty, c->d_cnt, c->d_gate, c->d_touch, c->d_sb, c->d_items, c->d_ftiles, c->d_Kcol,
	                c->d_bw_seg, c->d_bw_par, c->d_bw_tab, c->d_bpar};
	for (void *p : ptrs) if (p) (void)hipFree(p);
	if (c->h_par) (void)hipHostFree(c->h_par);
	if (c->h_cnt) (void)hipHostFree(c->h_cnt);
	if (c->h_ritems) (void)hipHostFree(c->h_ritems);
	for (int i = 0; i < 10; ++i) if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
	for (int i = 0; i < 14; ++i) if (c->evx[i]) (void)hipEventDestroy(c->evx[i]);
	if (c->stream4) (void)hipStreamDestroy(c->stream4);
	if (c->stream5) (void)hipStreamDestroy(c->stream5);
	if (c->stream2) (void)hipStreamDestroy(c->stream2);
	if (c->stream3) (void)hipStreamDestroy(c->stream3);
	if (c->stream) (void)hipStreamDestroy(c->stream);
	delete c;
}

extern "C" int psmc_hip_set_option(psmc_hip_ctx *c, const char *key, double v)
{
	if (!c || !key) return PSMC_HIP_EINVAL;
	std::string k(key);
	destroy_kids(c); // replicate contexts of a batch copied the options when they were made: start them afresh
	if (k == "chunk") { if (v < 0) return PSMC_HIP_EINVAL; c->chunk = (int)v; c->plan_dirty = true; }
	else if (k == "warmup") { if (v < 0) return PSMC_HIP_EINVAL; c->warmup = (int)v; c->plan_dirty = true; }
	else if (k == "max_rounds") c->max_rounds = (int)v;
	else if (k == "structured") { c->struct_opt = v != 0 ? 1 : 0; }
	else if (k == "learn") { c->learn = v != 0 ? 1 : 0; }
	else if (k == "warm_shift") { if (v < 0 || v > 4) return PSMC_HIP_EINVAL; c->warm_shift = (int)v; c->warm_shift_set = true; c->plan_dirty = true; }
	else if (k == "runs_late") { c->runs_late = v != 0 ? 1 : 0; c->items_dirty = true; }
	else if (k == "merge1") { if (v < -1 || v > 1) return PSMC_HIP_EINVAL; c->merge1 = (int)v; c->plan_dirty = true; }
	else if (k == "lanes8") { if (v < -1 || v > 1) return PSMC_HIP_EINVAL; c->lanes8 = (int)v; }
	else if (k == "gate") { if (v < -1 || v > 1) return PSMC_HIP_EINVAL; c->gate = (int)v; }
	else if (k == "coarse") { if (v < -1 || v > 16) return PSMC_HIP_EINVAL; c->coarse = (int)v; c->plan_dirty = true; c->items_dirty = true; }
	else if (k == "merge_order") { if (v < -1 || v > 1) return PSMC_HIP_EINVAL; c->merge_order = (int)v; }
	else if (k == "kcol_prio") { if (v < 0 || v > 2) return PSMC_HIP_EINVAL; c->kcol_prio = (int)v; }
	else if (k == "kc_sub") { if (v < 1 || v > 16) return PSMC_HIP_EINVAL; c->kc_sub = (int)v; c->kc_sub_set = true; c->plan_dirty = true; c->items_dirty = true; }
	else if (k == "two_phase") { if (v != -1 && v != 0 && v != 2) return PSMC_HIP_EINVAL; c->two_phase = (int)v; c->plan_dirty = true; c->items_dirty = true; }
	else if (k == "kc_div") { if (v < 1) return PSMC_HIP_EINVAL; c->kc_div = (int)v; c->items_dirty = true; }
	else if (k == "kc_min") { if (v < -1) return PSMC_HIP_EINVAL; c->kc_min = (int)v; c->items_dirty = true; }
	else if (k == "ckpt") { c->ckpt = v != 0 ? 1 : 0; }
	else if (k == "fuse") { c->fuse = v != 0 ? 1 : 0; c->plan_dirty = true; }
	else if (k == "fuse128") { if (v < 0 || v > 2) return PSMC_HIP_EINVAL; c->fuse128 = (int)v; c->plan_dirty = true; c->items_dirty = true; }
	else if (k == "group_cap") { if (v < 0) return PSMC_HIP_EINVAL; c->group_cap = (int)v; c->items_dirty = true; }
	else if (k == "struct_tiles") { if (v < 1) return PSMC_HIP_EINVAL; c->struct_tiles = (int)v; c->struct_tiles_set = true; c->plan_dirty = true; }
	else if (k == "exact_refwd") { if (v < -1 || v > 1) return PSMC_HIP_EINVAL; c->exact_refwd = (int)v; }
	else if (k == "batch_bins") { if (v < 0) return PSMC_HIP_EINVAL; c->batch_bins = (int64_t)v; }
	else if (k == "overlap") c->overlap = v != 0 ? 1 : 0;
	else if (k == "warm_tol") c->warm_tol = v;
	else if (k == "rep_impl") c->rep_impl = v != 0 ? 1 : 0;
	else if (k == "expect_impl") c->expect_impl = v != 0 ? 1 : 0;
	else if (k == "n_sub") { if (v < 1 || v > 64) return PSMC_HIP_EINVAL; c->n_sub = (int)v; c->plan_dirty = true; }
	else if (k == "target_waves") { if (v < 1) return PSMC_HIP_EINVAL; c->target_waves = (int)v; c->plan_dirty = true; }
	else return PSMC_HIP_EINVAL;
	return PSMC_HIP_OK;
}

static int set_segments_common(psmc_hip_ctx *c, int n_seg, const int32_t *L)
{
	destroy_kids(c); // batch children hold plans over the previous segments
	c->n_seg = n_seg;
	c->L.assign(L, L + n_seg);
	int rc;
	if ((rc = dev_alloc(c, &c->d_seg_off, (size_t)n_seg))) return rc;
	if ((rc = dev_alloc(c, &c->d_seg_len, (size_t)n_seg))) return rc;
	HIPCHK(c, hipMemcpy(c->d_seg_off, c->off.data(), sizeof(int64_t) * n_seg, hipMemcpyHostToDevice));
	HIPCHK(c, hipMemcpy(c->d_seg_len, c->L.data(), sizeof(int32_t) * n_seg, hipMemcpyHostToDevice));
	std::vector<int32_t> all(n_seg);
	for (int i = 0; i < n_seg; ++i) all[i] = i;
	return psmc_hip_select(c, n_seg, all.data());
}

extern "C" int psmc_hip_load_segments(psmc_hip_ctx *c, int n_seg, const uint8_t *const *seq, const int32_t *L)
{
	if (!c || n_seg < 1 || !seq || !L) return fail(c, PSMC_HIP_EINVAL, "load_segments: bad argument");
	HIPCHK(c, hipSetDevice(c->device));
	c->off.resize(n_seg);
	int64_t tot = 0;
	for (int i = 0; i < n_seg; ++i) {
		if (L[i] < 1 || !seq[i]) return fail(c, PSMC_HIP_EINVAL, "load_segments: empty segment");
		c->off[i] = tot;
		tot += ((int64_t)L[i] + 63) & ~(int64_t)63;
	}
	c->total = tot;
	std::vector<uint8_t> host((size_t)tot + 256, 2);
	for (int i = 0; i < n_seg; ++i) {
		for (int32_t j = 0; j < L[i]; ++j)
			if (seq[i][j] > 2) return fail(c, PSMC_HIP_EINVAL, "load_segments: symbol outside {0,1,2}");
		memcpy(host.data() + c->off[i], seq[i], (size_t)L[i]);
	}
	if (!c->obs_borrowed && c->d_obs) { (void)hipFree(c->d_obs); }
	c->d_obs = nullptr; c->obs_borrowed = false;
	int rc;
	if ((rc = dev_alloc(c, &c->d_obs, host.size()))) return rc;
	HIPCHK(c, hipMemcpy(c->d_obs, host.data(), host.size(), hipMemcpyHostToDevice));
	return set_segments_common(c, n_seg, L);
}

extern "C" int psmc_hip_load_segments_device(psmc_hip_ctx *c, int n_seg, const void *d_obs, const int64_t *off,
                                             const int32_t *L)
{
	if (!c || n_seg < 1 || !d_obs || !off || !L) return fail(c, PSMC_HIP_EINVAL, "load_segments_device: bad argument");
	HIPCHK(c, hipSetDevice(c->device));
	int64_t end = 0;
	for (int i = 0; i < n_seg; ++i) {
		if (L[i] < 1 || (off[i] & 63) || off[i] < end) return fail(c, PSMC_HIP_EINVAL, "load_segments_device: bad layout");
		end = off[i] + (((int64_t)L[i] + 63) & ~(int64_t)63);
	}
	if (!c->obs_borrowed && c->d_obs) (void)hipFree(c->d_obs);
	c->d_obs = (uint8_t *)d_obs; c->obs_borrowed = true;
	c->off.assign(off, off + n_seg);
	c->total = end;
	return set_segments_common(c, n_seg, L);
}

extern "C" int psmc_hip_select(psmc_hip_ctx *c, int n_sel, const int32_t *idx)
{
	if (!c || n_sel < 1 || !idx) return fail(c, PSMC_HIP_EINVAL, "select: bad argument");
	if (c->n_seg < 1) return fail(c, PSMC_HIP_ESTATE, "select: no segments loaded");
	std::vector<int32_t> pos(c->n_seg, -1);
	c->sel.assign(idx, idx + n_sel);
	c->work.clear(); c->mult.clear(); c->sel2work.resize(n_sel);
	for (int i = 0; i < n_sel; ++i) {
		if (idx[i] < 0 || idx[i] >= c->n_seg) return fail(c, PSMC_HIP_EINVAL, "select: index out of range");
		if (pos[idx[i]] < 0) { pos[idx[i]] = (int32_t)c->work.size(); c->work.push_back(idx[i]); c->mult.push_back(0); }
		c->sel2work[i] = pos[idx[i]];
		c->mult[pos[idx[i]]]++;
	}
	HIPCHK(c, hipSetDevice(c->device));
	int rc;
	if ((rc = dev_alloc(c, &c->d_work, c->work.size()))) return rc;
	HIPCHK(c, hipMemcpy(c->d_work, c->work.data(), sizeof(int32_t) * c->work.size(), hipMemcpyHostToDevice));
	c->plan_dirty = true;
	return PSMC_HIP_OK;
}

// Does a[][] have the PSMC form  a[k][l] = P_k qa_l (l<k),  R_k c_l (l>k)  (core.c:112-122)?
// Numerical factorisation with qa_0 = c_{n-1} = 1, then a check of EVERY off-diagonal entry to
// 64 ulp and of dd = diag - P.qa - R.c >= 0.  sp = P | R | qa | c | dd (64 each, zero padded).
static bool factor_structure(int n, int S, const double *a /* stride S */, double *sp /* 5 * S */)
{
	double *P = sp, *R = sp + S, *qa = sp + 2 * S, *cc = sp + 3 * S, *dd = sp + 4 * S;
	memset(sp, 0, 5 * (size_t)S * sizeof(double));
	if (n < 3) return false;
	const double pl = a[(n - 1) * S + 0], pu = a[0 * S + (n - 1)];
	if (!(pl > 1e-280) || !(pu > 1e-280)) return false;
	for (int l = 0; l < n - 1; ++l) qa[l] = a[(n - 1) * S + l] / pl;
	for (int k = 1; k < n; ++k) P[k] = a[k * S + 0];
	for (int l = 1; l < n; ++l) cc[l] = a[0 * S + l] / pu;
	for (int k = 0; k < n - 1; ++k) R[k] = a[k * S + (n - 1)];
	const double tol = 64 * 2.220446049250313e-16;
	for (int k = 0; k < n; ++k) {
		for (int l = 0; l < n; ++l) {
			if (l == k) continue;
			const double v = a[k * S + l], w = l < k ? P[k] * qa[l] : R[k] * cc[l];
			if (!(fabs(v - w) <= tol * fabs(v) + 1e-290)) return false;
		}
		dd[k] = a[k * S + k] - P[k] * qa[k] - R[k] * cc[k];
		if (!(dd[k] >= 0.0)) return false;
	}
	return true;
}

// pad the HMM parameters to 64 states and build aeT[b][l*64+k] = e[b][l]*a[k][l]
// (hmm_pre_backward, khmm.c:194-206: one rounding per product), then upload.
// Constant tables of k_kcol2_struct (transfer matrices with one column per lane): per direction, S = padded states,
//   mS | mP | { wS.e | wP.e | dd.e } for the symbols 0, 1, 2        (11 S doubles; forward: mS = P, wS = qa, mP = R, wP = c;
// backward: mS = c, wS = R, mP = qa, wP = P -- the roles load_struct_par gives the five vectors sp = P | R | qa | c | dd)
static void fill_kcc(int S, const double *sp, const double *e3 /* 3 rows of S */, double *out /* 2 * 11 S */)
{
	const double *P = sp, *R = sp + S, *qa = sp + 2 * S, *cv = sp + 3 * S, *dd = sp + 4 * S;
	for (int dir = 0; dir < 2; ++dir) {
		double *t = out + (size_t)dir * 11 * S;
		const double *mS = dir == 0 ? P : cv, *wS = dir == 0 ? qa : R, *mP = dir == 0 ? R : qa, *wP = dir == 0 ? cv : P;
		for (int k = 0; k < S; ++k) { t[k] = mS[k]; t[S + k] = mP[k]; }
		for (int sy = 0; sy < 3; ++sy)
			for (int k = 0; k < S; ++k) {
				const double ev = e3[sy * S + k];
				double *u = t + 2 * S + (size_t)sy * 3 * S;
				u[k] = wS[k] * ev; u[S + k] = wP[k] * ev; u[2 * S + k] = dd[k] * ev;
			}
	}
}

// host part: one parameter block (PAR_LEN doubles) at dst; returns whether the matrix has the PSMC form (fast mode)
static bool fill_params(const psmc_hip_ctx *c, const double *a, const double *e, const double *a0, double *dst)
{
	const int n = c->n;
	if (c->ns == 128) { // a | aT | e(3) | a0, stride 128; the e*a products are formed on the device
		double *pa = dst, *pt = pa + 16384, *pe = pt + 16384, *pa0 = pe + 384;
		memset(pa, 0, psmc_hip_ctx::PAR_LEN * sizeof(double));
		for (int k = 0; k < n; ++k) {
			for (int l = 0; l < n; ++l) { pa[k * 128 + l] = a[k * n + l]; pt[l * 128 + k] = a[k * n + l]; }
			pe[k] = e[k]; pe[128 + k] = e[n + k];
			pa0[k] = a0[k];
		}
		for (int k = 0; k < 128; ++k) pe[256 + k] = 1.0; // khmm.c:21
		if (c->mode == PSMC_HIP_MODE_FAST) {
			double *pre = pa + psmc_hip_ctx::RE128_OFF;
			for (int i = 0; i < 384; ++i) pre[i] = pe[i] > 0.0 ? 1.0 / pe[i] : 0.0;
			const bool st = c->struct_opt && factor_structure(n, 128, pa, pa + psmc_hip_ctx::SP128_OFF);
			if (st) fill_kcc(128, pa + psmc_hip_ctx::SP128_OFF, pe, pa + psmc_hip_ctx::KCC128_OFF);
			return st;
		}
		return false;
	}
	double *pa = dst, *pae = pa + 4096, *pe = pae + 3 * 4096, *pa0 = pe + 3 * 64, *pre = pa0 + 64;
	memset(pa, 0, psmc_hip_ctx::PAR_LEN * sizeof(double));
	for (int k = 0; k < n; ++k) {
		for (int l = 0; l < n; ++l) pa[k * 64 + l] = a[k * n + l];
		pe[k] = e[k]; pe[64 + k] = e[n + k];
		pa0[k] = a0[k];
	}
	for (int k = 0; k < 64; ++k) pe[128 + k] = 1.0; // khmm.c:21
	for (int i = 0; i < 192; ++i) pre[i] = pe[i] > 0.0 ? 1.0 / pe[i] : 0.0; // fast-mode expect divides the emission back out
	for (int b = 0; b < 3; ++b)
		for (int l = 0; l < 64; ++l)
			for (int k = 0; k < 64; ++k) pae[b * 4096 + l * 64 + k] = pe[b * 64 + l] * pa[k * 64 + l];
	const bool st = c->mode == PSMC_HIP_MODE_FAST && c->struct_opt && factor_structure(n, 64, pa, pa + psmc_hip_ctx::SP_OFF);
	if (st) fill_kcc(64, pa + psmc_hip_ctx::SP_OFF, pe, pa + psmc_hip_ctx::KCC_OFF);
	return st;
}

static int stage_params(psmc_hip_ctx *c, const double *a, const double *e, const double *a0, hipStream_t st)
{
	HIPCHK(c, hipStreamSynchronize(st)); // previous async copy out of the pinned staging buffer
	c->use_struct = fill_params(c, a, e, a0, c->h_par);
	HIPCHK(c, hipMemcpyAsync(c->d_par, c->h_par, psmc_hip_ctx::PAR_LEN * sizeof(double), hipMemcpyHostToDevice, st));
	return 0;
}

static int ensure_tables(psmc_hip_ctx *c, bool need_b, int64_t want_bins = 0, bool need_f = true)
{
	if (c->parent) { // a batch child works in its parent's tables (same segments, one E-step at a time)
		int rc = ensure_tables(c->parent, need_b);
		c->d_f = c->parent->d_f; c->d_b = c->parent->d_b; c->d_s = c->parent->d_s; c->d_sb = c->parent->d_sb;
		c->tab_bins = c->parent->tab_bins; c->have_b = c->parent->have_b;
		return rc;
	}
	const int64_t bins = std::max(c->total, want_bins) + 128;
	int rc;
	if (c->tab_bins < bins) { // grow: everything goes, and comes back as needed
		if (c->d_f) { (void)hipFree(c->d_f); c->d_f = nullptr; }
		if (c->d_b) { (void)hipFree(c->d_b); c->d_b = nullptr; }
		c->have_b = false;
		if ((rc = dev_alloc(c, &c->d_s, (size_t)bins))) { c->tab_bins = 0; return rc; }
		if (c->mode == PSMC_HIP_MODE_FAST && (rc = dev_alloc(c, &c->d_sb, (size_t)bins))) { c->tab_bins = 0; return rc; }
		c->tab_bins = bins;
	}
	// the exact batch with the recomputed forward sweep keeps no f table: give its memory to the replicates
	if (!need_f && c->d_f) { (void)hipFree(c->d_f); c->d_f = nullptr; }
	if (need_f && !c->d_f && (rc = dev_alloc(c, &c->d_f, (size_t)c->tab_bins * c->ns))) return rc;
	if (need_b && !c->have_b) {
		if ((rc = dev_alloc(c, &c->d_b, (size_t)c->tab_bins * c->ns))) return rc;
		c->have_b = true;
	}
	return 0;
}

// fused backward sweep + counts: structured matrices; 64 states, or 128 with "fuse128"
static bool fused_counts(const psmc_hip_ctx *c) { return c->fuse && c->expect_impl == 1 && (c->ns == 64 || (c->ns == 128 && c->fuse128)); }

// the column-per-lane transfer-matrix kernel (k_kcol2_struct): 64 states.  With 65..128 states it takes 2.3x fewer vector
// instructions too, but its chain path ends later and the E-step is slower -- round 2: 30.6 vs 28.3 ms factored; round 3, with
// the run tiles in the second launch of the counts and the faster chain kernel: 45.7 vs 44.1 ms full counts, 29.2 vs 27.3
// factored, whatever kc_sub (profiles/r03_kc_min_sweep.txt) -- so that instantiation is not built.
static bool kcol2_on(const psmc_hip_ctx *c) { return c->ns == 64; }

static void fill_common(psmc_hip_ctx *c, EstepLaunch &p, hipStream_t st, const double *par_base = nullptr)
{
	memset(&p, 0, sizeof(p));
	p.stream = st;
	p.rep_impl = c->rep_impl; p.expect_impl = c->expect_impl; p.n_states = c->n;
	const double *pb = par_base ? par_base : c->d_par; // parameter block (batch: the first of the group's blocks)
	p.d_a = pb; p.d_aeT = pb + 4096; p.d_e = pb + 4 * 4096; p.d_a0 = pb + 4 * 4096 + 192;
	p.d_re = pb + 4 * 4096 + 192 + 64;
	p.d_sp = pb + psmc_hip_ctx::SP_OFF; p.structured = c->use_struct ? 1 : 0;
	p.fused = (c->use_struct && fused_counts(c)) ? 1 : 0;
	if (c->want_factored) p.fused = 2;
	p.ckpt = (c->want_factored && c->ckpt && c->use_struct && c->ns == 64 && c->chunk_used % 8 == 0) ? 1 : 0;
	c->last_fused = p.fused; c->last_ckpt = p.ckpt;
	p.d_kcc = pb + psmc_hip_ctx::KCC_OFF;
	p.ns = c->ns;
	if (c->ns == 128) {
		p.d_aeT = pb + 16384; p.d_e = pb + 32768; p.d_a0 = pb + 32768 + 384;
		p.d_re = pb + psmc_hip_ctx::RE128_OFF; p.d_sp = pb + psmc_hip_ctx::SP128_OFF; p.d_kcc = pb + psmc_hip_ctx::KCC128_OFF;
	}
	p.d_obs = c->d_obs; p.d_seg_off = c->d_seg_off; p.d_seg_len = c->d_seg_len;
	p.d_work = c->d_work; p.n_work = (int)c->work.size();
	p.d_f = c->d_f; p.d_b = c->d_b; p.d_s = c->d_s; p.d_sb = c->d_sb;
	for (int i = 0; i < 10; ++i) p.ev[i] = c->mode == PSMC_HIP_MODE_FAST || i < 5 ? c->ev[i] : nullptr;
}

static void collect_timing(psmc_hip_ctx *c)
{
	float t;
	c->timing_valid = true;
	auto el = [&](hipEvent_t a, hipEvent_t b, double &out) {
		// an event this path never recorded makes the call fail: that is expected here, and must not stay behind as the
		// thread's "last error" for the next launch to trip over
		if (hipEventElapsedTime(&t, a, b) == hipSuccess) out = t; else { out = 0; c->timing_valid = false; (void)hipGetLastError(); }
	};
	el(c->ev[0], c->ev[4], c->last_ms[0]);
	c->last_ms[5] = c->last_ms[6] = 0;
	if (c->mode == PSMC_HIP_MODE_FAST) {
		el(c->ev[0], c->ev[1], c->last_ms[1]);  // both sweep chains (speculate + repairs), run concurrently
		el(c->ev[1], c->ev[3], c->last_ms[2]);  // LL + what is left of the counts after the chains
		if (c->timing_two_launches) { // lists A and B of the fused back half: the sum of the two launches, not the wait between them
			double a = 0, b = 0;
			el(c->ev[8], c->evx[11], a); el(c->evx[12], c->ev[9], b);
			c->last_ms[3] = a + b;
		} else el(c->ev[8], c->ev[9], c->last_ms[3]);  // the full expect pass (kernel alone)
		el(c->ev[3], c->ev[4], c->last_ms[4]);
		if (getenv("PSMC_HIP_DEBUG_TIMES")) { // where the step's time goes, from the marks the launchers leave anyway
			double cs = 0, ce = 0, wk = 0, rt = 0;
			auto el2 = [&](hipEvent_t a, hipEvent_t b, double &out) { if (hipEventElapsedTime(&t, a, b) == hipSuccess) out = t; else (void)hipGetLastError(); };
			el2(c->ev[0], c->ev[8], cs); el2(c->ev[0], c->ev[9], ce); el2(c->ev[0], c->evx[6], wk); el2(c->ev[0], c->evx[7], rt);
			double kc = 0, bg = 0; el2(c->ev[0], c->evx[8], kc); el2(c->ev[0], c->evx[1], bg);
			fprintf(stderr, "[psmc_hip] times: total %.3f | bulk grid end %.3f | matrices end %.3f walks+chain end %.3f run tiles end %.3f | counts %.3f .. %.3f\n", c->last_ms[0], bg, kc, wk, rt, cs, ce);
		}
		el(c->ev[0], c->ev[5], c->last_ms[5]);  // speculative forward sweep kernel
		el(c->ev[7], c->ev[6], c->last_ms[6]);  // speculative backward sweep kernel
	} else {
		for (int i = 0; i < 4; ++i) el(c->ev[i], c->ev[i + 1], c->last_ms[i + 1]);
	}
}

// hmm_lk, khmm.c:245-260, on the host with the platform libm (same log() the
// reference binary would call on this machine).
static double host_lk(const double *s, int L)
{
	double sum = 0.0, prod = 1.0;
	for (int u = 0; u < L; ++u) {
		prod *= s[u];
		if (prod < HMM_TINY_H || prod >= 1.0 / HMM_TINY_H) { sum += log(prod); prod = 1.0; }
	}
	sum += log(prod);
	return sum;
}

// ---------------------------------------------------------------- exact mode
// per-entry outputs of the exact kernels: he->A, he->E, he->A0 and the underflow check value, one slot per work item
static int ensure_seg_outputs(psmc_hip_ctx *c, int nw)
{
	if (c->seg_cap >= nw) return 0;
	const size_t S = (size_t)c->ns;
	int rc;
	if ((rc = dev_alloc(c, &c->d_segA, (size_t)nw * S * S))) return rc;
	if ((rc = dev_alloc(c, &c->d_segE, (size_t)nw * 3 * S))) return rc;
	if ((rc = dev_alloc(c, &c->d_segA0, (size_t)nw * S))) return rc;
	if ((rc = dev_alloc(c, &c->d_chk, (size_t)nw))) return rc;
	c->seg_cap = nw;
	return 0;
}

static int run_exact(psmc_hip_ctx *c, const double *a, const double *e, const double *a0)
{
	HIPCHK(c, hipSetDevice(c->device));
	if (c->n_seg < 1) return fail(c, PSMC_HIP_ESTATE, "estep: no segments loaded");
	int rc;
	if ((rc = ensure_tables(c, true))) return rc;
	const int nw = (int)c->work.size();
	const size_t S = (size_t)c->ns;
	if ((rc = ensure_seg_outputs(c, nw))) return rc;
	if ((rc = stage_params(c, a, e, a0, c->stream))) return rc;
	c->tables_batch = false;
	EstepLaunch p;
	fill_common(c, p, c->stream);
	p.d_segA = c->d_segA; p.d_segE = c->d_segE; p.d_segA0 = c->d_segA0; p.d_chk = c->d_chk;
	if (launch_exact(p) != 0) return fail(c, PSMC_HIP_EDEVICE, "launch_exact", hipGetLastError());
	c->h_segA.resize((size_t)nw * S * S); c->h_segE.resize((size_t)nw * 3 * S); c->h_segA0.resize((size_t)nw * S);
	c->h_chk.resize(nw); c->h_s.resize((size_t)c->total);
	HIPCHK(c, hipMemcpyAsync(c->h_segA.data(), c->d_segA, sizeof(double) * nw * S * S, hipMemcpyDeviceToHost, c->stream));
	HIPCHK(c, hipMemcpyAsync(c->h_segE.data(), c->d_segE, sizeof(double) * nw * 3 * S, hipMemcpyDeviceToHost, c->stream));
	HIPCHK(c, hipMemcpyAsync(c->h_segA0.data(), c->d_segA0, sizeof(double) * nw * S, hipMemcpyDeviceToHost, c->stream));
	HIPCHK(c, hipMemcpyAsync(c->h_chk.data(), c->d_chk, sizeof(double) * nw, hipMemcpyDeviceToHost, c->stream));
	HIPCHK(c, hipMemcpyAsync(c->h_s.data(), c->d_s, sizeof(double) * c->total, hipMemcpyDeviceToHost, c->stream));
	HIPCHK(c, hipStreamSynchronize(c->stream));
	collect_timing(c);
	return 0;
}

extern "C" int psmc_hip_estep_segments(psmc_hip_ctx *c, const double *a, const double *e, const double *a0,
                                       double *segA, double *segE, double *segA0, double *segLL, double *chk)
{
	if (!c || !a || !e || !a0) return fail(c, PSMC_HIP_EINVAL, "estep_segments: bad argument");
	if (c->mode != PSMC_HIP_MODE_EXACT) return fail(c, PSMC_HIP_ENOTSUP, "estep_segments: exact mode only");
	int rc = run_exact(c, a, e, a0);
	if (rc) return rc;
	const int n = c->n, ns = (int)c->sel.size();
	const size_t S = (size_t)c->ns;
	for (int i = 0; i < ns; ++i) {
		const int w = c->sel2work[i], seg = c->sel[i];
		if (segA)
			for (int k = 0; k < n; ++k)
				memcpy(segA + ((size_t)i * n + k) * n, &c->h_segA[(size_t)w * S * S + k * S], sizeof(double) * n);
		if (segE)
			for (int b = 0; b < 3; ++b)
				memcpy(segE + ((size_t)i * 3 + b) * n, &c->h_segE[(size_t)w * 3 * S + b * S], sizeof(double) * n);
		if (segA0) memcpy(segA0 + (size_t)i * n, &c->h_segA0[(size_t)w * S], sizeof(double) * n);
		if (segLL) segLL[i] = host_lk(&c->h_s[(size_t)c->off[seg]], c->L[seg]);
		if (chk) chk[i] = c->h_chk[w];
	}
	return PSMC_HIP_OK;
}

static int estep_exact(psmc_hip_ctx *c, const double *a, const double *e, const double *a0, double *A, double *E,
                       double *A0, double *LL, double *chk)
{
	int rc = run_exact(c, a, e, a0);
	if (rc) return rc;
	const int n = c->n, ns = (int)c->sel.size(), nw = (int)c->work.size();
	const size_t S = (size_t)c->ns;
	std::vector<double> lk(nw);
	for (int w = 0; w < nw; ++w) lk[w] = host_lk(&c->h_s[(size_t)c->off[c->work[w]]], c->L[c->work[w]]);
	// hmm_add_expect in input order (khmm.c:346-359), he_sum starting from calloc'ed zeros
	std::vector<double> sA((size_t)n * n, 0.0), sE((size_t)2 * n, 0.0), sA0(n, 0.0);
	double ll = 0.0;
	for (int i = 0; i < ns; ++i) {
		const int w = c->sel2work[i];
		const double *hA = &c->h_segA[(size_t)w * S * S], *hE = &c->h_segE[(size_t)w * 3 * S], *hA0 = &c->h_segA0[(size_t)w * S];
		ll += lk[w]; // em.c:48
		for (int k = 0; k < n; ++k) {
			sA0[k] += hA0[k];
			for (int l = 0; l < n; ++l) sA[(size_t)k * n + l] += hA[k * S + l];
		}
		for (int b = 0; b < 2; ++b)
			for (int l = 0; l < n; ++l) sE[(size_t)b * n + l] += hE[b * S + l];
		if (chk) chk[i] = c->h_chk[w];
	}
	if (A) memcpy(A, sA.data(), sizeof(double) * n * n);
	if (E) memcpy(E, sE.data(), sizeof(double) * 2 * n);
	if (A0) memcpy(A0, sA0.data(), sizeof(double) * n);
	if (LL) *LL = ll;
	return PSMC_HIP_OK;
}

// ---------------------------------------------------------------- fast mode
// Buffers that do not depend on the tiling (reduction staging, result vector, read-back words).  Kept apart from
// plan_fast so that an entry point can make sure its result buffer exists BEFORE the first E-step without planning:
// the plan depends on whether the matrix has the PSMC form, which only stage_params() finds out.
static int ensure_fast_buffers(psmc_hip_ctx *c)
{
	if (c->d_stage && c->d_stats && c->d_warm && c->d_cnt && c->h_cnt && c->d_gate) return 0;
	int rc;
	const size_t sl = (size_t)c->ns * c->ns + 3 * (size_t)c->ns + 1;
	if ((rc = dev_alloc(c, &c->d_stage, (size_t)RED_ROWS * sl))) return rc;
	if ((rc = dev_alloc(c, &c->d_stats, sl))) return rc;
	if ((rc = dev_alloc(c, &c->d_warm, (size_t)2))) return rc;
	if ((rc = dev_alloc(c, &c->d_cnt, (size_t)4))) return rc;
	if ((rc = dev_alloc(c, &c->d_gate, (size_t)4))) return rc;
	if (!c->h_cnt && (hipHostMalloc((void **)&c->h_cnt, 4 * sizeof(int), hipHostMallocMapped) != hipSuccess ||
	                  hipHostGetDevicePointer((void **)&c->m_cnt, c->h_cnt, 0) != hipSuccess)) {
		c->h_cnt = nullptr;
		return fail(c, PSMC_HIP_ENOMEM, "hipHostMalloc (mapped)");
	}
	return 0;
}

static int plan_fast(psmc_hip_ctx *c)
{
	int64_t bins = 0;
	for (int32_t s : c->work) bins += c->L[s];
	int T = c->chunk;
	const bool st = c->use_struct;
	// One ROUND of the fused back half = 1024 SIMDs x one wave x four tiles = 4096 tiles.  The default plan of a genome-sized
	// input is two rounds (8192 tiles, two launches, the second list starting from the exit vectors of the first:
	// two_phase = 2).  A shard-sized input -- one rank's share of the genome at 2/4/8 GPUs, a single chromosome -- is planned
	// as ONE round instead ("shard-sized inputs"): with T the tile of the two-round plan, phase 1 costs
	// (T + W) steps at three waves per SIMD there and (2T + W) steps at two waves per SIMD here, the counts the same 2T
	// steps either way -- one round wins while T < W, i.e. below 8192 * warmup bins (25 M).  All tiles speculate in both
	// directions (no second list to wait for), phase 1 is ONE grid (merge1) so that its waves land on distinct SIMDs, and a
	// tile that fails is glued at once instead of getting a doubled warm-up first (a 6144-step item would be the critical
	// path of a phase that is otherwise (T + W) steps long).
	// (Tried in round 3 and removed: two / four waves per group of four tiles in the fused back half, each owning a half / a
	// quarter of the 64 x 64 partial, so that a round is 2048 / 1024 tiles of twice / four times the length and phase 1 has
	// half / a quarter of the warm-ups.  Parity-green and slower at every shard size -- 3.75 M bins: 4.5 / 5.7 ms against
	// 3.5; 7.5 M: 7.6 / 9.9 against 5.2; profiles/r03_count_waves_sweep.txt -- a step of the back half got 1.0x / 1.9x
	// faster where the instruction counts promised 1.6x / 2.6x, and phase 1 is bound by the runs' transfer matrices and
	// walks, not by the bulk warm-ups alone.)
	const int64_t ROUND1 = 4096;
	bool one_round = false;
	if (T <= 0) { // auto: about target_waves (dense: 1 tile per wave) or struct_tiles (4 per wave) tiles, never below 256 bins
		int64_t want = st ? c->struct_tiles : c->target_waves;
		if (st && !c->struct_tiles_set && bins < 2 * ROUND1 * (int64_t)std::max(c->warmup, 1)) {
			want = std::max<int64_t>(ROUND1 - (int64_t)c->work.size(), ROUND1 / 2); // every segment ends in a ragged tile: stay inside the round
		}
		T = (int)((bins + want - 1) / want);
		T = std::max(256, (T + 63) & ~63);
	}
	c->chunks.clear();
	for (size_t w = 0; w < c->work.size(); ++w) {
		const int32_t s = c->work[w];
		for (int32_t lo = 1; lo <= c->L[s]; lo += T) {
			Chunk ch;
			ch.off = c->off[s]; ch.L = c->L[s]; ch.lo = lo; ch.hi = std::min(c->L[s], lo + T - 1); ch.mult = c->mult[w];
			ch.flags = 0; ch.wf = ch.wb = c->warmup;
			if (ch.lo - c->warmup <= 1) ch.flags |= CHUNK_ANCHOR_F;
			if ((int64_t)ch.hi + c->warmup + 1 >= ch.L) ch.flags |= CHUNK_ANCHOR_B;
			if (ch.hi == ch.L) ch.flags |= CHUNK_LAST;
			c->chunks.push_back(ch);
		}
	}
	const int nc = (int)c->chunks.size();
	c->chunk_used = T;
	c->planned_struct = st;
	one_round = st && nc <= ROUND1; // the criterion: the tiles fit one round of the fused back half (also when the caller chose the tile length)
	c->two_phase_used = c->two_phase >= 0 ? c->two_phase : (one_round ? 0 : 2);
	c->merge1_used = c->merge1 >= 0 ? c->merge1 : (one_round ? 1 : 0);
	c->warm_shift_used = c->warm_shift_set ? c->warm_shift : (one_round ? 0 : 1);
	// Coarse items (round 4).  The fused back half wants ~4096 tiles (four on every SIMD), but phase 1 does not: with one
	// speculation per TILE a 3.75 M-bin share pays 4096 x 2 x 3072 warm-up bins for 3.75 M owned ones, at two waves per SIMD.
	// With one speculation per ITEM of two tiles the bulk grid is 512 + 512 waves -- one per SIMD, the unloaded step latency --
	// and the backward pass walks W + T steps per item, leaving the start vector of both tiles.
	// Measured (profiles/r04_coarse_sweep.txt): 3.75 M bins 2.98 -> 2.69 ms, 7.5 M 4.86 -> 4.45; no gain once a tile is as long as its
	// warm-up (15 M: 7.0 vs 7.3) or when the fine tiles already fit one wave per SIMD (500 k), none with two rounds of tiles (genome).
	c->coarse_used = c->coarse >= 0 ? std::max(c->coarse, 1) : (one_round && nc > 2048 && T < c->warmup ? 2 : 1);
	// transfer matrices: a tile's steps are cut into ranges of about 1000 steps (one wave pair each), so that the column
	// kernel is no longer than a bulk sweep; short tiles need fewer ranges -- and every range is one more 64 x 64 product
	// in the sequential chain that follows
	// (ranges of about an eighth of a bulk item, T + W steps: 4-5 at the genome plan's 3712-bin tiles, 2 at 960, 1 at 256)
	c->kc_sub_used = c->kc_sub_set ? c->kc_sub : std::max(1, std::min(4, (int)((8 * (int64_t)T + T + c->warmup - 1) / std::max(T + c->warmup, 1))));
	// the counts kernel splits a tile over n_sub waves: keep about the same number of partial blocks
	c->n_sub_used = st ? (fused_counts(c) ? 1 : std::max(1, std::min(c->n_sub, (9216 + nc - 1) / std::max(nc, 1)))) : c->n_sub;
	c->glue_f.assign(nc, 0); c->glue_b.assign(nc, 0); // a new tiling forgets what was learned
	c->items_dirty = true;
	int rc;
	if (nc > c->chunk_cap) {
		if ((rc = dev_alloc(c, &c->d_chunks, (size_t)nc))) return rc;
		if ((rc = dev_alloc(c, &c->d_entry, (size_t)nc * c->ns))) return rc;
		if ((rc = dev_alloc(c, &c->d_bexit, (size_t)(nc + 1) * c->ns))) return rc;
		if ((rc = dev_alloc(c, &c->d_bentry, (size_t)nc * c->ns))) return rc;
		if ((rc = dev_alloc(c, &c->d_dirty, (size_t)2 * nc))) return rc;
		if ((rc = dev_alloc(c, &c->d_touch, (size_t)2 * nc))) return rc;
		if ((rc = dev_alloc(c, &c->d_LLpart, (size_t)nc))) return rc;
		if ((rc = dev_alloc(c, &c->d_items, (size_t)26 * nc + 64))) return rc;
		if ((rc = dev_alloc(c, &c->d_ftiles, (size_t)2 * (nc + 16)))) return rc;
		if (c->h_ritems) { (void)hipHostFree(c->h_ritems); c->h_ritems = nullptr; }
		if (hipHostMalloc((void **)&c->h_ritems, (size_t)4 * nc * sizeof(int), hipHostMallocMapped) != hipSuccess ||
		    hipHostGetDevicePointer((void **)&c->m_ritems, c->h_ritems, 0) != hipSuccess)
			return fail(c, PSMC_HIP_ENOMEM, "hipHostMalloc (mapped)");
		c->chunk_cap = nc;
	}
	if ((rc = dev_alloc(c, &c->d_Cpart, (size_t)nc * c->n_sub_used * c->ns * c->ns))) return rc;
	if ((rc = dev_alloc(c, &c->d_Epart, (size_t)nc * c->n_sub_used * 3 * c->ns))) return rc;
	if ((rc = ensure_fast_buffers(c))) return rc;
	HIPCHK(c, hipMemcpy(c->d_chunks, c->chunks.data(), sizeof(Chunk) * nc, hipMemcpyHostToDevice));
	c->plan_dirty = false; c->chunks_dirty = false;
	return 0;
}

// Sweep items of the structured kernels: maximal runs of glued tiles (one segment, at most group_cap
// bins), ordered by step count so that the four rows of a wave finish together (longest first).
static int build_items(psmc_hip_ctx *c, bool two_phase_bwd, int coarse)
{
	const int nc = (int)c->chunks.size(), W = c->warmup;
	// Coarse items (round 4): the tiles between two learned runs are cut at fixed boundaries (tile index inside the segment
	// % coarse == 0) into BULK items of up to `coarse` tiles.  An item speculates once per direction; the forward sweep runs
	// through its tiles (X stored, every tile's entry vector left on the way), the backward pass of the fused / factored
	// plans walks it from the top tile's warm-up down to the lowest tile's top and leaves every tile's start vector.  Only
	// item heads can fail a verify; a head that does is glued to its neighbour like any tile (learn_groups) and the run it
	// forms is walked / chained as before.  Backward: not in the two-phase plan, whose odd tiles do not speculate at all.
	const int cf = std::max(1, coarse), cb = two_phase_bwd ? 1 : cf;
	// Two-phase plan (fused back half, two launches): a single tile with an odd index inside its segment does not
	// speculate backward.  It is in the second list and starts from the exit vector its neighbour above left in the
	// first launch -- half of the backward warm-up work disappears.  Verify / repair / learning are unchanged: such a
	// tile trivially agrees with its neighbour unless a later repair changes that neighbour.  (A forward counterpart --
	// odd tiles from the X_{lo-1} of their neighbour in a second forward launch -- was built in round 2, measured equal
	// and removed in round 3: the dependency costs what the saved warm-ups gain.)
	std::vector<int> odd(nc, 0), idx(nc, 0); // idx: the tile's index inside its segment
	for (int b = 1; b < nc; ++b) if (c->chunks[b].off == c->chunks[b - 1].off) { odd[b] = !odd[b - 1]; idx[b] = idx[b - 1] + 1; }
	// key: glued runs first (launched apart from the bulk), then phase A longest first, then phase B
	std::vector<std::pair<long long, std::pair<int, int>>> kf, kb; // (key, (first, count))
	auto key = [](int steps, bool run, bool phase_b) { return (run ? -(1ll << 40) : (phase_b ? (1ll << 40) : 0ll)) - steps; };
	std::vector<char> from_above(nc, 0), in_run(nc, 0), in_run_f(nc, 0), in_run_b(nc, 0); // in_run: member of a glued run of either direction
	std::vector<std::pair<int, int>> gf, gb; // forward / backward groups (first, count): learned runs and bulk items
	auto same_seg = [&](int x, int y) { return c->chunks[x].off == c->chunks[y].off; };
	for (int b = 0; b < nc;) { // forward: head b, members b+1.. while glued
		int e = b + 1;
		while (e < nc && c->glue_f[e] && same_seg(e, b) && c->chunks[e].hi - c->chunks[b].lo + 1 <= c->group_cap) ++e;
		if (e - b > 1) for (int t = b; t < e; ++t) in_run[t] = in_run_f[t] = 1;
		else // a bulk item: the unglued tiles that follow, up to the next coarse boundary or the head of a run
			while (e < nc && e - b < cf && same_seg(e, b) && idx[e] % cf != 0 && !c->glue_f[e] && !(e + 1 < nc && c->glue_f[e + 1] && same_seg(e + 1, e))) ++e;
		gf.push_back({b, e - b});
		b = e;
	}
	for (int b = 0; b < nc;) { // backward: tiles b..e-1, top tile e-1; glue_b[t] ties t to t+1
		int e = b + 1;
		while (e < nc && c->glue_b[e - 1] && same_seg(e, b) && c->chunks[e].hi - c->chunks[b].lo + 1 <= c->group_cap &&
		       c->chunks[e].lo < c->chunks[e].L) // a last tile holding only position L owns no transition: never a group's top
			++e;
		if (e - b > 1) for (int t = b; t < e; ++t) in_run[t] = in_run_b[t] = 1;
		else
			while (e < nc && e - b < cb && same_seg(e, b) && idx[e] % cb != 0 && !c->glue_b[e - 1] && !(e + 1 < nc && c->glue_b[e] && same_seg(e + 1, e)) &&
			       c->chunks[e].lo < c->chunks[e].L)
				++e;
		gb.push_back({b, e - b});
		b = e;
	}
	// Run tiles in the SECOND launch (round 3).  The first launch of the fused back half then waits for the bulk sweeps only
	// and the runs' path (walk -> transfer-matrix chain -> run tiles, the longest dependency chain of phase 1) has until the
	// end of that launch to finish.  A tile above a from-above tile must be in the first list, so a tile under a run
	// member speculates like an even one; and the second list must still fit one round of waves, so from-above tiles
	// make room for the run members (they speculate again: one more warm-up each in the backward pass of phase 1).
	int n_run = 0;
	for (int b = 0; b < nc; ++b) n_run += in_run[b];
	const int cap_b = (nc + 7) / 8 * 4; // half of the tiles, in whole groups of four
	c->runs_in_b = two_phase_bwd && c->runs_late && n_run > 0 && n_run <= cap_b / 2;
	int above_budget = c->runs_in_b ? cap_b - n_run : nc;
	for (const auto &gr : gb) {
		const int b = gr.first, e = b + gr.second;
		const Chunk &lo = c->chunks[b], &top = c->chunks[e - 1];
		// from above: the tile over it must exist in the segment and own a transition (it leaves an exit vector)
		bool pb = two_phase_bwd && e - b == 1 && odd[b] && b + 1 < nc && c->chunks[b + 1].off == lo.off && c->chunks[b + 1].lo < c->chunks[b + 1].L;
		if (pb && c->runs_in_b && (in_run[b] || in_run[b + 1] || above_budget <= 0)) pb = false;
		if (pb) { from_above[b] = 1; --above_budget; }
		kb.push_back({key(std::min(top.hi + chunk_warm_b(top, W) + 1, top.L) - lo.lo, in_run_b[b] != 0, pb), {b, e - b}});
	}
	// tile lists of the fused back half: A = every tile whose X and start vector exist after phase A, B = the rest.
	// B must hold the from-above tiles; A must hold the tile above every from-above tile; the run tiles go to B (above);
	// the rest can go to either and balance the two launches (each should fit the device in one round of waves)
	std::vector<int> la, lb;
	{
		std::vector<int> freet;
		for (int b = 0; b < nc; ++b) {
			if (from_above[b]) lb.push_back(b | (1 << 30));
			else if (b > 0 && from_above[b - 1]) la.push_back(b);
			else if (c->runs_in_b && in_run[b]) lb.push_back(b);
			else freet.push_back(b);
		}
		const bool single = lb.empty() && nc <= 4096; // one round of waves holds every tile: one launch, nothing to balance
		for (int b : freet) { if (single || la.size() <= lb.size()) la.push_back(b); else lb.push_back(b); }
	}
	// (Tried in round 3 and removed: the forward sweep of list B's tiles as a SECOND launch beside the first launch of the back
	// half, which needs the X of list A only -- the counts of A would overlap the warm-ups and table stores of B.  Slower
	// whatever the wave priorities of the backward warm-up pass, the walks and the transfer-matrix kernel, 12.9-15.1 ms
	// against 12.7: the half-size forward launch is bound by its dependent chain and by the backward warm-up pass beside it
	// (3.9 + 5.0 ms), and the counts run 4.1 instead of 3.1 ms beside the second one.  profiles/r03_split_fwd_prio_sweep.txt)
	for (const auto &gr : gf) {
		const int b = gr.first, e = b + gr.second;
		const Chunk &h = c->chunks[b], &l = c->chunks[e - 1];
		kf.push_back({key(l.hi - std::max(1, h.lo - chunk_warm_f(h, W)) + 1, in_run_f[b] != 0, false), {b, e - b}});
	}
	std::sort(kf.begin(), kf.end()); std::sort(kb.begin(), kb.end());
	// layout of d_items (ints): items_f | items_b | ritems_f | ritems_b | members_f | members_b, 2*nc each
	std::vector<int> h((size_t)4 * nc, 0), mem((size_t)4 * nc, 0);
	c->n_mem_f = c->n_mem_b = 0;
	for (size_t i = 0; i < kf.size(); ++i) {
		h[2 * i] = kf[i].second.first; h[2 * i + 1] = kf[i].second.second;
		if (in_run_f[kf[i].second.first])
			for (int t = 0; t < kf[i].second.second; ++t) { mem[2 * (size_t)c->n_mem_f] = kf[i].second.first + t; mem[2 * (size_t)c->n_mem_f + 1] = 1; ++c->n_mem_f; }
	}
	for (size_t i = 0; i < kb.size(); ++i) {
		h[(size_t)2 * nc + 2 * i] = kb[i].second.first; h[(size_t)2 * nc + 2 * i + 1] = kb[i].second.second;
		if (in_run_b[kb[i].second.first])
			for (int t = 0; t < kb[i].second.second; ++t) {
				mem[(size_t)2 * nc + 2 * (size_t)c->n_mem_b] = kb[i].second.first + t; mem[(size_t)2 * nc + 2 * (size_t)c->n_mem_b + 1] = 1; ++c->n_mem_b;
			}
	}
	c->n_items_f = (int)kf.size(); c->n_items_b = (int)kb.size();
	c->n_long_f = c->n_long_b = 0; // glued runs sort first (more steps than any single tile)
	while (c->n_long_f < c->n_items_f && in_run_f[kf[c->n_long_f].second.first]) ++c->n_long_f;
	while (c->n_long_b < c->n_items_b && in_run_b[kb[c->n_long_b].second.first]) ++c->n_long_b;
	c->n_B_b = 0; // from-above singles sort last
	for (int b = 0; b < nc; ++b) c->n_B_b += from_above[b];
	{
		c->n_list_a = (int)la.size(); c->n_list_b = (int)lb.size();
		c->count_group = c->ns == 128 && c->fuse128 == 2 ? 16 : 4;
		while (la.size() % c->count_group) la.push_back(-1);
		while (lb.size() % c->count_group) lb.push_back(-1);
		la.insert(la.end(), lb.begin(), lb.end());
		if (!la.empty()) HIPCHK(c, hipMemcpy(c->d_ftiles, la.data(), sizeof(int) * la.size(), hipMemcpyHostToDevice));
	}
	c->items_two_phase = two_phase_bwd ? 2 : 0; c->items_coarse = coarse;
	HIPCHK(c, hipMemcpy(c->d_items, h.data(), sizeof(int) * h.size(), hipMemcpyHostToDevice));
	{ // every tile outside the backward runs as a one-tile item, bulk items in launch order: what the factored back half's main pass
	  // takes when its bulk items span several tiles (its kernels work on single tiles)
		std::vector<int> sg;
		for (size_t i = (size_t)c->n_long_b; i < kb.size(); ++i)
			for (int t = kb[i].second.second - 1; t >= 0; --t) { sg.push_back(kb[i].second.first + t); sg.push_back(1); }
		c->n_singles_b = (int)sg.size() / 2;
		if (!sg.empty()) HIPCHK(c, hipMemcpy(c->d_items + (size_t)24 * nc, sg.data(), sizeof(int) * sg.size(), hipMemcpyHostToDevice));
	}
	HIPCHK(c, hipMemcpy(c->d_items + (size_t)8 * nc, mem.data(), sizeof(int) * mem.size(), hipMemcpyHostToDevice));
	// Walk lists and transfer-matrix chains.  A run of >= kc_min tiles is a "chain run": a walk delivers the start vector
	// of its head tile (the usual speculative warm-up, nothing more: walk item with count <= 0, see k_walk1_struct; round 1
	// and most of round 2 also walked THROUGH the head tile, 3712 dependent steps whose result nobody read -- measured
	// neutral all the same, 12.95 vs 13.07 ms: the longest walk is a head with a doubled warm-up, 6144 steps), and the
	// boundary vectors of its other tiles come from the transfer matrices, the head's first.
	//   d_items + 12nc: wl_f (2nc) | wl_b (2nc) | kc tiles (4nc: KcTile) | runs (4nc + : KcRun)
	std::vector<int> wl((size_t)4 * nc, 0), kc, runs_f, runs_b;
	c->n_wl_f = c->n_wl_b = 0;
	// auto: measured per model size and plan (profiles/r03_kc_min_sweep.txt): one more tile is walked where there are two rounds of tiles
	const int kc_min = c->kc_min >= 0 ? c->kc_min : (c->ns == 128 ? (nc > 4096 ? 12 : 8) : (nc > 4096 ? 5 : 4));
	const bool chains = kc_min >= 2; // 64 states: one state per lane in the chain kernel; 128: two
	const int head_count = c->ns == 64 ? 0 : 1; // k_walk1_struct knows count 0 (stop at the head's start vector); the four-runs-per-wave walk of 65..128 states walks through the head tile
	auto add_runs = [&](const std::vector<std::pair<long long, std::pair<int, int>>> &k, int n_long, bool bwd) {
		int &nw = bwd ? c->n_wl_b : c->n_wl_f;
		std::vector<int> &rv = bwd ? runs_b : runs_f;
		int *w = wl.data() + (bwd ? (size_t)2 * nc : 0);
		// a transfer matrix costs 16 tile sweeps: keep them for the longest runs (the list is sorted longest first)
		// and let the rest walk -- at most 1/16 of the tiles, i.e. about one more bulk sweep of work
		int budget = std::max(64, nc / c->kc_div);
		for (int i = 0; i < n_long; ++i) {
			int first = k[i].second.first, count = k[i].second.second;
			const bool chain = chains && count >= kc_min && count - 1 <= budget;
			if (chain) budget -= count - 1;
			if (chain && !bwd && c->chunks[first].lo == 1) {
				// position 1 is an initial condition, not a step: there is no X_0 for a transfer matrix to start from.
				// Walk through the first tile as well and chain from the second one.
				w[2 * nw] = first; w[2 * nw + 1] = (head_count == 0 && count - 1 >= 2) ? -1 : 2; ++nw; // -1: through the first tile, up to the head's start vector
				first += 1; count -= 1;
				if (count >= 2) {
					rv.push_back(first); rv.push_back(count); rv.push_back((int)(kc.size() / 2)); rv.push_back(0);
					for (int t = first; t < first + count - 1; ++t) { kc.push_back(t); kc.push_back(0); }
				}
			} else if (chain) {
				w[2 * nw] = bwd ? first + count - 1 : first; w[2 * nw + 1] = head_count; ++nw; // the head tile's start vector
				rv.push_back(first); rv.push_back(count); rv.push_back((int)(kc.size() / 2)); rv.push_back(0);
				if (!bwd) for (int t = first; t < first + count - 1; ++t) { kc.push_back(t); kc.push_back(0); }
				else for (int t = first + count - 1; t > first; --t) { kc.push_back(t); kc.push_back(1); }
			} else { w[2 * nw] = first; w[2 * nw + 1] = (head_count == 0 && !bwd) ? -(count - 1) : count; ++nw; } // k_walk1_struct forward: stop where the last tile starts
		}
	};
	add_runs(kf, c->n_long_f, false);
	add_runs(kb, c->n_long_b, true);
	c->n_chain_f = (int)runs_f.size() / 4; c->n_chain_b = (int)runs_b.size() / 4; c->n_kc = (int)kc.size() / 2;
	if ((size_t)c->n_kc > (size_t)2 * nc || (size_t)(c->n_chain_f + c->n_chain_b) > (size_t)nc) return fail(c, PSMC_HIP_ESTATE, "build_items: list overflow");
	HIPCHK(c, hipMemcpy(c->d_items + (size_t)12 * nc, wl.data(), sizeof(int) * wl.size(), hipMemcpyHostToDevice));
	if (c->n_kc > 0) {
		std::vector<int> runs(runs_f); runs.insert(runs.end(), runs_b.begin(), runs_b.end());
		HIPCHK(c, hipMemcpy(c->d_items + (size_t)16 * nc, kc.data(), sizeof(int) * kc.size(), hipMemcpyHostToDevice));
		HIPCHK(c, hipMemcpy(c->d_items + (size_t)20 * nc, runs.data(), sizeof(int) * runs.size(), hipMemcpyHostToDevice));
		const size_t nsub = kcol2_on(c) ? (size_t)c->kc_sub_used : 1; // k_kcol2_struct: kc_sub matrices per tile
		const size_t need = (size_t)c->n_kc * nsub * ((size_t)c->ns * c->ns + c->ns); // the matrices, then one exponent per column
		if (need > c->kcol_cap) { int rc; if ((rc = dev_alloc(c, &c->d_Kcol, need))) return rc; c->kcol_cap = need; }
	}
	c->items_dirty = false;
	if (getenv("PSMC_HIP_DEBUG"))
		fprintf(stderr, "[psmc_hip] buffers: obs %p (+%lld) chunks %p (%d) items %p entry %p bentry %p bexit %p f %p b %p s %p sb %p Cpart %p Epart %p ftiles %p touch %p par %p Kcol %p\n",
		        (void *)c->d_obs, (long long)c->total + 256, (void *)c->d_chunks, nc, (void *)c->d_items, (void *)c->d_entry, (void *)c->d_bentry, (void *)c->d_bexit,
		        (void *)c->d_f, (void *)c->d_b, (void *)c->d_s, (void *)c->d_sb, (void *)c->d_Cpart, (void *)c->d_Epart, (void *)c->d_ftiles, (void *)c->d_touch, (void *)c->d_par, (void *)c->d_Kcol);
	if (getenv("PSMC_HIP_DEBUG"))
		fprintf(stderr, "[psmc_hip] items: %d tiles of %d bins, fwd %d items (%d runs, %d run tiles, %d walks, %d chains), bwd %d items (%d runs, %d run tiles, %d walks, %d chains, %d from above), %d transfer matrices, fused lists %d + %d\n",
		        nc, c->chunk_used, c->n_items_f, c->n_long_f, c->n_mem_f, c->n_wl_f, c->n_chain_f, c->n_items_b, c->n_long_b, c->n_mem_b, c->n_wl_b, c->n_chain_b, c->n_B_b, c->n_kc, c->n_list_a, c->n_list_b);
	return 0;
}

// Tiles a repair round had to touch start where the chain forgets slowly: first a longer warm-up of their own
// ("warm_shift": up to warmup << shift bins; a tile's warm-up lengths are its own, Chunk::wf / wb), then glue each to the
// neighbour it depends on, so that from the next E-step on one row walks the region (or a chain of transfer matrices
// crosses it) while the sweep is still running.  Measured on the benchmark genome (profiles/r02_warm_shift_ab.json):
// 3072 + one doubling beats the 4096 of round 1 by 3-4 % (full counts) and 12 % (factored statistics); longer second
// warm-ups (16 K, 32 K) cost more than the runs they avoid, because a 20 k-step item is as long as the whole phase.
//
// Tried in round 3 and removed: warm-ups that FOLLOW the measured mismatch of every tile's speculation (k_verify left it
// in host-mapped memory; log10(err) ~ -w / lambda_t gives the length that lands two decades below the tolerance, a
// measurement at the rounding floor shrinks the warm-up by an eighth per E-step).  It works as intended -- the mean
// warm-up of the benchmark genome falls from 3271 to 2200 bins in 17 E-steps, parity-green -- and buys nothing: the
// steady-state E-step stays at 12.9-13.0 ms (3.75 M-bin share: 2.97 vs 3.0), because phase 1 ends when the runs' path
// (walk -> transfer-matrix chain -> run tiles) does, not when the bulk warm-ups do; and while it adapts, 5-30 of 8127
// tiles per E-step overshoot and fail, and ONE repair round costs 8 ms at this size (20-24 ms per E-step for the first
// 17): profiles/r03_adaptive_warmup_trace.txt.
static void learn_groups(psmc_hip_ctx *c)
{
	const int nc = (int)c->chunks.size(), W = c->warmup;
	const int cap = W << c->warm_shift_used;
	for (int b : c->flagged_f)
		if (b > 0 && b < nc && !c->glue_f[b] && c->chunks[b - 1].off == c->chunks[b].off) {
			Chunk &ch = c->chunks[b];
			if (ch.wf < cap && ch.lo - ch.wf > 1) { ch.wf = std::min(cap, std::max(2 * ch.wf, W)); c->chunks_dirty = true; } // not yet at the cap, and there is sequence left to warm up on
			else c->glue_f[b] = 1;
			c->items_dirty = true;
		}
	for (int b : c->flagged_b)
		if (b >= 0 && b + 1 < nc && !c->glue_b[b] && c->chunks[b + 1].off == c->chunks[b].off) {
			Chunk &ch = c->chunks[b];
			if (ch.wb < cap && (int64_t)ch.hi + ch.wb + 1 < ch.L) { ch.wb = std::min(cap, std::max(2 * ch.wb, W)); c->chunks_dirty = true; }
			else c->glue_b[b] = 1;
			c->items_dirty = true;
		}
}

static int enqueue_fast(psmc_hip_ctx *c, const double *a, const double *e, const double *a0, double *d_out,
                        hipStream_t st)
{
	HIPCHK(c, hipSetDevice(c->device));
	if (c->n_seg < 1) return fail(c, PSMC_HIP_ESTATE, "estep: no segments loaded");
	int rc;
	if ((rc = ensure_tables(c, false))) return rc;
	c->tables_batch = false;
	if ((rc = stage_params(c, a, e, a0, st))) return rc; // also decides whether the structured sweeps apply
	// the backward table: only for the unfused back half (the fused and the factored one never store bt)
	if (!c->want_factored && !(c->use_struct && fused_counts(c)) && (rc = ensure_tables(c, true))) return rc;
	if (c->ns == 128 && !c->use_struct)
		return fail(c, PSMC_HIP_ENOTSUP, "fast mode beyond 64 states needs a transition matrix of the PSMC form (structured sweeps)");
	if (c->want_factored && !c->use_struct)
		return fail(c, PSMC_HIP_ENOTSUP, "factored statistics need a transition matrix of the PSMC form");
	if ((c->plan_dirty || c->planned_struct != c->use_struct) && (rc = plan_fast(c))) return rc;
	EstepLaunch p;
	fill_common(c, p, st);
	p.d_chunks = c->d_chunks; p.n_chunks = (int)c->chunks.size(); p.warmup = c->warmup; p.n_sub = c->n_sub_used;
	// the fused back half takes both directions of the two-phase plan, the factored one (item lists, two waves per SIMD) the forward one
	// 1: both directions; 2: backward only (the fused back half runs as two launches anyway, so its second list can start
	// from the exit vectors of the first at no cost in scheduling, and half of the backward warm-up pass disappears)
	const bool two_phase_bwd = c->two_phase_used >= 1 && p.fused == 1; // the fused back half only: the factored one is a single pass over item lists
	p.merge1 = c->merge1_used; p.merge_order = c->merge_order >= 0 ? c->merge_order : (c->chunk_used < c->warmup ? 1 : 0);
	if (c->chunks_dirty) { // learned warm-ups (learn_groups) reach the device before the next launch reads them
		HIPCHK(c, hipStreamSynchronize(st));
		HIPCHK(c, hipMemcpy(c->d_chunks, c->chunks.data(), sizeof(Chunk) * c->chunks.size(), hipMemcpyHostToDevice));
		c->chunks_dirty = false;
	}
	// coarse bulk items: the fused and the factored back half only (their backward pass of phase 1 leaves start vectors, no table)
	const int coarse = (c->use_struct && p.fused != 0) ? c->coarse_used : 1;
	if (c->use_struct && (c->items_dirty || c->items_two_phase != (two_phase_bwd ? 2 : 0) || c->items_coarse != coarse) && (rc = build_items(c, two_phase_bwd, coarse))) return rc;
	p.lanes8 = (c->lanes8 >= 0 ? c->lanes8 != 0 && p.fused != 0 : p.fused == 2) && c->ns == 64 ? 1 : 0;
	p.d_gate = (c->gate >= 0 ? c->gate != 0 : coarse > 1) ? c->d_gate : nullptr;
	p.coarse = coarse; p.d_singles_b = c->d_items + 24 * (size_t)p.n_chunks; p.n_singles_b = c->n_singles_b;
	p.n_B_b = c->n_B_b; p.runs_in_b = c->runs_in_b ? 1 : 0; p.n_list_a = c->n_list_a; p.n_list_b = c->n_list_b; p.d_ftiles = c->d_ftiles; p.count_group = c->count_group;
	c->timing_two_launches = p.fused == 1 && p.n_list_b > 0;
	p.d_items_f = c->d_items; p.d_items_b = c->d_items + 2 * p.n_chunks;
	p.d_ritems_f = c->d_items + 4 * p.n_chunks; p.d_ritems_b = c->d_items + 6 * p.n_chunks;
	p.n_items_f = c->n_items_f; p.n_items_b = c->n_items_b; p.tile_len = c->chunk_used; p.h_ritems = c->h_ritems; p.m_ritems = c->m_ritems; p.m_cnt = c->m_cnt;
	c->flagged_f.clear(); c->flagged_b.clear();
	p.flagged_f = &c->flagged_f; p.flagged_b = &c->flagged_b;
	p.d_entry = c->d_entry; p.d_bentry = c->d_bentry; p.d_bexit = c->d_bexit; p.d_Cpart = c->d_Cpart; p.d_Epart = c->d_Epart;
	p.d_dirty = c->d_dirty; p.d_cnt = c->d_cnt; p.h_cnt = c->h_cnt; p.tol = c->warm_tol; p.max_rounds = c->max_rounds;
	p.d_touch_f = c->d_touch; p.d_touch_b = c->d_touch + p.n_chunks; p.d_dirty_b = c->d_dirty + p.n_chunks;
	p.stream2 = c->stream2; p.stream3 = c->stream3; p.stream4 = c->stream4; p.overlap = c->overlap;
	p.n_long_f = c->n_long_f; p.n_long_b = c->n_long_b; p.n_mem_f = c->n_mem_f; p.n_mem_b = c->n_mem_b;
	p.d_members_f = c->d_items + 8 * p.n_chunks; p.d_members_b = c->d_items + 10 * p.n_chunks;
	p.d_wl_f = c->d_items + 12 * (size_t)p.n_chunks; p.d_wl_b = c->d_items + 14 * (size_t)p.n_chunks; p.n_wl_f = c->n_wl_f; p.n_wl_b = c->n_wl_b;
	p.d_kc = c->d_items + 16 * (size_t)p.n_chunks; p.d_kruns = c->d_items + 20 * (size_t)p.n_chunks;
	p.n_kc = c->n_kc; p.n_chain_f = c->n_chain_f; p.n_chain_b = c->n_chain_b;
	p.kcol_prio = c->kcol_prio;
	p.d_Kcol = c->d_Kcol; p.kc_sub = kcol2_on(c) ? c->kc_sub_used : 1;
	p.d_Kexp = c->d_Kcol ? c->d_Kcol + (size_t)c->n_kc * p.kc_sub * c->ns * c->ns : nullptr; p.stream5 = c->stream5;
	for (int i = 0; i < 14; ++i) p.evx[i] = c->evx[i];
	p.d_LLpart = c->d_LLpart; p.d_stage = c->d_stage; p.d_stats = d_out; p.d_warm = c->d_warm;
	p.tiny_total = (double)c->sel.size() * HMM_TINY_H;
	if (launch_fast(p, &c->report) != 0) return fail(c, PSMC_HIP_EDEVICE, "launch_fast", hipGetLastError());
	if (!c->report.converged) return fail(c, PSMC_HIP_ECONVERGE, "fast mode: tile boundaries did not converge within max_rounds");
	if (c->use_struct && c->learn) learn_groups(c);
	return 0;
}

static int read_warm(psmc_hip_ctx *c, hipStream_t st)
{
	unsigned long long w[2];
	HIPCHK(c, hipMemcpyAsync(w, c->d_warm, sizeof(w), hipMemcpyDeviceToHost, st));
	HIPCHK(c, hipStreamSynchronize(st));
	memcpy(&c->warm_err[0], &w[0], 8); memcpy(&c->warm_err[1], &w[1], 8);
	return 0;
}

extern "C" int psmc_hip_estep_device(psmc_hip_ctx *c, const double *a, const double *e, const double *a0,
                                     void *d_stats, void *stream)
{
	if (!c || !a || !e || !a0 || !d_stats) return fail(c, PSMC_HIP_EINVAL, "estep_device: bad argument");
	if (c->mode != PSMC_HIP_MODE_FAST) return fail(c, PSMC_HIP_ENOTSUP, "estep_device: fast mode only");
	c->timing_valid = false;
	return enqueue_fast(c, a, e, a0, (double *)d_stats, (hipStream_t)stream);
}

extern "C" int psmc_hip_fast_diag(psmc_hip_ctx *c, double *wf, double *wb, int *n_chunks, int *warmup_used)
{
	if (!c) return PSMC_HIP_EINVAL;
	if (c->mode != PSMC_HIP_MODE_FAST || !c->d_warm) return fail(c, PSMC_HIP_ESTATE, "fast_diag: no fast E-step yet");
	HIPCHK(c, hipSetDevice(c->device));
	unsigned long long w[2];
	HIPCHK(c, hipMemcpy(w, c->d_warm, sizeof(w), hipMemcpyDeviceToHost));
	memcpy(&c->warm_err[0], &w[0], 8); memcpy(&c->warm_err[1], &w[1], 8);
	if (wf) *wf = c->warm_err[0];
	if (wb) *wb = c->warm_err[1];
	if (n_chunks) *n_chunks = (int)c->chunks.size();
	if (warmup_used) *warmup_used = c->warmup;
	return PSMC_HIP_OK;
}

extern "C" int psmc_hip_fast_repairs(psmc_hip_ctx *c, int out[4])
{
	if (!c || !out) return PSMC_HIP_EINVAL;
	out[0] = c->report.fwd_rounds; out[1] = c->report.bwd_rounds;
	out[2] = c->report.fwd_tiles; out[3] = c->report.bwd_tiles;
	return PSMC_HIP_OK;
}

extern "C" int psmc_hip_estep_factored(psmc_hip_ctx *c, const double *a, const double *e, const double *a0, double *sums,
                                       double *E, double *LL)
{
	if (!c || !a || !e || !a0) return fail(c, PSMC_HIP_EINVAL, "estep_factored: bad argument");
	if (c->mode != PSMC_HIP_MODE_FAST) return fail(c, PSMC_HIP_ENOTSUP, "estep_factored: fast mode only");
	HIPCHK(c, hipSetDevice(c->device));
	int rc;
	if ((rc = ensure_fast_buffers(c))) return rc; // d_stats must exist before the first enqueue (the plan follows stage_params)
	c->want_factored = true;
	rc = enqueue_fast(c, a, e, a0, c->d_stats, c->stream);
	c->want_factored = false;
	if (rc) return rc;
	if ((rc = read_warm(c, c->stream))) return rc;
	collect_timing(c);
	const int n = c->n;
	std::vector<double> h((size_t)7 * n + 1);
	HIPCHK(c, hipMemcpy(h.data(), c->d_stats, sizeof(double) * h.size(), hipMemcpyDeviceToHost));
	if (sums) memcpy(sums, h.data(), sizeof(double) * 5 * n);
	if (E) memcpy(E, h.data() + (size_t)5 * n, sizeof(double) * 2 * n);
	if (LL) *LL = h[(size_t)7 * n];
	return PSMC_HIP_OK;
}

// the factored statistics left in HBM: [SL | SU | DG | CL | CU | E(2n) | LL], 7n + 1 doubles (what the sharded
// E-step of group.hip reduces over the devices)
extern "C" int psmc_hip_estep_factored_device(psmc_hip_ctx *c, const double *a, const double *e, const double *a0, void *d_stats, void *stream)
{
	if (!c || !a || !e || !a0 || !d_stats) return fail(c, PSMC_HIP_EINVAL, "estep_factored_device: bad argument");
	if (c->mode != PSMC_HIP_MODE_FAST) return fail(c, PSMC_HIP_ENOTSUP, "estep_factored_device: fast mode only");
	HIPCHK(c, hipSetDevice(c->device));
	c->timing_valid = false;
	c->want_factored = true;
	const int rc = enqueue_fast(c, a, e, a0, (double *)d_stats, (hipStream_t)stream);
	c->want_factored = false;
	return rc;
}

extern "C" int psmc_hip_fast_info(psmc_hip_ctx *c, int out[8])
{
	if (!c || !out) return PSMC_HIP_EINVAL;
	out[0] = c->use_struct ? 1 : 0; out[1] = c->chunk_used; out[2] = c->use_struct ? c->n_items_f : (int)c->chunks.size();
	out[3] = c->use_struct ? c->n_items_b : (int)c->chunks.size();
	out[4] = c->last_fused; out[5] = c->last_ckpt; out[6] = c->timing_two_launches ? 2 : 1; out[7] = c->merge1_used;
	return PSMC_HIP_OK;
}

extern "C" int psmc_hip_fast_plan(psmc_hip_ctx *c, double out[8])
{
	if (!c || !out) return PSMC_HIP_EINVAL;
	// the plan is made by the first fast E-step after load / select / an option that changes it (it depends on the matrix's form)
	if (c->plan_dirty || c->chunks.empty()) return fail(c, PSMC_HIP_ESTATE, "fast_plan: no current plan (run a fast E-step first)");
	const int nc = (int)c->chunks.size();
	double sf = 0, sb = 0, mf = 0, mb = 0; int nf = 0, nb = 0, gf = 0, gb = 0;
	for (int b = 0; b < nc; ++b) {
		const Chunk &ch = c->chunks[b];
		if (c->glue_f[b]) ++gf; else if (!(ch.flags & CHUNK_ANCHOR_F)) { sf += ch.wf; mf = std::max(mf, (double)ch.wf); ++nf; }
		if (c->glue_b[b]) ++gb; else if (!(ch.flags & (CHUNK_ANCHOR_B | CHUNK_LAST))) { sb += ch.wb; mb = std::max(mb, (double)ch.wb); ++nb; }
	}
	out[0] = nc; out[1] = c->chunk_used; out[2] = nf ? sf / nf : 0.0; out[3] = nb ? sb / nb : 0.0; out[4] = mf; out[5] = mb; out[6] = gf; out[7] = gb;
	return PSMC_HIP_OK;
}

static int estep_fast(psmc_hip_ctx *c, const double *a, const double *e, const double *a0, double *A, double *E,
                      double *A0, double *LL, double *chk)
{
	const int n = c->n;
	int rc = enqueue_fast(c, a, e, a0, c->d_stats, c->stream);
	if (rc) return rc;
	if ((rc = read_warm(c, c->stream))) return rc;
	collect_timing(c);
	std::vector<double> h((size_t)n * n + 2 * n + 1);
	HIPCHK(c, hipMemcpy(h.data(), c->d_stats, sizeof(double) * h.size(), hipMemcpyDeviceToHost));
	if (A) memcpy(A, h.data(), sizeof(double) * n * n);
	if (E) memcpy(E, h.data() + (size_t)n * n, sizeof(double) * 2 * n);
	if (LL) *LL = h[(size_t)n * n + 2 * n];
	if (A0) memset(A0, 0, sizeof(double) * n); // unused downstream (khmm.c:321-322); not computed in fast mode
	if (chk) for (size_t i = 0; i < c->sel.size(); ++i) chk[i] = 1.0;
	return PSMC_HIP_OK;
}

extern "C" int psmc_hip_estep(psmc_hip_ctx *c, const double *a, const double *e, const double *a0, double *A, double *E,
                              double *A0, double *LL, double *chk)
{
	if (!c || !a || !e || !a0) return fail(c, PSMC_HIP_EINVAL, "estep: bad argument");
	if (c->mode == PSMC_HIP_MODE_EXACT) return estep_exact(c, a, e, a0, A, E, A0, LL, chk);
	HIPCHK(c, hipSetDevice(c->device));
	int rc = ensure_fast_buffers(c); // d_stats must exist before the first enqueue (the plan follows stage_params)
	if (rc) return rc;
	rc = estep_fast(c, a, e, a0, A, E, A0, LL, chk);
	// 65..128 states and a matrix without the two rank-1 triangles (psmc_cap_matrix, a foreign HMM): the tiled dense
	// sweeps keep one lane per state, so the dense fallback there is the exact path (two states per lane, the matrix in
	// LDS, one wave per segment) -- slower, any matrix, and trivially inside the fast-mode tolerance
	if (rc == PSMC_HIP_ENOTSUP && c->ns == 128 && !c->use_struct) return estep_exact(c, a, e, a0, A, E, A0, LL, chk);
	return rc;
}

// ---------------------------------------------------------------- batch (bootstrap replicates)
namespace {
struct RepSel { std::vector<int32_t> work, sel2work; int64_t bins = 0; }; // unique segments in order of first appearance
}

static int batch_selections(psmc_hip_ctx *c, int n_rep, const int32_t *sel_off, const int32_t *sel_idx, std::vector<RepSel> &reps)
{
	std::vector<int32_t> pos(c->n_seg);
	reps.assign(n_rep, RepSel());
	for (int r = 0; r < n_rep; ++r) {
		const int n_sel = sel_off[r + 1] - sel_off[r];
		if (n_sel < 1) return fail(c, PSMC_HIP_EINVAL, "estep_batch: empty selection");
		std::fill(pos.begin(), pos.end(), -1);
		RepSel &R = reps[r];
		R.sel2work.resize(n_sel);
		for (int i = 0; i < n_sel; ++i) {
			const int32_t sg = sel_idx[sel_off[r] + i];
			if (sg < 0 || sg >= c->n_seg) return fail(c, PSMC_HIP_EINVAL, "estep_batch: index out of range");
			if (pos[sg] < 0) { pos[sg] = (int32_t)R.work.size(); R.work.push_back(sg); R.bins += ((int64_t)c->L[sg] + 63) & ~(int64_t)63; }
			R.sel2work[i] = pos[sg];
		}
	}
	return 0;
}

// Table bins an exact batch may use: "batch_bins", or 0.9 of the device memory that is free or already in this context's tables.
// (Round 3 took 0.9 of the free memory PLUS all of what the context held: the second call saw a larger capacity than the first,
// re-planned its groups and re-allocated 250 GB of tables -- 8 s; profiles/r04_boot_breakdown.txt.)
static bool batch_refwd(const psmc_hip_ctx *c) { return c->ns == 64 && c->exact_refwd != 0; } // k_expect_exact_rf: 64 states
static int batch_capacity(psmc_hip_ctx *c, int64_t *cap)
{
	*cap = c->batch_bins;
	if (*cap > 0) return 0;
	size_t fr = 0, tot = 0;
	HIPCHK(c, hipMemGetInfo(&fr, &tot));
	const double S = (double)c->ns, per_bin = S * 8.0 * (batch_refwd(c) ? 1.0 : 2.0) + 8.0;
	const double held = (double)c->tab_bins * (S * 8.0 * ((c->have_b ? 1.0 : 0.0) + (c->d_f ? 1.0 : 0.0)) + 8.0 + (c->d_sb ? 8.0 : 0.0));
	*cap = (int64_t)(((double)fr + held) * 0.9 / per_bin) - 256;
	if (*cap < 1) return fail(c, PSMC_HIP_ENOMEM, "estep_batch: no device memory left for tables");
	return 0;
}

extern "C" int psmc_hip_reserve_batch_tables(psmc_hip_ctx *c, int64_t max_bins)
{
	if (!c) return PSMC_HIP_EINVAL;
	if (c->mode != PSMC_HIP_MODE_EXACT) return PSMC_HIP_OK; // fast mode keeps one replicate's tables: nothing to reserve
	HIPCHK(c, hipSetDevice(c->device));
	if (c->n_seg < 1) return fail(c, PSMC_HIP_ESTATE, "reserve_batch_tables: no segments loaded");
	int64_t cap = 0;
	int rc = batch_capacity(c, &cap);
	if (rc) return rc;
	return ensure_tables(c, true, max_bins > 0 ? std::min(cap, max_bins) : cap, !batch_refwd(c));
}

// Exact mode: the sweeps of ALL replicates of a group in one launch each (forward, backward, expect), replicate-major;
// every (replicate, unique segment) entry has its own table slot and reads its replicate's parameter block.  Groups =
// as many consecutive replicates as fit the table memory.  Statistics are added per replicate in selection order on
// the host, exactly like psmc_hip_estep: bit-identical to n_rep separate calls.
static int batch_exact(psmc_hip_ctx *c, int n_rep, const double *a, const double *e, const double *a0, const int32_t *sel_off,
                       const int32_t *sel_idx, double *A, double *sums, double *E, double *LL)
{
	const int n = c->n;
	const size_t S = (size_t)c->ns, PL = psmc_hip_ctx::PAR_LEN;
	std::vector<RepSel> reps;
	int rc;
	if ((rc = batch_selections(c, n_rep, sel_off, sel_idx, reps))) return rc;
	// how many table bins fit (batch_capacity: the same answer in every call, whatever the context holds already)
	int64_t cap = 0;
	if ((rc = batch_capacity(c, &cap))) return rc;
	size_t n_entries_all = 0;
	for (const RepSel &R : reps) n_entries_all += R.work.size();
	const int align = c->ns == 128 ? (n_entries_all <= 256 ? 1 : (n_entries_all <= 512 ? 2 : 4)) : 4; // sweeps per block sharing one parameter set in LDS
	c->last_batch_groups = 0;
	std::vector<int32_t> wseg, wpar; std::vector<int64_t> wtab; std::vector<int> first;
	std::vector<double> hp, lk;
	{ // the groups are known before the first launch: size the tables once for the largest of them (no re-allocation in the loop)
		int64_t worst = 0; size_t worst_entries = 0; int worst_reps = 0;
		for (int r0 = 0; r0 < n_rep;) {
			if (reps[r0].bins > cap) return fail(c, PSMC_HIP_ENOMEM, "estep_batch: the tables of one replicate do not fit the device memory (batch_bins)");
			int r1 = r0; int64_t bins = 0; size_t ent = 0;
			while (r1 < n_rep && bins + reps[r1].bins <= cap) { bins += reps[r1].bins; ent += (reps[r1].work.size() + align - 1) / align * align; ++r1; }
			worst = std::max(worst, bins); worst_entries = std::max(worst_entries, ent); worst_reps = std::max(worst_reps, r1 - r0);
			r0 = r1;
		}
		// tables: for the largest group when the caller fixed "batch_bins"; else for everything that fits (or all replicates at once), ONCE -- a hipMalloc of
		// 250 GB takes 4-6 s on this driver (it clears the memory: scripts/r04/malloc_probe.py), so the size must not depend on
		// this call's groups, and psmc_hip_reserve_batch_tables lets a caller pay for it while it loads
		int64_t all_bins = 0; // what ONE group of all replicates would need: never allocate beyond it
		for (const RepSel &R : reps) all_bins += R.bins;
		if ((rc = ensure_tables(c, true, c->batch_bins > 0 ? worst : std::max(worst, std::min(cap, all_bins)), !batch_refwd(c)))) return rc;
		if ((rc = ensure_seg_outputs(c, (int)worst_entries))) return rc;
		if (c->bw_cap < worst_entries) {
			if ((rc = dev_alloc(c, &c->d_bw_seg, worst_entries))) return rc;
			if ((rc = dev_alloc(c, &c->d_bw_par, worst_entries))) return rc;
			if ((rc = dev_alloc(c, &c->d_bw_tab, worst_entries))) return rc;
			c->bw_cap = worst_entries;
		}
		if (c->bpar_cap < (size_t)worst_reps) { if ((rc = dev_alloc(c, &c->d_bpar, (size_t)worst_reps * PL))) return rc; c->bpar_cap = (size_t)worst_reps; }
	}
	static const bool dbg_t = getenv("PSMC_HIP_DEBUG_TIMES") != nullptr;
	auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
	for (int r0 = 0; r0 < n_rep;) {
		const double t_a = now();
		int r1 = r0; int64_t bins = 0;
		while (r1 < n_rep && bins + reps[r1].bins <= cap) { bins += reps[r1].bins; ++r1; }
		const int ng = r1 - r0;
		wseg.clear(); wpar.clear(); wtab.clear(); first.assign(ng, 0);
		int64_t run = 0;
		for (int r = r0; r < r1; ++r) {
			first[r - r0] = (int)wseg.size();
			for (int32_t sg : reps[r].work) { wseg.push_back(sg); wpar.push_back(r - r0); wtab.push_back(run); run += ((int64_t)c->L[sg] + 63) & ~(int64_t)63; }
			while (wseg.size() % align) { wseg.push_back(-1); wpar.push_back(r - r0); wtab.push_back(0); }
		}
		const int nw = (int)wseg.size();
		hp.resize((size_t)ng * PL);
		for (int r = r0; r < r1; ++r) (void)fill_params(c, a + (size_t)r * n * n, e + (size_t)r * 2 * n, a0 + (size_t)r * n, hp.data() + (size_t)(r - r0) * PL);
		HIPCHK(c, hipStreamSynchronize(c->stream));
		HIPCHK(c, hipMemcpy(c->d_bw_seg, wseg.data(), sizeof(int32_t) * nw, hipMemcpyHostToDevice));
		HIPCHK(c, hipMemcpy(c->d_bw_par, wpar.data(), sizeof(int32_t) * nw, hipMemcpyHostToDevice));
		HIPCHK(c, hipMemcpy(c->d_bw_tab, wtab.data(), sizeof(int64_t) * nw, hipMemcpyHostToDevice));
		HIPCHK(c, hipMemcpy(c->d_bpar, hp.data(), sizeof(double) * hp.size(), hipMemcpyHostToDevice));
		c->tables_batch = true;
		EstepLaunch p;
		fill_common(c, p, c->stream, c->d_bpar);
		p.d_work = c->d_bw_seg; p.n_work = nw; p.d_work_par = c->d_bw_par; p.d_work_tab = c->d_bw_tab; p.par_stride = (int64_t)PL; p.work_align = align;
		p.exact_refwd = batch_refwd(c) ? 1 : 0;
		p.d_segA = c->d_segA; p.d_segE = c->d_segE; p.d_segA0 = c->d_segA0; p.d_chk = c->d_chk;
		const double t_b = now();
		if (launch_exact(p) != 0) return fail(c, PSMC_HIP_EDEVICE, "launch_exact (batch)", hipGetLastError());
		if (dbg_t) (void)hipStreamSynchronize(c->stream);
		const double t_c = now();
		c->h_segA.resize((size_t)nw * S * S); c->h_segE.resize((size_t)nw * 3 * S); c->h_s.resize((size_t)run);
		HIPCHK(c, hipMemcpyAsync(c->h_segA.data(), c->d_segA, sizeof(double) * nw * S * S, hipMemcpyDeviceToHost, c->stream));
		HIPCHK(c, hipMemcpyAsync(c->h_segE.data(), c->d_segE, sizeof(double) * nw * 3 * S, hipMemcpyDeviceToHost, c->stream));
		HIPCHK(c, hipMemcpyAsync(c->h_s.data(), c->d_s, sizeof(double) * (size_t)run, hipMemcpyDeviceToHost, c->stream));
		HIPCHK(c, hipStreamSynchronize(c->stream));
		const double t_d = now();
		collect_timing(c);
		// hmm_add_expect in selection order per replicate (khmm.c:346-359), he_sum starting from zeros; LL += hmm_lk (em.c:48)
		std::vector<double> sA((size_t)n * n), sE((size_t)2 * n);
		// hmm_lk of every (replicate, segment) entry: a running product over all of its bins with the platform log() -- 0.7 ns per bin,
		// 0.36 s per group of 28 replicates on one core; the entries are independent, so host threads share them (each value is
		// computed by one thread exactly as before: bit-identical)
		std::vector<double> lk_all(wseg.size(), 0.0);
		{
			const unsigned nt = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
			std::atomic<size_t> next(0);
			auto work = [&]() {
				for (size_t i = next.fetch_add(1); i < wseg.size(); i = next.fetch_add(1))
					if (wseg[i] >= 0) lk_all[i] = host_lk(&c->h_s[(size_t)wtab[i]], c->L[wseg[i]]);
			};
			std::vector<std::thread> th;
			for (unsigned t = 1; t < nt; ++t) th.emplace_back(work);
			work();
			for (std::thread &t : th) t.join();
		}
		for (int r = r0; r < r1; ++r) {
			const RepSel &R = reps[r];
			const int f0 = first[r - r0];
			lk.resize(R.work.size());
			for (size_t j = 0; j < R.work.size(); ++j) lk[j] = lk_all[(size_t)f0 + j];
			std::fill(sA.begin(), sA.end(), 0.0); std::fill(sE.begin(), sE.end(), 0.0);
			double ll = 0.0;
			for (size_t i = 0; i < R.sel2work.size(); ++i) {
				const int w = f0 + R.sel2work[i];
				const double *hA = &c->h_segA[(size_t)w * S * S], *hE = &c->h_segE[(size_t)w * 3 * S];
				ll += lk[R.sel2work[i]];
				for (int k = 0; k < n; ++k)
					for (int l = 0; l < n; ++l) sA[(size_t)k * n + l] += hA[k * S + l];
				for (int b = 0; b < 2; ++b)
					for (int l = 0; l < n; ++l) sE[(size_t)b * n + l] += hE[b * S + l];
			}
			if (A) memcpy(A + (size_t)r * n * n, sA.data(), sizeof(double) * n * n);
			if (E) memcpy(E + (size_t)r * 2 * n, sE.data(), sizeof(double) * 2 * n);
			if (LL) LL[r] = ll;
			if (sums) { // SL | SU | DG | CL | CU of the summed matrix, for callers with the O(N) objective
				double *q = sums + (size_t)r * 5 * n;
				memset(q, 0, sizeof(double) * 5 * n);
				for (int k = 0; k < n; ++k)
					for (int l = 0; l < n; ++l) {
						const double v = sA[(size_t)k * n + l];
						if (l < k) { q[k] += v; q[3 * n + l] += v; } else if (l > k) { q[n + k] += v; q[4 * n + l] += v; } else q[2 * n + k] = v;
					}
			}
		}
		if (dbg_t) fprintf(stderr, "[psmc_hip] batch group %d: %d replicates, %d entries, %.1f M table bins | prepare %.3f s, kernels %.3f (fwd %.0f bwd %.0f expect %.0f ms), read-back %.3f, host sums %.3f\n",
		                   c->last_batch_groups, ng, nw, run / 1e6, t_b - t_a, t_c - t_b, c->last_ms[1], c->last_ms[2], c->last_ms[3], t_d - t_c, now() - t_d);
		++c->last_batch_groups;
		r0 = r1;
	}
	return PSMC_HIP_OK;
}

// Fast mode: a single replicate already fills the device, so the replicates run one after the other -- but each
// keeps ITS OWN tile plan (tiling of its selection, learned glued runs, transfer-matrix lists) in a child context,
// so nothing is re-planned or re-learned from one EM iteration to the next.  Children share the parent's tables.
static psmc_hip_ctx *batch_child(psmc_hip_ctx *c, int r)
{
	while ((int)c->kids.size() <= r) {
		psmc_hip_ctx *k = new (std::nothrow) psmc_hip_ctx();
		if (!k) return nullptr;
		k->n = c->n; k->ns = c->ns; k->device = c->device; k->mode = c->mode; k->parent = c;
		k->chunk = c->chunk; k->warmup = c->warmup; k->max_rounds = c->max_rounds; k->rep_impl = c->rep_impl; k->expect_impl = c->expect_impl;
		k->n_sub = c->n_sub; k->target_waves = c->target_waves; k->overlap = c->overlap; k->warm_tol = c->warm_tol; k->struct_opt = c->struct_opt;
		k->struct_tiles_set = c->struct_tiles_set; k->struct_tiles = c->struct_tiles;
		k->two_phase = c->two_phase; k->kc_div = c->kc_div; k->kc_min = c->kc_min; k->ckpt = c->ckpt; k->fuse = c->fuse;
		k->learn = c->learn; k->group_cap = c->group_cap; k->warm_shift = c->warm_shift; k->kc_sub = c->kc_sub; k->kcol_prio = c->kcol_prio;
		k->fuse128 = c->fuse128; k->coarse = c->coarse; k->gate = c->gate; k->lanes8 = c->lanes8; k->exact_refwd = c->exact_refwd;
		k->merge1 = c->merge1; k->merge_order = c->merge_order; k->runs_late = c->runs_late; k->warm_shift_set = c->warm_shift_set; k->kc_sub_set = c->kc_sub_set;
		k->stream = c->stream; k->stream2 = c->stream2; k->stream3 = c->stream3; k->stream4 = c->stream4; k->stream5 = c->stream5;
		for (int i = 0; i < 14; ++i) k->evx[i] = c->evx[i];
		for (int i = 0; i < 10; ++i) k->ev[i] = c->ev[i];
		k->h_par = c->h_par; k->d_par = c->d_par;
		// same segments, same offsets, the parent's copy of the observations
		k->d_obs = c->d_obs; k->obs_borrowed = true; k->off = c->off; k->total = c->total;
		if (set_segments_common(k, c->n_seg, c->L.data()) != 0) { c->err = k->err; psmc_hip_destroy(k); return nullptr; } // never keep a half-built child
		c->kids.push_back(k);
	}
	return c->kids[r];
}

static int batch_fast(psmc_hip_ctx *c, int n_rep, const double *a, const double *e, const double *a0, const int32_t *sel_off,
                      const int32_t *sel_idx, double *A, double *sums, double *E, double *LL)
{
	const int n = c->n;
	for (int r = 0; r < n_rep; ++r) {
		psmc_hip_ctx *k = batch_child(c, r);
		if (!k) return fail(c, PSMC_HIP_ENOMEM, "estep_batch: cannot create the replicate context");
		const int n_sel = sel_off[r + 1] - sel_off[r];
		if (n_sel < 1) return fail(c, PSMC_HIP_EINVAL, "estep_batch: empty selection");
		const int32_t *idx = sel_idx + sel_off[r];
		int rc = 0;
		if ((int)k->sel.size() != n_sel || memcmp(k->sel.data(), idx, sizeof(int32_t) * n_sel) != 0) rc = psmc_hip_select(k, n_sel, idx);
		const double *ar = a + (size_t)r * n * n, *er = e + (size_t)r * 2 * n, *a0r = a0 + (size_t)r * n;
		if (rc == 0) {
			if (A) rc = psmc_hip_estep(k, ar, er, a0r, A + (size_t)r * n * n, E ? E + (size_t)r * 2 * n : nullptr, nullptr, LL ? LL + r : nullptr, nullptr);
			if (rc == 0 && sums) rc = psmc_hip_estep_factored(k, ar, er, a0r, sums + (size_t)r * 5 * n, E ? E + (size_t)r * 2 * n : nullptr, LL ? LL + r : nullptr);
		}
		if (rc) { c->err = "replicate " + std::to_string(r) + ": " + k->err; return rc; }
	}
	c->tables_batch = true;
	c->last_batch_groups = n_rep;
	return PSMC_HIP_OK;
}

extern "C" int psmc_hip_estep_batch(psmc_hip_ctx *c, int n_rep, const double *a, const double *e, const double *a0,
                                    const int32_t *sel_off, const int32_t *sel_idx, double *A, double *sums, double *E, double *LL)
{
	if (!c || n_rep < 1 || !a || !e || !a0 || !sel_off || !sel_idx || (!A && !sums)) return fail(c, PSMC_HIP_EINVAL, "estep_batch: bad argument");
	if (c->parent) return fail(c, PSMC_HIP_EINVAL, "estep_batch: not on a replicate context");
	if (c->n_seg < 1) return fail(c, PSMC_HIP_ESTATE, "estep_batch: no segments loaded");
	HIPCHK(c, hipSetDevice(c->device));
	if (c->mode == PSMC_HIP_MODE_EXACT) return batch_exact(c, n_rep, a, e, a0, sel_off, sel_idx, A, sums, E, LL);
	return batch_fast(c, n_rep, a, e, a0, sel_off, sel_idx, A, sums, E, LL);
}

extern "C" int psmc_hip_batch_info(psmc_hip_ctx *c, int out[2])
{
	if (!c || !out) return PSMC_HIP_EINVAL;
	out[0] = c->last_batch_groups; out[1] = (int)c->kids.size();
	return PSMC_HIP_OK;
}

// Diagnostic, not part of the ABI (tests and scripts only): after a fast E-step with the fused back half (64 states), check the start
// vector of every tile against a plain dense backward recursion on the host and name the tiles that are off (stderr).
extern "C" int psmcdbg_check_bentry(psmc_hip_ctx *c)
{
	if (!c || c->ns != 64 || !c->d_bentry || !c->d_ftiles) return -1;
	(void)hipSetDevice(c->device);
	(void)hipDeviceSynchronize();
	const int nc = (int)c->chunks.size();
	const int ga = (c->n_list_a + 3) / 4, gb = (c->n_list_b + 3) / 4, ng = ga + gb;
	std::vector<int> tl(4 * (size_t)ng), tf(nc), tb(nc);
	const std::vector<Chunk> &ch = c->chunks;
	(void)hipMemcpy(tl.data(), c->d_ftiles, sizeof(int) * tl.size(), hipMemcpyDeviceToHost);
	(void)hipMemcpy(tf.data(), c->d_touch, sizeof(int) * tf.size(), hipMemcpyDeviceToHost);
	(void)hipMemcpy(tb.data(), c->d_touch + nc, sizeof(int) * tb.size(), hipMemcpyDeviceToHost);
	const double *d_e = c->d_par + 4 * 4096; // 64 states: a | aeT[3] | e[3] | a0 | 1/e (fill_common)
	{ // the start vector of every tile against a plain dense backward recursion on the host
			std::vector<double> ha(64 * 64), he(3 * 64), hb((size_t)nc * 64), hx((size_t)nc * 64);
			(void)hipMemcpy(ha.data(), c->d_par, sizeof(double) * ha.size(), hipMemcpyDeviceToHost);
			(void)hipMemcpy(he.data(), d_e, sizeof(double) * he.size(), hipMemcpyDeviceToHost);
			(void)hipMemcpy(hb.data(), c->d_bentry, sizeof(double) * hb.size(), hipMemcpyDeviceToHost);
			(void)hipMemcpy(hx.data(), c->d_bexit, sizeof(double) * hx.size(), hipMemcpyDeviceToHost);
			for (int k = 0; k < 64; ++k) he[128 + k] = 1.0;
			std::vector<int> from_above(nc, 0), in_list(nc, 0);
			for (size_t i = 0; i < tl.size(); ++i) if (tl[i] >= 0) { from_above[tl[i] & ~(1 << 30)] = (tl[i] >> 30) & 1; in_list[tl[i] & ~(1 << 30)] = i < 4 * (size_t)ga ? 1 : 2; }
			for (int b0 = 0; b0 < nc;) {
				int b1 = b0; while (b1 + 1 < nc && ch[b1 + 1].off == ch[b0].off) ++b1; // tiles b0..b1 of one segment
				const int L = ch[b0].L;
				std::vector<uint8_t> o(L);
				(void)hipMemcpy(o.data(), c->d_obs + ch[b0].off, (size_t)L, hipMemcpyDeviceToHost);
				std::vector<double> bt(64), y(64);
				for (int k = 0; k < 64; ++k) bt[k] = he[(o[L - 1] & 3) * 64 + k]; // bt_L = e[o_L] . 1
				int b = b1;
				for (int pp = L - 1; pp >= 1; --pp) { // bt = bt_{pp+1}
					while (b >= b0 && std::min(ch[b].hi, L - 1) < ch[b].lo) --b; // tiles without a transition
					if (b >= b0 && pp == std::min(ch[b].hi, L - 1)) { // bentry[b] should be bt_{top+1} up to a factor
						double sx = 0, sy = 0, num = 0, den = 0;
						for (int k = 0; k < 64; ++k) { sx += hb[(size_t)b * 64 + k]; sy += bt[k]; }
						for (int k = 0; k < 64; ++k) { num = std::max(num, std::fabs(hb[(size_t)b * 64 + k] / sx - bt[k] / sy)); den = std::max(den, bt[k] / sy); }
						if (!(num / den < 1e-9))
							fprintf(stderr, "[psmc_hip] RECHECK bentry of tile %d (lo %d hi %d L %d list %d from_above %d tf %d tb %d; above: tf %d tb %d list %d) off by %.2e, sum %.3e; rounds %d/%d\n",
							        b, ch[b].lo, ch[b].hi, L, in_list[b], from_above[b], tf[b], tb[b], b + 1 <= b1 ? tf[b + 1] : -1, b + 1 <= b1 ? tb[b + 1] : -1, b + 1 <= b1 ? in_list[b + 1] : -1, num / den, sx,
							        c->report.fwd_rounds, c->report.bwd_rounds);
						--b;
					}
					double tot = 0;
					for (int k = 0; k < 64; ++k) { double acc = 0; for (int l = 0; l < 64; ++l) acc += ha[k * 64 + l] * bt[l]; y[k] = acc; }
					for (int k = 0; k < 64; ++k) { bt[k] = he[(o[pp - 1] & 3) * 64 + k] * y[k]; tot += bt[k]; }
					for (int k = 0; k < 64; ++k) bt[k] /= tot;
				}
				b0 = b1 + 1;
			}
		}
	return 0;
}

extern "C" int psmc_hip_get_tables(psmc_hip_ctx *c, int seg, double *f, double *b, double *s)
{
	if (!c || seg < 0 || seg >= c->n_seg) return fail(c, PSMC_HIP_EINVAL, "get_tables: bad argument");
	if (!c->d_f || c->tables_batch) return fail(c, PSMC_HIP_ESTATE, "get_tables: no single E-step yet");
	HIPCHK(c, hipSetDevice(c->device));
	const int n = c->n, L = c->L[seg];
	const int64_t off = c->off[seg];
	const size_t S = (size_t)c->ns;
	std::vector<double> tmp((size_t)L * S);
	for (int which = 0; which < 2; ++which) {
		double *dst = which == 0 ? f : b;
		const double *src = which == 0 ? c->d_f : c->d_b;
		if (!dst) continue;
		if (which == 1 && !c->have_b) return fail(c, PSMC_HIP_ESTATE, "get_tables: no backward table (the fused and factored back halves never store bt; set fuse=0)");
		HIPCHK(c, hipMemcpy(tmp.data(), src + off * S, sizeof(double) * (size_t)L * S, hipMemcpyDeviceToHost));
		for (int u = 0; u < L; ++u) memcpy(dst + (size_t)u * n, &tmp[(size_t)u * S], sizeof(double) * n);
	}
	if (s) HIPCHK(c, hipMemcpy(s, c->d_s + off, sizeof(double) * (size_t)L, hipMemcpyDeviceToHost));
	return PSMC_HIP_OK;
}

extern "C" int psmc_hip_decode(psmc_hip_ctx *c, int seg, int32_t *path, double *maxp)
{
	if (!c || seg < 0 || seg >= c->n_seg || !path) return fail(c, PSMC_HIP_EINVAL, "decode: bad argument");
	if (c->mode != PSMC_HIP_MODE_EXACT) return fail(c, PSMC_HIP_ENOTSUP, "decode: exact mode only");
	if (!c->d_f || !c->have_b || c->tables_batch) return fail(c, PSMC_HIP_ESTATE, "decode: no single E-step yet");
	HIPCHK(c, hipSetDevice(c->device));
	const int L = c->L[seg];
	int32_t *dp = nullptr; double *dm = nullptr;
	if (hipMalloc((void **)&dp, sizeof(int32_t) * (size_t)L) != hipSuccess) return fail(c, PSMC_HIP_ENOMEM, "hipMalloc");
	if (hipMalloc((void **)&dm, sizeof(double) * (size_t)L) != hipSuccess) { (void)hipFree(dp); return fail(c, PSMC_HIP_ENOMEM, "hipMalloc"); }
	int rc = launch_post_decode(c->stream, c->d_f, c->d_b, c->d_s, c->off[seg], L, c->n, c->ns, dp, dm);
	hipError_t e1 = hipMemcpyAsync(path, dp, sizeof(int32_t) * (size_t)L, hipMemcpyDeviceToHost, c->stream);
	hipError_t e2 = maxp ? hipMemcpyAsync(maxp, dm, sizeof(double) * (size_t)L, hipMemcpyDeviceToHost, c->stream) : hipSuccess;
	hipError_t e3 = hipStreamSynchronize(c->stream);
	(void)hipFree(dp); (void)hipFree(dm);
	if (rc || e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) return fail(c, PSMC_HIP_EDEVICE, "decode");
	return PSMC_HIP_OK;
}

extern "C" int psmc_hip_posterior(psmc_hip_ctx *c, int seg, double *post, double *recomb)
{
	if (!c || seg < 0 || seg >= c->n_seg || (!post && !recomb)) return fail(c, PSMC_HIP_EINVAL, "posterior: bad argument");
	if (c->mode != PSMC_HIP_MODE_EXACT) return fail(c, PSMC_HIP_ENOTSUP, "posterior: exact mode only");
	if (!c->d_f || !c->have_b || c->tables_batch) return fail(c, PSMC_HIP_ESTATE, "posterior: no single E-step yet");
	HIPCHK(c, hipSetDevice(c->device));
	const int L = c->L[seg], n = c->n;
	double *dp = nullptr, *dr = nullptr;
	if (post && hipMalloc((void **)&dp, sizeof(double) * (size_t)L * n) != hipSuccess) return fail(c, PSMC_HIP_ENOMEM, "hipMalloc");
	if (recomb && hipMalloc((void **)&dr, sizeof(double) * (size_t)L) != hipSuccess) { if (dp) (void)hipFree(dp); return fail(c, PSMC_HIP_ENOMEM, "hipMalloc"); }
	const double *d_e = c->ns == 128 ? c->d_par + 32768 : c->d_par + 4 * 4096;
	int rc = launch_post_full(c->stream, c->d_par, d_e, c->d_obs, c->d_f, c->d_b, c->d_s, c->off[seg], L, n, c->ns, dp, dr);
	hipError_t e1 = post ? hipMemcpyAsync(post, dp, sizeof(double) * (size_t)L * n, hipMemcpyDeviceToHost, c->stream) : hipSuccess;
	hipError_t e2 = recomb ? hipMemcpyAsync(recomb, dr, sizeof(double) * (size_t)L, hipMemcpyDeviceToHost, c->stream) : hipSuccess;
	hipError_t e3 = hipStreamSynchronize(c->stream);
	if (dp) (void)hipFree(dp);
	if (dr) (void)hipFree(dr);
	if (rc || e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) return fail(c, PSMC_HIP_EDEVICE, "posterior");
	return PSMC_HIP_OK;
}

extern "C" int psmc_hip_post_counts(psmc_hip_ctx *c, int seg, const int32_t *cnt1, int32_t l, int32_t n_cnt, double *cnt)
{
	if (!c || seg < 0 || seg >= c->n_seg || !cnt || l < 0 || n_cnt < 1 || (l > 0 && !cnt1)) return fail(c, PSMC_HIP_EINVAL, "post_counts: bad argument");
	if (c->mode != PSMC_HIP_MODE_EXACT) return fail(c, PSMC_HIP_ENOTSUP, "post_counts: exact mode only");
	if (!c->d_f || !c->have_b || c->tables_batch) return fail(c, PSMC_HIP_ESTATE, "post_counts: no single E-step yet");
	HIPCHK(c, hipSetDevice(c->device));
	const int L = c->L[seg], n = c->n, min_l = L < l ? L : l;
	if (min_l == 0) return PSMC_HIP_OK;
	int32_t *d1 = nullptr; double *dc = nullptr;
	if (hipMalloc((void **)&d1, sizeof(int32_t) * (size_t)min_l * n_cnt) != hipSuccess) return fail(c, PSMC_HIP_ENOMEM, "hipMalloc");
	if (hipMalloc((void **)&dc, sizeof(double) * (size_t)n * n_cnt) != hipSuccess) { (void)hipFree(d1); return fail(c, PSMC_HIP_ENOMEM, "hipMalloc"); }
	hipError_t e0 = hipMemcpyAsync(d1, cnt1, sizeof(int32_t) * (size_t)min_l * n_cnt, hipMemcpyHostToDevice, c->stream);
	hipError_t e1 = hipMemcpyAsync(dc, cnt, sizeof(double) * (size_t)n * n_cnt, hipMemcpyHostToDevice, c->stream);
	int rc = launch_post_counts(c->stream, c->d_f, c->d_b, c->d_s, c->off[seg], min_l, d1, n_cnt, n, c->ns, dc);
	hipError_t e2 = hipMemcpyAsync(cnt, dc, sizeof(double) * (size_t)n * n_cnt, hipMemcpyDeviceToHost, c->stream);
	hipError_t e3 = hipStreamSynchronize(c->stream);
	(void)hipFree(d1); (void)hipFree(dc);
	if (rc || e0 != hipSuccess || e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) return fail(c, PSMC_HIP_EDEVICE, "post_counts");
	return PSMC_HIP_OK;
}

extern "C" int psmc_hip_last_timing(psmc_hip_ctx *c, double ms[7])
{
	if (!c || !ms) return PSMC_HIP_EINVAL;
	if (!c->timing_valid) {
		HIPCHK(c, hipSetDevice(c->device));
		if (hipEventSynchronize(c->ev[4]) != hipSuccess) return fail(c, PSMC_HIP_ESTATE, "last_timing: nothing recorded");
		collect_timing(c);
		if (!c->timing_valid) return fail(c, PSMC_HIP_ESTATE, "last_timing: events incomplete");
	}
	for (int i = 0; i < 7; ++i) ms[i] = c->last_ms[i];
	return PSMC_HIP_OK;
}

extern "C" int psmc_hip_selftest(int device)
{
	int nd = psmc_hip_device_count();
	if (nd <= 0 || device < 0 || device >= nd) return PSMC_HIP_EDEVICE;
	if (hipSetDevice(device) != hipSuccess) return PSMC_HIP_EDEVICE;
	unsigned *d = nullptr, h = 0xffffffffu;
	if (hipMalloc((void **)&d, sizeof(unsigned)) != hipSuccess) return PSMC_HIP_ENOMEM;
	(void)hipMemset(d, 0, sizeof(unsigned));
	int rc = run_selftest(nullptr, d);
	if (rc == 0 && hipDeviceSynchronize() == hipSuccess && hipMemcpy(&h, d, sizeof(unsigned), hipMemcpyDeviceToHost) == hipSuccess)
		rc = (int)h;
	else
		rc = PSMC_HIP_EDEVICE;
	(void)hipFree(d);
	return rc;
}

extern "C" int psmc_hip_microbench(int device, double *out, int n)
{
	int nd = psmc_hip_device_count();
	if (!out || n < 1) return PSMC_HIP_EINVAL;
	if (nd <= 0 || device < 0 || device >= nd) return PSMC_HIP_EDEVICE;
	if (hipSetDevice(device) != hipSuccess) return PSMC_HIP_EDEVICE;
	double *d = nullptr, h[64];
	if (hipMalloc((void **)&d, sizeof(h)) != hipSuccess) return PSMC_HIP_ENOMEM;
	(void)hipMemset(d, 0, sizeof(h));
	int rc = run_microbench(nullptr, d); // first launch warms the clocks / instruction cache
	if (rc == 0) rc = run_microbench(nullptr, d);
	if (rc == 0 && hipDeviceSynchronize() == hipSuccess && hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess) {
		for (int i = 0; i < n && i < 64; ++i) out[i] = h[i];
		rc = PSMC_HIP_OK;
	} else rc = PSMC_HIP_EDEVICE;
	(void)hipFree(d);
	return rc;
}

extern "C" int psmc_hip_pipe_probe(int device, double *out, int n)
{
	// configurations: (waves, mask of matrix waves)
	static const struct { int waves; unsigned mask; } cfg[PSMC_HIP_PIPE_PROBE_CONFIGS] = {
		{4, 0xFu}, {4, 0x0u}, {8, 0xFFu}, {8, 0x00u}, {8, 0x0Fu}, {8, 0x55u}};
	int nd = psmc_hip_device_count();
	if (!out || n < PSMC_HIP_PIPE_PROBE_CONFIGS * 8) return PSMC_HIP_EINVAL;
	if (nd <= 0 || device < 0 || device >= nd) return PSMC_HIP_EDEVICE;
	if (hipSetDevice(device) != hipSuccess) return PSMC_HIP_EDEVICE;
	double *d = nullptr;
	if (hipMalloc((void **)&d, sizeof(double) * 8) != hipSuccess) return PSMC_HIP_ENOMEM;
	int rc = 0;
	for (int i = 0; i < PSMC_HIP_PIPE_PROBE_CONFIGS && rc == 0; ++i) {
		(void)hipMemset(d, 0, sizeof(double) * 8);
		rc = run_pipe_probe(nullptr, d, cfg[i].waves, cfg[i].mask, 8);        // warm: clocks, instruction cache
		if (rc == 0) rc = run_pipe_probe(nullptr, d, cfg[i].waves, cfg[i].mask, 64);
		if (rc == 0 && (hipDeviceSynchronize() != hipSuccess ||
		                hipMemcpy(out + 8 * i, d, sizeof(double) * 8, hipMemcpyDeviceToHost) != hipSuccess)) rc = 1;
		for (int w = cfg[i].waves; w < 8; ++w) out[8 * i + w] = 0.0;
	}
	(void)hipFree(d);
	return rc ? PSMC_HIP_EDEVICE : PSMC_HIP_OK;
}

extern "C" int psmc_hip_pipe_probe2(int device, const int *kinds8, int rounds, double *out8)
{
	int nd = psmc_hip_device_count();
	if (!kinds8 || !out8 || rounds < 1) return PSMC_HIP_EINVAL;
	for (int i = 0; i < 8; ++i) if (kinds8[i] < 0 || kinds8[i] > 9) return PSMC_HIP_EINVAL;
	if (nd <= 0 || device < 0 || device >= nd) return PSMC_HIP_EDEVICE;
	if (hipSetDevice(device) != hipSuccess) return PSMC_HIP_EDEVICE;
	double *d = nullptr; void *src = nullptr;
	if (hipMalloc((void **)&d, sizeof(double) * 8) != hipSuccess || hipMalloc(&src, 4096) != hipSuccess) { if (d) (void)hipFree(d); return PSMC_HIP_ENOMEM; }
	(void)hipMemset(d, 0, sizeof(double) * 8); (void)hipMemset(src, 1, 4096);
	int rc = run_pipe_probe2(nullptr, d, kinds8, std::max(1, rounds / 8), src); // warm: clocks, instruction cache
	if (rc == 0) rc = run_pipe_probe2(nullptr, d, kinds8, rounds, src);
	if (rc == 0 && (hipDeviceSynchronize() != hipSuccess || hipMemcpy(out8, d, sizeof(double) * 8, hipMemcpyDeviceToHost) != hipSuccess)) rc = 1;
	(void)hipFree(d); (void)hipFree(src);
	return rc ? PSMC_HIP_EDEVICE : PSMC_HIP_OK;
}

extern "C" int psmc_hip_place_probe(int device, int n_waves, int waves_per_block, int n_kernels, int steps, double *out, double *ms_out)
{
	int nd = psmc_hip_device_count();
	if (n_waves < 1 || n_waves > (1 << 16) || waves_per_block < 1 || waves_per_block > 4 || n_kernels < 1 || n_kernels > 4 || steps < 4 || !out) return PSMC_HIP_EINVAL;
	if (nd <= 0 || device < 0 || device >= nd) return PSMC_HIP_EDEVICE;
	if (hipSetDevice(device) != hipSuccess) return PSMC_HIP_EDEVICE;
	const size_t per = (size_t)3 * ((n_waves + waves_per_block - 1) / waves_per_block) * waves_per_block;
	double *d = nullptr;
	if (hipMalloc((void **)&d, sizeof(double) * per * n_kernels) != hipSuccess) return PSMC_HIP_ENOMEM;
	hipStream_t st[4] = {nullptr, nullptr, nullptr, nullptr};
	hipEvent_t e0, e1[4];
	int rc = 0;
	(void)hipEventCreate(&e0);
	for (int k = 0; k < n_kernels; ++k) { if (hipStreamCreateWithFlags(&st[k], hipStreamNonBlocking) != hipSuccess) rc = 1; (void)hipEventCreate(&e1[k]); }
	for (int pass = 0; pass < 2 && rc == 0; ++pass) { // pass 0 warms clocks and the instruction cache
		(void)hipDeviceSynchronize();
		(void)hipEventRecord(e0, st[0]);
		for (int k = 1; k < n_kernels; ++k) (void)hipStreamWaitEvent(st[k], e0, 0);
		for (int k = 0; k < n_kernels && rc == 0; ++k) { rc = run_place_probe(st[k], d + per * k, n_waves, waves_per_block, steps & ~3); (void)hipEventRecord(e1[k], st[k]); }
	}
	if (rc == 0 && hipDeviceSynchronize() == hipSuccess && hipMemcpy(out, d, sizeof(double) * per * n_kernels, hipMemcpyDeviceToHost) == hipSuccess) {
		float worst = 0;
		for (int k = 0; k < n_kernels; ++k) { float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1[k]); worst = std::max(worst, ms); }
		if (ms_out) *ms_out = worst;
	} else rc = 1;
	(void)hipEventDestroy(e0);
	for (int k = 0; k < n_kernels; ++k) { (void)hipEventDestroy(e1[k]); if (st[k]) (void)hipStreamDestroy(st[k]); }
	(void)hipFree(d);
	return rc ? PSMC_HIP_EDEVICE : PSMC_HIP_OK;
}

extern "C" int psmc_hip_stream_probe(int device, long long n_doubles, double *ms_out)
{
	int nd = psmc_hip_device_count();
	if (n_doubles < 1) return PSMC_HIP_EINVAL;
	if (nd <= 0 || device < 0 || device >= nd) return PSMC_HIP_EDEVICE;
	if (hipSetDevice(device) != hipSuccess) return PSMC_HIP_EDEVICE;
	double *a = nullptr, *b = nullptr;
	if (hipMalloc((void **)&a, sizeof(double) * n_doubles) != hipSuccess) return PSMC_HIP_ENOMEM;
	if (hipMalloc((void **)&b, sizeof(double) * n_doubles) != hipSuccess) { (void)hipFree(a); return PSMC_HIP_ENOMEM; }
	(void)hipMemset(a, 0, sizeof(double) * n_doubles);
	hipEvent_t e0, e1;
	(void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
	int rc = run_stream_probe(nullptr, a, b, (size_t)n_doubles); // warm
	(void)hipEventRecord(e0, nullptr);
	for (int i = 0; i < 4 && rc == 0; ++i) rc = run_stream_probe(nullptr, a, b, (size_t)n_doubles);
	(void)hipEventRecord(e1, nullptr);
	float ms = 0;
	if (rc == 0 && hipEventSynchronize(e1) == hipSuccess && hipEventElapsedTime(&ms, e0, e1) == hipSuccess) { if (ms_out) *ms_out = ms / 4; rc = PSMC_HIP_OK; }
	else rc = PSMC_HIP_EDEVICE;
	(void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipFree(a); (void)hipFree(b);
	return rc;
}

extern "C" int psmc_hip_hbm_probe(int device, long long bytes, double *gbps_out)
{
	int nd = psmc_hip_device_count();
	if (bytes < (1 << 24) || !gbps_out) return PSMC_HIP_EINVAL;
	if (nd <= 0 || device < 0 || device >= nd) return PSMC_HIP_EDEVICE;
	if (hipSetDevice(device) != hipSuccess) return PSMC_HIP_EDEVICE;
	bytes &= ~(long long)((1 << 23) - 1); // whole 8 MB: four 2 MB streams per wave in the sweep-store probe
	double *a = nullptr, *b = nullptr;
	if (hipMalloc((void **)&a, (size_t)bytes) != hipSuccess) return PSMC_HIP_ENOMEM;
	if (hipMalloc((void **)&b, (size_t)bytes) != hipSuccess) { (void)hipFree(a); return PSMC_HIP_ENOMEM; }
	(void)hipMemset(a, 0, (size_t)bytes); (void)hipMemset(b, 0, (size_t)bytes);
	hipEvent_t e0, e1;
	(void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
	int rc = 0;
	for (int which = 0; which < 4 && rc == 0; ++which) {
		rc = run_hbm_probe(nullptr, which, a, b, (size_t)bytes); // warm
		(void)hipEventRecord(e0, nullptr);
		for (int i = 0; i < 3 && rc == 0; ++i) rc = run_hbm_probe(nullptr, which, a, b, (size_t)bytes);
		(void)hipEventRecord(e1, nullptr);
		float ms = 0;
		if (rc == 0 && hipEventSynchronize(e1) == hipSuccess && hipEventElapsedTime(&ms, e0, e1) == hipSuccess)
			gbps_out[which] = (which == 2 ? 2.0 : 1.0) * (double)bytes / (ms / 3 * 1e-3) / 1e9;
		else rc = PSMC_HIP_EDEVICE;
	}
	(void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipFree(a); (void)hipFree(b);
	return rc ? PSMC_HIP_EDEVICE : PSMC_HIP_OK;
}

extern "C" int psmc_hip_load_probe(int device, int n_waves, int steps, double *out)
{
	int nd = psmc_hip_device_count();
	if (n_waves < 1 || n_waves > (1 << 20) || steps < 4 || !out) return PSMC_HIP_EINVAL;
	if (nd <= 0 || device < 0 || device >= nd) return PSMC_HIP_EDEVICE;
	if (hipSetDevice(device) != hipSuccess) return PSMC_HIP_EDEVICE;
	double *d = nullptr;
	if (hipMalloc((void **)&d, sizeof(double) * 2 * (size_t)n_waves) != hipSuccess) return PSMC_HIP_ENOMEM;
	std::vector<double> h((size_t)2 * n_waves);
	hipEvent_t e0, e1;
	(void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
	const int arg = steps & ~3;
	int rc = run_load_probe(nullptr, d, n_waves, arg); // warm
	(void)hipEventRecord(e0, nullptr);
	if (rc == 0) rc = run_load_probe(nullptr, d, n_waves, arg);
	(void)hipEventRecord(e1, nullptr);
	float ms = 0;
	if (rc == 0 && hipEventSynchronize(e1) == hipSuccess && hipEventElapsedTime(&ms, e0, e1) == hipSuccess &&
	    hipMemcpy(h.data(), d, sizeof(double) * h.size(), hipMemcpyDeviceToHost) == hipSuccess) {
		double cyc = 0, mhz = 0, cmax = 0;
		for (int i = 0; i < n_waves; ++i) { cyc += h[2 * (size_t)i]; mhz += h[2 * (size_t)i + 1]; cmax = std::max(cmax, h[2 * (size_t)i]); }
		out[0] = ms; out[1] = cyc / n_waves; out[2] = cmax; out[3] = mhz / n_waves;
		rc = PSMC_HIP_OK;
	} else rc = PSMC_HIP_EDEVICE;
	(void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipFree(d);
	return rc;
}

// Diagnostic: load_probe with the table stores of a sweep; modes: see include/psmc_hip.h and microbench.hip
extern "C" int psmc_hip_load_probe_st(int device, int n_waves, int steps, int store_steps, int mode, double *out)
{
	int nd = psmc_hip_device_count();
	if (n_waves < 1 || n_waves > (1 << 16) || steps < 4 || store_steps < 4 || store_steps > steps || !out) return PSMC_HIP_EINVAL;
	if (nd <= 0 || device < 0 || device >= nd) return PSMC_HIP_EDEVICE;
	if (hipSetDevice(device) != hipSuccess) return PSMC_HIP_EDEVICE;
	steps &= ~3; store_steps &= ~3;
	double *d = nullptr, *tbl = nullptr;
	const size_t tb = (size_t)n_waves * 4 * (size_t)store_steps * 512;
	if (hipMalloc((void **)&d, sizeof(double) * 2 * (size_t)n_waves) != hipSuccess) return PSMC_HIP_ENOMEM;
	if (hipMalloc((void **)&tbl, tb) != hipSuccess) { (void)hipFree(d); return PSMC_HIP_ENOMEM; }
	(void)hipMemset(tbl, 0, tb);
	std::vector<double> h((size_t)2 * n_waves);
	hipEvent_t e0, e1;
	(void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
	int rc = run_load_probe_st(nullptr, d, n_waves, steps, tbl, store_steps, mode);
	(void)hipEventRecord(e0, nullptr);
	if (rc == 0) rc = run_load_probe_st(nullptr, d, n_waves, steps, tbl, store_steps, mode);
	(void)hipEventRecord(e1, nullptr);
	float ms = 0;
	if (rc == 0 && hipEventSynchronize(e1) == hipSuccess && hipEventElapsedTime(&ms, e0, e1) == hipSuccess &&
	    hipMemcpy(h.data(), d, sizeof(double) * h.size(), hipMemcpyDeviceToHost) == hipSuccess) {
		double cyc = 0, mhz = 0, cmax = 0;
		for (int i = 0; i < n_waves; ++i) { cyc += h[2 * (size_t)i]; mhz += h[2 * (size_t)i + 1]; cmax = std::max(cmax, h[2 * (size_t)i]); }
		out[0] = ms; out[1] = cyc / n_waves; out[2] = cmax; out[3] = mhz / n_waves; out[4] = (double)tb / (ms * 1e-3) / 1e9;
		rc = PSMC_HIP_OK;
	} else rc = PSMC_HIP_EDEVICE;
	(void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipFree(d); (void)hipFree(tbl);
	return rc;
}
