// psmc_hip_internal.h -- shared between the api*.hip files (the C-ABI) and the kernel translation units.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <vector>

namespace psmc {

constexpr int NS = 64;
constexpr int NORM_EVERY = 4;   // fast mode rescales the forward vector at positions p % NORM_EVERY == 0          // padded number of states (one wave lane per state)
constexpr int STATS_LEN = NS * NS + 3 * NS + 1; // padded device stats: C/A | E[3] | LL

// One tile of one segment (fast mode).  Positions are 1-based like khmm.c.
struct Chunk {
	int64_t off;  // global bin index of the segment's first position
	int32_t L;    // segment length
	int32_t lo;   // first position owned by the tile
	int32_t hi;   // last position owned by the tile
	int32_t mult; // how many times the segment occurs in the selection (bootstrap)
	int32_t flags; // CHUNK_*
	int32_t wf, wb; // structured sweeps: the tile's own speculative warm-up in bins, forward / backward (api_fast.hip learn_groups:
	                // starts at the "warmup" option and grows after a failure)
};
__host__ __device__ inline int chunk_warm_f(const Chunk &c, int) { return c.wf; }
__host__ __device__ inline int chunk_warm_b(const Chunk &c, int) { return c.wb; }
constexpr int CHUNK_ANCHOR_F = 1; // forward speculation started at the true segment start: exact
constexpr int CHUNK_ANCHOR_B = 2; // backward speculation started at the true segment end: exact
constexpr int CHUNK_LAST = 4;     // last tile of its segment

// Exact mode: the list of sweeps of one launch, one entry per (parameter set, segment).  A single E-step has one
// parameter set and keeps every segment's tables at the segment's own offset (par == tab == nullptr); a batch
// (psmc_hip_estep_batch) gives each entry its parameter-set index and its own table slot.
struct ExWork {
	const int32_t *seg;  // [n] segment id; < 0: padding entry (keeps the entries of one parameter set block-aligned)
	const int32_t *par;  // [n] parameter-set index, or nullptr: all 0
	const int64_t *tab;  // [n] offset of the entry's f/b/s tables in bins, or nullptr: seg_off[seg]
	const int64_t *tab_s; // [n] offset of the entry's scale factors when they live elsewhere (the exact batch's forward pass over ALL replicates
	                      // at once, api_batch.hip), or nullptr: the same as its f/b tables
	int n;
	int64_t par_stride;  // doubles between consecutive parameter sets
};

struct FastReport { int fwd_rounds, bwd_rounds, fwd_tiles, bwd_tiles, converged, merged, recounted; }; // merged: tiles the forward fix pass rewrote in part (until they met their stored trajectory); recounted: 1 = a second pass of the counts ran

// Everything a launch needs (device pointers unless noted).
struct EstepLaunch {
	hipStream_t stream, stream2, stream3; // forward chain (main), backward chain, early expect
	hipEvent_t evx[14];          // cross-stream dependencies; 4/5: count read-backs of the two chains; 6/7: glued runs done
	int overlap;
	int rep_impl, n_states;
	const int64_t *d_work_tab_s; // ExWork::tab_s
	int *d_cu_mask = nullptr;    // k_expect_exact_rf2: 4096 zeroed words, or null (static wave roles)
	int exact_only;              // exact batch: 1 = only the forward pass (scale factors), 2 = everything but the forward pass, 0 = all three
	int exact_refwd;             // exact batch, 64 states: no f table -- the expect pass recomputes the forward sweep (estep_exact.hip k_expect_exact_rf)
	int ns;                      // padded number of states: 64, or 128 (exact mode only; then d_aeT is a transposed)
	// parameters (padded to NS)
	const double *d_a;   // a[l*64+k] row-major P(l->k)... i.e. a[row*64+col]
	const double *d_aeT; // aeT[b][l*64+k] = e[b][l]*a[k][l], b=0..2 (b=2 is a transposed)
	const double *d_e;   // e[b*64+k], b=0..2 (row 2 = 1)
	const double *d_a0;  // a0[k]
	const double *d_re;  // re[b*64+k] = 1/e[b][k] (0 where e is 0), b=0..2
	const double *d_sp;  // structured transition: P | R | qa | c | dd, 64 each (estep_struct.hip); valid when structured
	int structured;      // a[k][l] = P_k qa_l (l<k), R_k c_l (l>k): O(N) sweeps, 4 tiles per wave
	const double *d_kcc; // 64 states: constant tables of the column-per-lane transfer-matrix kernel k_kcol2_struct (api.hip fill_params)
	int kc_sub;          // ... the number of step ranges (transfer matrices) per tile
	int kcol_prio;       // ... and its wave priority (0..2; the bulk forward sweep runs at 1, the walks at 3)
	int fused;           // structured only: 1 = backward sweep and counts in one kernel, bt never stored (estep_fused.hip);
	                     // 2 = factored statistics, no N x N counts at all (estep_factored.hip)
	int ckpt;            // fused == 2: the forward sweep stores X at p % 8 == 0 only, the counts recompute the rest
	int merge_order;     // merged phase-1 grid: 1 = all forward blocks, then all backward blocks; 0 = alternating
	int merge1;          // fused != 0: bulk forward sweep and backward warm-up pass in ONE grid (k_sweep_struct) -- the dispatcher spreads
	                     // the waves of one grid over distinct SIMDs, not those of concurrent grids (shard-sized inputs: api_fast.hip plan_fast)
	const int *d_items_f, *d_items_b; // [n_items_*][2] sweep items (first tile, count) in launch order (estep_struct.hip)
	int n_items_f, n_items_b, tile_len;
	int n_long_f, n_long_b;           // leading items that are glued runs: walked beside the bulk (stream4 / stream3)
	const int *d_members_f, *d_members_b; // every tile of the glued runs as a one-tile item
	int n_mem_f, n_mem_b;
	int lanes8;                       // 64 states: the bulk sweeps of phase 1 run eight tiles per wave (estep_struct.hip launch_fwd_struct)
	int *d_gate;                      // [0] walk blocks started, [1] bulk blocks started: the gates that order the DISPATCH of phase 1's grids (estep_struct.hip
	                                  // k_gate); null: no gates
	int coarse;                       // > 1: a bulk item spans up to this many tiles (one speculation per item; the backward pass of the fused /
	                                  // factored plans WALKS its item and leaves every tile's start vector): api_fast.hip build_items
	const int *d_singles_b; int n_singles_b; // coarse > 1, factored back half: every tile outside the backward runs as a one-tile item
	int n_B_b;                        // trailing backward items of the two-phase plan: they start from the exit vector of the tile above (second list of the fused back half)
	const int *d_ftiles; int n_list_a, n_list_b; // fused back half: tile lists A | B (each padded to a multiple of 4 with -1)
	int count_group;                  // ... tiles per work-group of the fused back half (one C partial each): 4; 16 = k_bwd_count8x_struct (128 states, "fuse128" = 2)
	int runs_in_b;                    // ... and every tile of a glued run is in list B: only the second launch waits for the runs' path
	hipStream_t stream4, stream5;
	// walks: heads of the chain runs (count 1) followed by the short runs; transfer-matrix chains of the long runs
	const int *d_wl_f, *d_wl_b; int n_wl_f, n_wl_b;
	const int *d_kc, *d_kruns; int n_kc, n_chain_f, n_chain_b; // KcTile[n_kc], KcRun[n_chain_f + n_chain_b]
	const int *d_kslot, *d_kuniq; int n_kuniq;                 // matrix slot of every KcTile; the KcTile that computes slot u (round 5: the full tiles of a
	                                                           // run of missing data all have the SAME transfer matrix -- one slot per direction for all of them)
	double *d_Kcol, *d_Kexp;                                  // [n_kc][64][64], [n_kc][64]
	int *d_ritems_f, *d_ritems_b;     // [n_chunks][2] flagged tiles of the current repair round as one-tile items
	int *h_ritems;                    // host view of pinned, device-mapped [2][n_chunks][2]: the same lists, so that the host can learn the groups
	int *m_ritems, *m_cnt;            // device views of the mapped h_ritems / h_cnt (written by k_compact, no copy commands)
	std::vector<int> *flagged_f, *flagged_b; // out: tiles flagged in any round of this E-step (may be null)
	// data
	const uint8_t *d_obs;
	const int64_t *d_seg_off;
	const int32_t *d_seg_len;
	const int32_t *d_work; // exact: unique selected segment ids (batch: segment of every entry, -1 = padding)
	int n_work;
	const int32_t *d_work_par; const int64_t *d_work_tab; int64_t par_stride; int work_align; // exact batch, see ExWork (single E-step: null, null, 0, 0)
	double *d_f, *d_b, *d_s; // exact: f,b tables + s; fast: f = X (lag-normalised), d_b = bt, d_s = inv_d
	// exact outputs
	double *d_segA, *d_segE, *d_segA0, *d_chk;
	// fast
	const Chunk *d_chunks;
	int n_chunks, warmup, n_sub;
	double *d_entry, *d_bentry, *d_bexit; // [n_chunks][64] boundary vectors used / produced by each tile
	int *d_dirty, *d_dirty_b, *d_cnt, *h_cnt; // per-tile repair flags (fwd, bwd), flagged counts [2] (device, pinned host)
	int *d_touch_f, *d_touch_b;           // [n_chunks] each, contiguous: repaired-this-E-step flags
	// The forward fix pass (round 6; estep_struct.hip FwdCtl, launch_fast): between the forward sweep and the back half every tile's start vector is
	// checked and a tile that fails is rewritten until it meets its stored trajectory.  merge == 0: rounds 1-5 (verify afterwards, whole tiles again,
	// their groups of the counts again).
	int merge;
	const int *d_fix_f; int n_fix_f;      // one-tile items: every tile outside the forward runs whose predecessor is outside them too (checked right after the bulk sweep)
	int *d_fmerge;                        // [n_chunks] (follows d_touch_b) the 16-bin block (1-based, from the tile's first) at whose end a fix stopped, 0 = none
	double *d_finv;                       // [n_chunks] ... and the factor stored rows / rewritten rows at that point
	int *m_mlen, *h_mlen;                 // [n_chunks] mapped host memory (device / host view): 16-bin blocks a fix rewrote (> 0), -1 = the whole tile
	double *m_mis;                        // [2][n_chunks] mapped host memory: each tile's mismatch when first checked in this E-step (forward | backward):
	                                      // what the per-tile warm-ups follow (api_fast.hip adapt_warmups)
	const double *d_prevx;                // [n_chunks][ns] start vectors of the forward speculation taken from the previous E-step's X, or null (stationary vector)
	double *d_sb;                         // backward scale factors (fast mode), like d_s
	double tol; int max_rounds;
	double *d_Cpart;            // [n_chunks*n_sub][4096]
	double *d_Epart;            // [n_chunks*n_sub][192]  S partials
	double *d_LLpart;           // [n_chunks]
	double *d_stage;            // [RED_ROWS][STATS_LEN]
	double *d_stats;            // [n*n + 2n + 1] final, unpadded [A | E | LL]
	unsigned long long *d_warm; // [2] max warm-up mismatch (double bits)
	double tiny_total;          // n_selected_segments * HMM_TINY
	hipEvent_t ev[10];          // timing marks: 0 start, 1 chains done, 2 before expect(redo), 3 after, 4 end,
	                            // 5 fwd sweep end, 7/6 bwd sweep start/end, 8/9 full expect pass start/end
};

// PSMC_HIP_DEBUG_SYNC=1: synchronise the device after every launcher of the fast E-step and name it on stderr -- the last name
// before a "Memory access fault" is the kernel that did it (diagnostic; serialises everything)
inline bool dbg_sync_on() { static const bool on = getenv("PSMC_HIP_DEBUG_SYNC") != nullptr; return on; }
#define PSMC_DBG(NAME, A, B, C) do { if (psmc::dbg_sync_on()) { hipError_t e_ = hipDeviceSynchronize(); \
	fprintf(stderr, "[psmc_hip] %s(%d, %d, %d): %s\n", NAME, (int)(A), (int)(B), (int)(C), hipGetErrorString(e_)); fflush(stderr); } } while (0)

constexpr int RED_ROWS = 64;

int launch_exact(const EstepLaunch &p);
constexpr int LKP_DIV = 16, LKP_MIN = 64; // k_lk_products: an entry of L bins may log L / LKP_DIV + LKP_MIN products before the host takes over
int launch_lk_products(hipStream_t st, const EstepLaunch &p, const double *d_s, const int64_t *d_lk_off, double *d_out); // estep_exact.hip: hmm_lk's products on the device
int launch_exact_wide(const EstepLaunch &p); // estep_wide.hip: 129 .. 1024 states
int launch_post_decode_wide(hipStream_t st, const double *f, const double *b, const double *s, int64_t off, int L, int n, int S, int32_t *path, double *maxp);
int launch_post_full_wide(hipStream_t st, const double *a, const double *e, const uint8_t *obs, const double *f, const double *b, const double *s,
                          int64_t off, int L, int n, int S, double *post, double *recomb);
int launch_post_counts_wide(hipStream_t st, const double *f, const double *b, const double *s, int64_t off, int min_l, const int32_t *cnt1,
                            int n_cnt, int n, int S, double *cnt);
int launch_fast(const EstepLaunch &p, FastReport *rep);
void launch_fwd_struct(const EstepLaunch &p, hipStream_t st, int which, int first, int n_items);
void launch_bwd_struct(const EstepLaunch &p, hipStream_t st, int which, int first, int n_items);
void launch_compact(const EstepLaunch &p, hipStream_t st, bool bwd);
void launch_gather_prev(const EstepLaunch &p, hipStream_t st, int first, int n_items, double *prevx); // estep_struct.hip
int launch_tile_allmiss(hipStream_t st, const uint8_t *d_obs, const Chunk *d_chunks, int n_chunks, int *d_flags); // estep_struct.hip
void launch_walks(const EstepLaunch &p, hipStream_t st);
void launch_gate(hipStream_t st, const int *ctr, int want);
int walk_blocks(const EstepLaunch &p);
void launch_kchain(const EstepLaunch &p, hipStream_t st_cols, hipStream_t st_chain, hipEvent_t ev_cols);
void launch_sweeps(const EstepLaunch &p, hipStream_t st, int ff, int nf, int fb, int nb, bool top_only);
void launch_bwd_count(const EstepLaunch &p, hipStream_t st, int list, bool redo, bool all_from_bentry = false);
void launch_bwd_acc(const EstepLaunch &p, hipStream_t st, int which, int first, int n);
void launch_reduce_factored(const EstepLaunch &p, hipStream_t st);
int launch_post_decode(hipStream_t st, const double *f, const double *b, const double *s, int64_t off, int L, int n,
                       int ns, int32_t *path, double *maxp);
int launch_post_full(hipStream_t st, const double *a, const double *e, const uint8_t *obs, const double *f, const double *b,
                     const double *s, int64_t off, int L, int n, int ns, double *post, double *recomb);
int launch_post_counts(hipStream_t st, const double *f, const double *b, const double *s, int64_t off, int min_l,
                       const int32_t *cnt1, int n_cnt, int n, int ns, double *cnt);
int run_selftest(hipStream_t stream, unsigned *d_flags);
int run_microbench(hipStream_t stream, double *d_out);
int run_pipe_probe(hipStream_t stream, double *d_out, int n_waves, unsigned mask, int rounds);
int run_pipe_probe2(hipStream_t stream, double *d_out, const int *kinds8, int rounds, const void *gsrc);
int run_place_probe(hipStream_t stream, double *d_out, int n_waves, int wpb, int steps);
int run_stream_probe(hipStream_t stream, const double *src, double *dst, size_t n);
int run_hbm_probe(hipStream_t stream, int which, double *a, double *b, size_t bytes);
int run_load_probe(hipStream_t stream, double *d_out, int n_waves, int steps);
int run_load_probe_st(hipStream_t stream, double *d_out, int n_waves, int steps, double *tbl, int store_steps, int mode);

} // namespace psmc
