// api_fast.hip -- fast mode behind the C-ABI: the tile plan of a segment set (plan_fast), the sweep items, runs and tile lists of
// one E-step (build_items), what a context learns from the tiles that needed a repair (learn_groups), the launch of one fast
// E-step (enqueue_fast -> launch_fast in estep_fast.hip) and the fast entry points of include/psmc_hip.h.
#include "psmc_hip_ctx.h"

// ---------------------------------------------------------------- fast mode
// Buffers that do not depend on the tiling (reduction staging, result vector, read-back words).  Kept apart from
// plan_fast so that an entry point can make sure its result buffer exists BEFORE the first E-step without planning:
// the plan depends on whether the matrix has the PSMC form, which only stage_params() finds out.
int ensure_fast_buffers(psmc_hip_ctx *c)
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

// the tile length the planner picks for `bins` bins in n_work segments (see plan_fast)
int auto_tile_len(const psmc_hip_ctx *c, int64_t bins, size_t n_work, bool st)
{
	const int64_t ROUND1 = 4096;
	int64_t want = st ? c->struct_tiles : c->target_waves; // about target_waves (dense: 1 tile per wave) or struct_tiles (4 per wave) tiles, never below 256 bins
	if (st && !c->struct_tiles_set && bins < 2 * ROUND1 * (int64_t)std::max(c->warmup, 1))
		want = std::max<int64_t>(ROUND1 - (int64_t)n_work, ROUND1 / 2); // every segment ends in a ragged tile: stay inside the round
	const int T = (int)((bins + want - 1) / want);
	return std::max(256, (T + 63) & ~63);
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
	if (T <= 0) T = auto_tile_len(c, bins, c->work.size(), st);
	c->chunks.clear(); c->chunk_seg.clear(); c->chunk_idx.clear();
	for (size_t w = 0; w < c->work.size(); ++w) {
		const int32_t s = c->work[w];
		for (int32_t lo = 1; lo <= c->L[s]; lo += T) {
			c->chunk_seg.push_back(s); c->chunk_idx.push_back((lo - 1) / T);
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
	c->glue_f.assign(nc, 0); c->glue_b.assign(nc, 0); // a new tiling forgets what was learned ...
	if (c->parent && c->parent->share_T == T && st) { // ... but a batch replicate starts from what its predecessors learned at the same tiles
		const psmc_hip_ctx *P = c->parent;
		for (int b = 0; b < nc; ++b) {
			const int sg = c->chunk_seg[b], ix = c->chunk_idx[b];
			if (sg >= (int)P->sh_glue_f.size() || ix >= (int)P->sh_glue_f[sg].size()) continue;
			c->glue_f[b] = P->sh_glue_f[sg][ix]; c->glue_b[b] = P->sh_glue_b[sg][ix];
			c->chunks[b].wf = std::max(c->chunks[b].wf, P->sh_wf[sg][ix]); c->chunks[b].wb = std::max(c->chunks[b].wb, P->sh_wb[sg][ix]);
		}
	}
	c->items_dirty = true;
	int rc;
	c->gap.assign(nc, 0);
	if (nc > c->chunk_cap) {
		if ((rc = dev_alloc(c, &c->d_chunks, (size_t)nc))) return rc;
		if ((rc = dev_alloc(c, &c->d_entry, (size_t)nc * c->ns))) return rc;
		if ((rc = dev_alloc(c, &c->d_bexit, (size_t)(nc + 1) * c->ns))) return rc;
		if ((rc = dev_alloc(c, &c->d_bentry, (size_t)nc * c->ns))) return rc;
		if ((rc = dev_alloc(c, &c->d_dirty, (size_t)2 * nc))) return rc;
		if ((rc = dev_alloc(c, &c->d_touch, (size_t)3 * nc))) return rc; // touch_f | touch_b | fmerge
		if ((rc = dev_alloc(c, &c->d_finv, (size_t)nc))) return rc;
		// (what only the round-6 options use is allocated only with them: a fast bootstrap keeps a plan per replicate, and a hundred pairs of
		// pinned buffers cost seconds to make and to free)
		if (c->h_mlen) { (void)hipHostFree(c->h_mlen); c->h_mlen = nullptr; c->m_mlen = nullptr; }
		if (c->h_mis) { (void)hipHostFree(c->h_mis); c->h_mis = nullptr; c->m_mis = nullptr; }
		if (c->prev_start && (rc = dev_alloc(c, &c->d_prevx, (size_t)nc * c->ns))) return rc;
		if ((c->merge || c->adapt) &&
		    (hipHostMalloc((void **)&c->h_mlen, (size_t)nc * sizeof(int), hipHostMallocMapped) != hipSuccess ||
		     hipHostGetDevicePointer((void **)&c->m_mlen, c->h_mlen, 0) != hipSuccess ||
		     hipHostMalloc((void **)&c->h_mis, (size_t)2 * nc * sizeof(double), hipHostMallocMapped) != hipSuccess ||
		     hipHostGetDevicePointer((void **)&c->m_mis, c->h_mis, 0) != hipSuccess))
			return fail(c, PSMC_HIP_ENOMEM, "hipHostMalloc (mapped)");
		if ((rc = dev_alloc(c, &c->d_LLpart, (size_t)nc))) return rc;
		if ((rc = dev_alloc(c, &c->d_items, (size_t)32 * nc + 64))) return rc; // (+ 4 nc: matrix slot of every KcTile | the KcTile that computes a slot; + 2 nc: the fix pass's tiles)
		if ((rc = dev_alloc(c, &c->d_ftiles, (size_t)2 * (nc + 16)))) return rc;
		if (c->h_ritems) { (void)hipHostFree(c->h_ritems); c->h_ritems = nullptr; }
		if (hipHostMalloc((void **)&c->h_ritems, (size_t)4 * nc * sizeof(int), hipHostMallocMapped) != hipSuccess ||
		    hipHostGetDevicePointer((void **)&c->m_ritems, c->h_ritems, 0) != hipSuccess)
			return fail(c, PSMC_HIP_ENOMEM, "hipHostMalloc (mapped)");
		c->chunk_cap = nc;
	}
	// (the partial counts, d_Cpart, are sized per E-step by what its back half writes: enqueue_fast)
	if ((rc = dev_alloc(c, &c->d_Epart, (size_t)nc * c->n_sub_used * 3 * c->ns))) return rc;
	if ((rc = ensure_fast_buffers(c))) return rc;
	HIPCHK(c, hipMemcpy(c->d_chunks, c->chunks.data(), sizeof(Chunk) * nc, hipMemcpyHostToDevice));
	if (st && c->gap_tiles) {
		// Tiles of missing data (round 5; estep_struct.hip k_tile_allmiss).  A FULL tile inside its segment (not the first, not the last: position 1
		// and position L are special) that is all `N` is a "gap tile": glued to its neighbours now, in both directions, together with the first
		// tile after the gap in either direction -- their start vectors hang on what entered the gap however long it is.
		std::vector<int> fl(nc, 0);
		if (launch_tile_allmiss(c->stream, c->d_obs, c->d_chunks, nc, c->d_dirty) != 0) return fail(c, PSMC_HIP_EDEVICE, "k_tile_allmiss", hipGetLastError());
		HIPCHK(c, hipMemcpyAsync(fl.data(), c->d_dirty, sizeof(int) * nc, hipMemcpyDeviceToHost, c->stream));
		HIPCHK(c, hipMemsetAsync(c->d_dirty, 0, sizeof(int) * nc, c->stream)); // (borrowed as scratch: back to what an allocation holds)
		HIPCHK(c, hipStreamSynchronize(c->stream));
		for (int b = 0; b < nc; ++b) {
			const Chunk &ch = c->chunks[b];
			c->gap[b] = fl[b] && ch.lo > 1 && ch.hi < ch.L && ch.hi - ch.lo + 1 == T;
		}
		auto same = [&](int x, int y) { return x >= 0 && y < nc && c->chunks[x].off == c->chunks[y].off; };
		for (int b = 0; b < nc; ++b) {
			if (c->gap[b]) { if (same(b - 1, b)) c->glue_f[b] = 1; if (same(b, b + 1)) c->glue_b[b] = 1; }
			if (b > 0 && c->gap[b - 1] && same(b - 1, b)) c->glue_f[b] = 1;      // the first tile after a gap, forward
			if (b + 1 < nc && c->gap[b + 1] && same(b, b + 1)) c->glue_b[b] = 1;  // ... and backward
		}
	}
	c->plan_dirty = false; c->chunks_dirty = false; c->prev_ok = false;
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
	// "group_cap" bounds the bins of a run that carry DATA: a run of missing data costs its tiles one shared transfer matrix, however long it is
	const bool gaps = c->gap_tiles && (int)c->gap.size() == nc;
	std::vector<int64_t> eff(nc + 1, 0);
	for (int b = 0; b < nc; ++b) eff[b + 1] = eff[b] + ((gaps && c->gap[b]) ? 0 : c->chunks[b].hi - c->chunks[b].lo + 1);
	auto span_ok = [&](int b, int e) { return eff[e + 1] - eff[b] <= (int64_t)c->group_cap; }; // tiles b .. e
	for (int b = 0; b < nc;) { // forward: head b, members b+1.. while glued
		int e = b + 1;
		while (e < nc && c->glue_f[e] && same_seg(e, b) && span_ok(b, e)) ++e;
		if (e - b > 1) for (int t = b; t < e; ++t) in_run[t] = in_run_f[t] = 1;
		else // a bulk item: the unglued tiles that follow, up to the next coarse boundary or the head of a run
			while (e < nc && e - b < cf && same_seg(e, b) && idx[e] % cf != 0 && !c->glue_f[e] && !(e + 1 < nc && c->glue_f[e + 1] && same_seg(e + 1, e))) ++e;
		gf.push_back({b, e - b});
		b = e;
	}
	for (int b = 0; b < nc;) { // backward: tiles b..e-1, top tile e-1; glue_b[t] ties t to t+1
		int e = b + 1;
		while (e < nc && c->glue_b[e - 1] && same_seg(e, b) && span_ok(b, e) &&
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
	c->order_f.clear();
	for (const auto &it : kf) if (!in_run_f[it.second.first]) c->order_f.push_back(it.second.first);
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
	{ // the forward fix pass (estep_struct.hip FwdCtl) right after the bulk sweep: every tile outside the forward runs whose predecessor is outside
	  // them too (a bulk tile above a run is checked against a row the runs' path writes later: left to the verify that follows the back half)
		std::vector<int> fx;
		for (int b = 0; b < nc; ++b)
			if (!in_run_f[b] && !(b > 0 && in_run_f[b - 1] && same_seg(b - 1, b))) { fx.push_back(b); fx.push_back(1); }
		c->n_fix_f = (int)fx.size() / 2;
		if (!fx.empty()) HIPCHK(c, hipMemcpy(c->d_items + (size_t)30 * nc, fx.data(), sizeof(int) * fx.size(), hipMemcpyHostToDevice));
	}
	// Walk lists and transfer-matrix chains.  A run of >= kc_min tiles is a "chain run": a walk delivers the start vector
	// of its head tile (the usual speculative warm-up, nothing more: walk item with count <= 0, see k_walk1_struct; round 1
	// and most of round 2 also walked THROUGH the head tile, 3712 dependent steps whose result nobody read -- measured
	// neutral all the same, 12.95 vs 13.07 ms: the longest walk is a head with a doubled warm-up, 6144 steps), and the
	// boundary vectors of its other tiles come from the transfer matrices, the head's first.
	//   d_items + 12nc: wl_f (2nc) | wl_b (2nc) | kc tiles (4nc: KcTile) | runs (4nc + : KcRun)
	std::vector<int> wl((size_t)4 * nc, 0), kc, runs_f, runs_b, kslot, kuniq; // kslot[j]: matrix slot of KcTile j; kuniq[u]: the KcTile that computes slot u
	int gap_slot[2] = {-1, -1}; // the slot every gap tile of a direction shares
	auto push_kc = [&](int t, int dir) {
		const int j = (int)kc.size() / 2;
		kc.push_back(t); kc.push_back(dir);
		// (one shared matrix only when every gap tile covers the same steps: the backward matrix starts at ((lo + 3) & ~3) + 1 and the chain kernel
		// adds the tile's own last steps, T in all only if every gap tile has the same lo mod 4, i.e. T % 4 == 0 -- ADVICE r5; a tile length the
		// caller chose otherwise gets a matrix per tile)
		if (gaps && c->gap[t] && c->chunk_used % 4 == 0) {
			if (gap_slot[dir] < 0) { gap_slot[dir] = (int)kuniq.size(); kuniq.push_back(j); }
			kslot.push_back(gap_slot[dir]);
		} else { kslot.push_back((int)kuniq.size()); kuniq.push_back(j); }
	};
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
			// what a run's matrices cost: one column kernel per tile that carries data (gap tiles share a matrix computed once per direction)
			int cost = 0;
			for (int tt = first; tt < first + count; ++tt) cost += (gaps && c->gap[tt] && c->chunk_used % 4 == 0) ? 0 : 1;
			cost = std::max(cost - 1, 1);
			const bool chain = chains && count >= kc_min && cost <= budget;
			if (chain) budget -= cost;
			if (chain && !bwd && c->chunks[first].lo == 1) {
				// position 1 is an initial condition, not a step: there is no X_0 for a transfer matrix to start from.
				// Walk through the first tile as well and chain from the second one.
				w[2 * nw] = first; w[2 * nw + 1] = (head_count == 0 && count - 1 >= 2) ? -1 : 2; ++nw; // -1: through the first tile, up to the head's start vector
				first += 1; count -= 1;
				if (count >= 2) {
					rv.push_back(first); rv.push_back(count); rv.push_back((int)(kc.size() / 2)); rv.push_back(0);
					for (int t = first; t < first + count - 1; ++t) push_kc(t, 0);
				}
			} else if (chain) {
				w[2 * nw] = bwd ? first + count - 1 : first; w[2 * nw + 1] = head_count; ++nw; // the head tile's start vector
				rv.push_back(first); rv.push_back(count); rv.push_back((int)(kc.size() / 2)); rv.push_back(0);
				if (!bwd) for (int t = first; t < first + count - 1; ++t) push_kc(t, 0);
				else for (int t = first + count - 1; t > first; --t) push_kc(t, 1);
			} else { w[2 * nw] = first; w[2 * nw + 1] = (head_count == 0 && !bwd) ? -(count - 1) : count; ++nw; } // k_walk1_struct forward: stop where the last tile starts
		}
	};
	add_runs(kf, c->n_long_f, false);
	add_runs(kb, c->n_long_b, true);
	c->n_chain_f = (int)runs_f.size() / 4; c->n_chain_b = (int)runs_b.size() / 4; c->n_kc = (int)kc.size() / 2; c->n_kuniq = (int)kuniq.size();
	if ((size_t)c->n_kc > (size_t)2 * nc || (size_t)(c->n_chain_f + c->n_chain_b) > (size_t)nc) return fail(c, PSMC_HIP_ESTATE, "build_items: list overflow");
	HIPCHK(c, hipMemcpy(c->d_items + (size_t)12 * nc, wl.data(), sizeof(int) * wl.size(), hipMemcpyHostToDevice));
	if (c->n_kc > 0) {
		std::vector<int> runs(runs_f); runs.insert(runs.end(), runs_b.begin(), runs_b.end());
		HIPCHK(c, hipMemcpy(c->d_items + (size_t)16 * nc, kc.data(), sizeof(int) * kc.size(), hipMemcpyHostToDevice));
		HIPCHK(c, hipMemcpy(c->d_items + (size_t)20 * nc, runs.data(), sizeof(int) * runs.size(), hipMemcpyHostToDevice));
		HIPCHK(c, hipMemcpy(c->d_items + (size_t)26 * nc, kslot.data(), sizeof(int) * kslot.size(), hipMemcpyHostToDevice));
		HIPCHK(c, hipMemcpy(c->d_items + (size_t)28 * nc, kuniq.data(), sizeof(int) * kuniq.size(), hipMemcpyHostToDevice));
		const size_t nsub = kcol2_on(c) ? (size_t)c->kc_sub_used : 1; // k_kcol2_struct: kc_sub matrices per tile
		const size_t need = (size_t)c->n_kuniq * nsub * ((size_t)c->ns * c->ns + c->ns); // the matrices (one per slot), then one exponent per column
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
	if (c->parent && c->parent->share_T == c->chunk_used && (!c->flagged_f.empty() || !c->flagged_b.empty())) { // share it with the replicates that plan later
		psmc_hip_ctx *P = c->parent;
		auto put = [&](int b) {
			if (b < 0 || b >= nc) return;
			const int sg = c->chunk_seg[b], ix = c->chunk_idx[b];
			if (sg >= (int)P->sh_glue_f.size() || ix >= (int)P->sh_glue_f[sg].size()) return;
			P->sh_glue_f[sg][ix] |= c->glue_f[b]; P->sh_glue_b[sg][ix] |= c->glue_b[b];
			P->sh_wf[sg][ix] = std::max(P->sh_wf[sg][ix], c->chunks[b].wf); P->sh_wb[sg][ix] = std::max(P->sh_wb[sg][ix], c->chunks[b].wb);
		};
		for (int b : c->flagged_f) put(b);
		for (int b : c->flagged_b) put(b);
	}
}

// Per-tile forward warm-ups (round 6).  Round 3 built this once and took it out again: a tile whose shrunken warm-up overshot cost a repair
// round of 8 ms (the whole tile again, then its group of the counts again), so the plan had to be sized by its slowest tiles.  With the fix
// pass (estep_struct.hip FwdCtl) a tile that falls short is rewritten for the blocks it was short by, before the back half starts, and
// nothing is counted twice -- so every speculating tile can follow what ITS stretch of the sequence needs: the pass leaves the mismatch of
// every speculation and the blocks every repair took (host-mapped memory); a tile that failed gets the bins its repair needed, plus a margin; a
// tile more than a decade inside the tolerance gives up part of the surplus, an eighth of its warm-up at most per E-step (at 1e-16 the
// measurement is at its floor: all it says is "at least four decades").  Tiles anchored at their segment's start, members and heads of glued
// runs keep what they have.  What this buys is bounded by profiles/r06_warmup_sensitivity.txt: half the warm-up at no cost at all is 0.35 ms
// of 12.25 (the forward sweep ends earlier, the runs' path then overlaps the counts instead).
static void adapt_warmups(psmc_hip_ctx *c)
{
	const int nc = (int)c->chunks.size();
	const int cap = c->warmup << c->warm_shift_used, floor_w = std::min(c->warmup, 256);
	const double tol = c->warm_tol;
	bool changed = false;
	for (int b = 0; b < nc; ++b) {
		const double m = c->h_mis[b];
		if (!(m > 0.0)) continue; // not checked (anchored, first tile), or not a speculation (inside a bulk item: bitwise its neighbour's vector)
		if (c->glue_f[b] || (b + 1 < nc && c->glue_f[b + 1] && c->chunks[b + 1].off == c->chunks[b].off)) continue;
		Chunk &ch = c->chunks[b];
		if (ch.lo - ch.wf <= 1) continue; // the warm-up reaches the start of the segment: exact
		int w = ch.wf;
		if (m > tol) {
			const int ml = c->h_mlen[b]; // blocks the fix pass rewrote; -1: the whole tile
			w = std::min(std::max(cap, w), w + (ml > 0 ? 16 * ml + 64 : w));
		} else {
			const double slack = log10(tol / std::max(m, 1e-17)), keep = 2.0; // decades inside the tolerance a speculation may keep; the parameters of the next E-step are other parameters
			if (slack > keep) w = std::max(floor_w, w - (((int)std::min((slack - keep) * 160.0, w / 8.0) + 15) & ~15));
		}
		if (w != ch.wf) { ch.wf = w; changed = true; }
	}
	if (changed) {
		c->chunks_dirty = true;
		// The four rows of a wave should end together: the items were sorted by step count when the lists were built, and the warm-ups have
		// moved since.  Issue slots the bulk forward sweep now wastes on rows that are done while another row of their wave is not: when
		// they pass 3 % of its work, sort again (build_items: a few hundred microseconds of host time).
		int64_t total = 0, waste = 0;
		for (size_t i = 0; i + 3 < c->order_f.size(); i += 4) {
			int64_t s4 = 0, mx = 0;
			for (int r = 0; r < 4; ++r) { const Chunk &ch = c->chunks[c->order_f[i + r]]; const int64_t st = (int64_t)ch.hi - std::max(1, ch.lo - ch.wf) + 1; s4 += st; mx = std::max(mx, st); }
			total += 4 * mx; waste += 4 * mx - s4;
		}
		if (total > 0 && waste * 100 > total * 3) c->items_dirty = true;
	}
}

int enqueue_fast(psmc_hip_ctx *c, const double *a, const double *e, const double *a0, double *d_out,
                        hipStream_t st)
{
	HIPCHK(c, hipSetDevice(c->device));
	if (c->n_seg < 1) return fail(c, PSMC_HIP_ESTATE, "estep: no segments loaded");
	int rc;
	const unsigned long long serial0 = dbg_root(c)->tab_serial; // has anybody used the tables since this context's last fast E-step?
	if ((rc = ensure_tables(c, false))) return rc;
	c->tables_batch = false;
	if ((rc = stage_params(c, a, e, a0, st))) return rc; // also decides whether the structured sweeps apply
	// the backward table: only for the unfused back half (the fused and the factored one never store bt)
	if (!c->want_factored && !(c->use_struct && fused_counts(c)) && (rc = ensure_tables(c, true))) return rc;
	if (c->ns == 128 && !c->use_struct)
		return fail(c, PSMC_HIP_ENOTSUP, "fast mode beyond 64 states needs a transition matrix of the PSMC form (structured sweeps)");
	if (c->want_factored && !c->use_struct)
		return fail(c, PSMC_HIP_ENOTSUP, "factored statistics need a transition matrix of the PSMC form");
	{
		const double t0 = dbg_now();
		if ((c->plan_dirty || c->planned_struct != c->use_struct) && (rc = plan_fast(c))) return rc;
		dbg_root(c)->dbg_acc[2] += dbg_now() - t0;
	}
	EstepLaunch p;
	fill_common(c, p, st);
	p.d_chunks = c->d_chunks; p.n_chunks = (int)c->chunks.size(); p.warmup = c->warmup; p.n_sub = c->n_sub_used;
	// the fused back half takes both directions of the two-phase plan, the factored one (item lists, two waves per SIMD) the forward one
	// 1: both directions; 2: backward only (the fused back half runs as two launches anyway, so its second list can start
	// from the exit vectors of the first at no cost in scheduling, and half of the backward warm-up pass disappears)
	const bool two_phase_bwd = c->two_phase_used >= 1 && p.fused == 1; // the fused back half only: the factored one is a single pass over item lists
	p.merge1 = c->merge1_used; p.merge_order = c->merge_order >= 0 ? c->merge_order : (c->chunk_used < c->warmup ? 1 : 0);
	if (c->chunks_dirty) { // learned / adapted warm-ups (learn_groups, adapt_warmups) reach the device before the next launch reads them
		// (from a pinned copy, on the E-step's stream: a blocking copy of pageable memory waits for every kernel on the device, and with
		// per-tile warm-ups this happens before every E-step)
		psmc_hip_ctx *R = dbg_root(c); // (one staging buffer per root context: its replicates run one after the other)
		if (R->h_chunks_cap < c->chunks.size()) {
			if (R->h_chunks) { (void)hipHostFree(R->h_chunks); R->h_chunks = nullptr; }
			if (hipHostMalloc((void **)&R->h_chunks, sizeof(Chunk) * c->chunks.size(), hipHostMallocDefault) != hipSuccess) return fail(c, PSMC_HIP_ENOMEM, "hipHostMalloc");
			R->h_chunks_cap = c->chunks.size();
		}
		HIPCHK(c, hipStreamSynchronize(st)); // (the previous copy out of the same buffer is done)
		memcpy(R->h_chunks, c->chunks.data(), sizeof(Chunk) * c->chunks.size());
		HIPCHK(c, hipMemcpyAsync(c->d_chunks, R->h_chunks, sizeof(Chunk) * c->chunks.size(), hipMemcpyHostToDevice, st));
		c->chunks_dirty = false;
	}
	// coarse bulk items: the fused and the factored back half only (their backward pass of phase 1 leaves start vectors, no table)
	const int coarse = (c->use_struct && p.fused != 0) ? c->coarse_used : 1;
	{
		const double t0 = dbg_now();
		if (c->use_struct && (c->items_dirty || c->items_two_phase != (two_phase_bwd ? 2 : 0) || c->items_coarse != coarse) && (rc = build_items(c, two_phase_bwd, coarse))) return rc;
		dbg_root(c)->dbg_acc[3] += dbg_now() - t0;
	}
	// auto: the factored statistics of a genome-sized input (issue-bound at three waves per SIMD); a shard-sized one runs one wave per SIMD,
	// where the longer step of the 8 x 8 form costs more than its fewer instructions save (3.75 M bins: 3.23 vs 2.96 ms)
	p.lanes8 = (c->lanes8 >= 0 ? c->lanes8 != 0 && p.fused != 0 : p.fused == 2 && p.n_chunks > 4096) && c->ns == 64 ? 1 : 0;
	{
		// Partial counts.  Round 4 gave every plan n_tiles x ns^2 doubles (266 MB for a genome's 8127 tiles) whatever its back half: the unfused
		// counts kernel writes that much, the fused one a partial per GROUP of four tiles (a quarter), the factored one seven vectors per
		// tile (a ninth) -- and a fast bootstrap keeps a plan per replicate: 26 GB for 100 of them, cleared by the driver on allocation
		// (profiles/r05_boot_first_iteration.txt).  Sized here, per E-step, for the back half that runs; grows, never shrinks.
		const size_t nc_ = c->chunks.size(), S_ = (size_t)c->ns;
		const size_t need = p.fused == 2 ? nc_ * 7 * S_ + 64 : (p.fused == 1 ? (nc_ / 4 + 8) * S_ * S_ : nc_ * (size_t)c->n_sub_used * S_ * S_);
		if (c->cpart_cap < need) { if ((rc = dev_alloc(c, &c->d_Cpart, need))) return rc; c->cpart_cap = need; }
	}
	{
		// the forward fix pass: where the back half knows about its merge records (the fused one of a 64-state model), tiles of whole 16-bin blocks
		c->merge_used = c->merge != 0 && c->h_mlen && c->use_struct && p.fused == 1 && c->ns == 64 && c->chunk_used % 16 == 0;
		p.merge = c->merge_used ? 1 : 0;
		p.d_fmerge = c->d_touch + 2 * (size_t)p.n_chunks; p.d_finv = c->d_finv; p.m_mlen = c->m_mlen; p.h_mlen = c->h_mlen; p.m_mis = c->m_mis;
		p.d_fix_f = c->d_items + 30 * (size_t)p.n_chunks; p.n_fix_f = c->n_fix_f;
		if (c->h_mlen) memset(c->h_mlen, 0, sizeof(int) * p.n_chunks); // (the E-step that wrote it is over: launch_fast reads the verify counts before it returns)
		if (c->h_mis) for (int b = 0; b < p.n_chunks; ++b) c->h_mis[b] = -1.0; // "not measured": the tiles neither fix launch looks at keep their warm-ups
		// forward warm-ups from the previous E-step's X: same plan, same kind of table, and nobody else has used the tables in between
		const bool prev = c->prev_start && c->d_prevx && c->prev_ok && c->use_struct && c->prev_serial == serial0 && c->prev_f == c->d_f && c->prev_ckpt == p.ckpt;
		p.d_prevx = prev ? c->d_prevx : nullptr;
		c->prev_ok = false;
	}
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
	p.d_kslot = c->d_items + 26 * (size_t)p.n_chunks; p.d_kuniq = c->d_items + 28 * (size_t)p.n_chunks; p.n_kuniq = c->n_kuniq;
	p.kcol_prio = c->kcol_prio;
	p.d_Kcol = c->d_Kcol; p.kc_sub = kcol2_on(c) ? c->kc_sub_used : 1;
	p.d_Kexp = c->d_Kcol ? c->d_Kcol + (size_t)c->n_kuniq * p.kc_sub * c->ns * c->ns : nullptr; p.stream5 = c->stream5;
	for (int i = 0; i < 14; ++i) p.evx[i] = c->evx[i];
	p.d_LLpart = c->d_LLpart; p.d_stage = c->d_stage; p.d_stats = d_out; p.d_warm = c->d_warm;
	p.tiny_total = (double)c->sel.size() * HMM_TINY_H;
	if (p.d_prevx) launch_gather_prev(p, st, c->n_long_f, c->n_items_f - c->n_long_f, c->d_prevx); // before the sweep overwrites those rows
	{
		const double t0 = dbg_now();
		if (launch_fast(p, &c->report) != 0) return fail(c, PSMC_HIP_EDEVICE, "launch_fast", hipGetLastError());
		psmc_hip_ctx *R = dbg_root(c);
		R->dbg_acc[4] += dbg_now() - t0;
		R->dbg_acc[6] += c->report.fwd_rounds + c->report.bwd_rounds; R->dbg_acc[7] += c->report.fwd_tiles + c->report.bwd_tiles;
	}
	if (getenv("PSMC_HIP_DEBUG_FLAGGED")) { // which tiles the verify / repair net had to touch in this E-step, and what the plan knew about them
		auto dump = [&](const char *dir, const std::vector<int> &fl, const std::vector<uint8_t> &glue, bool bwd) {
			std::vector<int> u(fl); std::sort(u.begin(), u.end()); u.erase(std::unique(u.begin(), u.end()), u.end());
			fprintf(stderr, "[psmc_hip] flagged %s: %zu tile repairs, %zu distinct tiles, rounds %d:", dir, fl.size(), u.size(), bwd ? c->report.bwd_rounds : c->report.fwd_rounds);
			for (size_t i = 0; i < u.size() && i < 400; ++i) {
				const int b = u[i]; const Chunk &ch = c->chunks[b];
				int seg = 0; while (seg + 1 < c->n_seg && c->off[seg + 1] <= ch.off) ++seg;
				fprintf(stderr, " [t%d seg%d %d..%d w%d g%d x%d m%.1e ml%d]", b, seg, ch.lo, ch.hi, bwd ? ch.wb : ch.wf, (int)glue[b], (int)std::count(fl.begin(), fl.end(), b),
				        c->h_mis ? c->h_mis[(bwd ? c->chunks.size() : 0) + b] : 0.0, (!bwd && c->h_mlen) ? c->h_mlen[b] : 0);
			}
			fprintf(stderr, "\n");
		};
		dump("fwd", c->flagged_f, c->glue_f, false); dump("bwd", c->flagged_b, c->glue_b, true);
	}
	if (!c->report.converged) return fail(c, PSMC_HIP_ECONVERGE, "fast mode: tile boundaries did not converge within max_rounds");
	if (c->merge_used && c->h_mlen)
		for (size_t b = 0; b < c->chunks.size(); ++b) {
			c->report.merged += c->h_mlen[b] > 0;
			// what the fix pass had to rewrite to its end -- or, without per-tile warm-ups, at all -- is what the plan learns from, like a
			// tile the verify rounds flag: a longer warm-up of its own first, then glued to its neighbour (learn_groups)
			if (c->h_mlen[b] < 0 || (c->h_mlen[b] > 0 && !c->adapt)) c->flagged_f.push_back((int)b);
		}
	if (c->use_struct && c->adapt && c->merge_used && c->h_mis) adapt_warmups(c); // (only where falling short is cheap)
	if (c->use_struct && c->learn) learn_groups(c);
	if (c->use_struct) { c->prev_ok = true; c->prev_serial = dbg_root(c)->tab_serial; c->prev_f = c->d_f; c->prev_ckpt = p.ckpt; }
	return 0;
}

int read_warm(psmc_hip_ctx *c, hipStream_t st)
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

extern "C" int psmc_hip_fast_repairs(psmc_hip_ctx *c, int out[6])
{
	if (!c || !out) return PSMC_HIP_EINVAL;
	out[0] = c->report.fwd_rounds; out[1] = c->report.bwd_rounds;
	out[2] = c->report.fwd_tiles; out[3] = c->report.bwd_tiles;
	out[4] = c->report.merged; out[5] = c->report.recounted;
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

int estep_fast(psmc_hip_ctx *c, const double *a, const double *e, const double *a0, double *A, double *E,
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
