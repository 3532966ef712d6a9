// api_batch.hip -- psmc_hip_estep_batch (bootstrap replicates over ONE loaded trunk set; replaces the
// xargs farm of the reference's README:57-62): exact mode as launch groups that share the table memory, fast mode as one
// plan-holding child context per replicate.
#include "psmc_hip_ctx.h"

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

// Table bins ONE launch of an exact batch may use: "batch_bins", or what 0.9 of the device memory that is free or already in this
// context's tables holds.  (Round 3 took 0.9 of the free memory PLUS all of what the context held: the second call saw a larger
// capacity than the first, re-planned its groups and re-allocated 250 GB of tables -- 8 s; profiles/r04_boot_breakdown.txt.)
// all_bins (> 0: known) = the table bins of every entry of the call: a batch without the f table whose entries do not fit one
// launch keeps the scale factors of ALL of them in a call-wide table (one forward pass, see below) -- 8 bytes per bin of the whole
// call, taken out of the budget BEFORE the per-launch capacity is computed (it was a fixed 1/64 of the capacity, right
// for four launches and too little beyond seven).
static int batch_capacity(psmc_hip_ctx *c, int64_t *cap, bool refwd, int64_t all_bins)
{
	*cap = c->batch_bins;
	if (*cap > 0) return 0;
	size_t fr = 0, tot = 0;
	HIPCHK(c, hipMemGetInfo(&fr, &tot));
	const double S = (double)c->ns, per_bin = S * 8.0 * (refwd ? 1.0 : 2.0) + 8.0;
	const double held = (double)c->tab_bins * (S * 8.0 * ((c->have_b ? 1.0 : 0.0) + (c->d_f ? 1.0 : 0.0)) + 8.0 + (c->d_sb ? 8.0 : 0.0)) + (double)c->b2_alloc * S * 8.0;
	// (the call-wide scale-factor table stays allocated between calls: count it as free, or the second call would see a smaller
	// capacity than the first and cut other launches)
	double budget = ((double)fr + held + (double)c->s_all_cap * 8.0) * 0.9;
	*cap = (int64_t)(budget / per_bin) - 256;
	if (refwd && (all_bins <= 0 || all_bins > *cap)) { // several launches: room for the call-wide scale factors
		const double s_all = all_bins > 0 ? (double)all_bins * 8.0 : budget / 65.0; // unknown size: what four launches need
		budget -= s_all;
		*cap = (int64_t)(budget / per_bin) - 256;
	}
	if (*cap < 1) return fail(c, PSMC_HIP_ENOMEM, "estep_batch: no device memory left for tables");
	return 0;
}

// Does this batch run without the f table (k_expect_exact_rf: 64 states)?  "exact_refwd" 1 / 0: yes / no.  Auto: only when the f and b
// tables of all its entries (need_bins; <= 0: unknown) would NOT fit one launch -- the recompute pass runs at the forward sweep's
// latency (0.83 instead of 0.45 us per bin of the longest segment), which pays when it halves the number of launches (100
// replicates of a genome: 11.9 against 13.2 s per EM iteration) and costs when one would have done (16 replicates: 2.4 against 1.9 s).
// The decision is made ONCE per table reservation: psmc_hip_reserve_batch_tables decides from the caller's upper
// bound and sizes the tables for it; a batch that follows keeps that decision instead of deciding again from its own (smaller)
// count of unique bins -- two different answers meant an f table allocated at the b-only capacity: out of memory.
static int batch_refwd(psmc_hip_ctx *c, int64_t need_bins, bool *refwd)
{
	*refwd = false;
	if (c->ns != 64 || c->exact_refwd == 0) return 0;
	if (c->exact_refwd >= 1) { *refwd = true; return 0; }
	if (c->reserved_refwd >= 0) { *refwd = c->reserved_refwd != 0; return 0; }
	int64_t cap_tab = 0;
	int rc = batch_capacity(c, &cap_tab, false, need_bins);
	if (rc) return rc;
	*refwd = need_bins <= 0 || need_bins > cap_tab;
	return 0;
}

extern "C" int psmc_hip_reserve_batch_tables(psmc_hip_ctx *c, int64_t max_bins)
{
	if (!c) return PSMC_HIP_EINVAL;
	if (c->mode != PSMC_HIP_MODE_EXACT && c->ns <= 128) return PSMC_HIP_OK; // fast mode keeps one replicate's tables: nothing to reserve
	HIPCHK(c, hipSetDevice(c->device));
	if (c->n_seg < 1) return fail(c, PSMC_HIP_ESTATE, "reserve_batch_tables: no segments loaded");
	// A SECOND call while the reservation stands and more memory has become free (psmc_boot --main: the main run that shared the device is over
	// and its 31 GB of tables are gone): a second chunk of b table beside the first.  Growing the first would mean freeing and allocating 250+ GB
	// again -- this driver clears what it hands out, 24 ms per GB: 6.2 s + 1.7 s for the free (profiles/experiments/r06_realloc_probe.cpp) --
	// while 31 GB more cost half a second; an entry's table is a 64-bit offset from d_b, so a slot in the second chunk is just another offset
	// (batch_exact).  Only without the f table (the scale factors of such a batch live in a call-wide table of their own), once.
	if (c->reserved_cap > 0 && c->have_b && !c->d_f && c->tab_bins > 128 && c->batch_bins <= 0 && c->ns == 64 && (c->reserved_refwd == 1 || c->exact_refwd >= 1)) {
		const int64_t cap1 = c->tab_bins - 128;
		if (c->d_b2 || (max_bins > 0 && cap1 >= max_bins)) return PSMC_HIP_OK;
		size_t fr = 0, tot = 0;
		HIPCHK(c, hipMemGetInfo(&fr, &tot));
		const int64_t slack = 1 << 20; // (an entry that does not fit the rest of the first chunk starts the second: at most one entry's bins are left unused)
		int64_t extra = (int64_t)((double)fr * 0.9 / ((double)c->ns * 8.0)) - slack - 256;
		if (max_bins > 0) extra = std::min(extra, max_bins - cap1);
		if (extra < std::max<int64_t>(4096, cap1 / 64)) return PSMC_HIP_OK; // (not worth a launch)
		double *p2 = nullptr;
		if (hipMalloc((void **)&p2, (size_t)(extra + slack + 8) * c->ns * sizeof(double)) != hipSuccess) { (void)hipGetLastError(); return PSMC_HIP_OK; } // (as before, then)
		c->d_b2 = p2; c->b2_alloc = extra + slack + 8; c->b2_bins = extra + slack;
		c->reserved_cap = cap1 + extra;
		return PSMC_HIP_OK;
	}
	int64_t cap = 0;
	bool refwd = false;
	c->reserved_refwd = -1; c->reserved_cap = 0;
	if (c->d_b2) { (void)hipFree(c->d_b2); c->d_b2 = nullptr; c->b2_bins = c->b2_alloc = 0; } // a new reservation starts from one chunk
	int rc = batch_refwd(c, max_bins, &refwd);
	if (rc || (rc = batch_capacity(c, &cap, refwd, max_bins))) return rc;
	rc = ensure_tables(c, true, max_bins > 0 ? std::min(cap, max_bins) : cap, !refwd);
	// (several launches without the f table: the call-wide scale factors -- 15 GB for 100 replicates of a genome -- now as well, not inside the first EM iteration)
	if (rc == 0 && refwd && max_bins > cap && c->s_all_cap < (size_t)max_bins + 128 && (rc = dev_alloc(c, &c->d_s_all, (size_t)max_bins + 128)) == 0) c->s_all_cap = (size_t)max_bins + 128;
	if (rc == 0) {
		if (c->exact_refwd < 0) c->reserved_refwd = refwd ? 1 : 0; // (a fixed "exact_refwd" needs no memory)
		c->reserved_cap = cap; // the batches that follow plan their launches for THIS capacity: a second estimate from a different bound
		                       // (the call-wide scale-factor table is sized by the bound) would re-allocate 250 GB in the first EM iteration
	}
	return rc;
}

// Exact mode.  An ENTRY is one (replicate, unique segment) pair: one sequential sweep over that segment with that replicate's
// parameters, into a table slot of its own.  Round 4 cut the call into groups of consecutive replicates; every group then held
// a copy of the longest trunk and every launch lasted as long as that trunk (the sweeps are sequential: one wave per entry), with
// 30 % of the slot time idle beside the shorter ones.  Round 5 schedules ENTRIES: the entries of a replicate are sorted by
// length and cut into blocks of `align` (the sweeps of a work-group share one parameter set in LDS / registers), the blocks of
// all replicates are sorted by length, longest first, and dealt to launches that hold what the table memory and the wave slots
// (four entries per compute unit of this context's share of the device) allow.  The long trunks then share ONE launch and the
// others end when their -- shorter -- longest entry does (utils/splitfa.c: most trunks are exactly 500 k bins, the tails up to
// 750 k).  "batch_sort" = 0 keeps the replicate-major order (A/B).  Per-entry statistics stay on the host until the last launch;
// they are added per replicate in selection order exactly like psmc_hip_estep does: bit-identical to n_rep separate calls.
namespace {
struct Block { int rep; int first, n; int64_t bins; int32_t maxL; }; // entries ord[rep][first .. first + n) of a replicate
}
static int batch_exact(psmc_hip_ctx *c, int n_rep, const double *a, const double *e, const double *a0, const int32_t *sel_off,
                       const int32_t *sel_idx, double *A, double *sums, double *E, double *LL, psmc_hip_batch_done_fn done, void *user)
{
	const int n = c->n;
	const size_t S = (size_t)c->ns, PL = c->par_len;
	std::vector<RepSel> reps;
	int rc;
	if ((rc = batch_selections(c, n_rep, sel_off, sel_idx, reps))) return rc;
	int64_t cap = 0, all_bins = 0; // all_bins: what ONE launch of all entries would need
	for (const RepSel &R : reps) all_bins += R.bins;
	bool refwd = false;
	if ((rc = batch_refwd(c, all_bins, &refwd)) || (rc = batch_capacity(c, &cap, refwd, all_bins))) return rc;
	if (c->reserved_cap > 0 && c->batch_bins <= 0) cap = std::min(cap, c->reserved_cap);
	size_t n_entries_all = 0;
	for (const RepSel &R : reps) n_entries_all += R.work.size();
	// sweeps per work-group sharing one parameter set: four with 64 states (k_bwd_exact, k_expect_exact_rf2), up to four with
	// 65..128 (the matrix in LDS: fewer while the device has idle compute units), one beyond (estep_wide.hip)
	const int align = c->ns > 128 ? 1 : (c->ns == 128 ? (n_entries_all <= 256 ? 1 : (n_entries_all <= 512 ? 2 : 4)) : 4);
	c->last_batch_groups = 0;
	auto padded_len = [&](int32_t sg) { return ((int64_t)c->L[sg] + 63) & ~(int64_t)63; };
	// Entries per launch: the backward sweep has one wave slot per SIMD before waves share one, the recompute pass
	// (k_expect_exact_rf2) four entries per compute unit -- both 4 x CUs entries; a fifth entry on a unit waits for a whole
	// round (1088 entries on 1024 slots: measured 1.85 s against 0.95 s for 1024).  Only a call that needs several launches is cut.
	size_t ent_cap = SIZE_MAX;
	if (refwd && all_bins > cap) {
		int cus = c->cu_count;
		if (cus <= 0 && (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device) != hipSuccess || cus <= 0)) { (void)hipGetLastError(); cus = 256; }
		ent_cap = (size_t)cus * (c->exact_refwd == 1 ? 2 : 4);
	}
	// blocks of every replicate
	std::vector<std::vector<int32_t>> ord(n_rep); // work indices of a replicate, longest first ("batch_sort") or as selected
	std::vector<Block> blocks;
	for (int r = 0; r < n_rep; ++r) {
		const RepSel &R = reps[r];
		ord[r].resize(R.work.size());
		for (size_t i = 0; i < R.work.size(); ++i) ord[r][i] = (int32_t)i;
		if (c->batch_sort) std::stable_sort(ord[r].begin(), ord[r].end(), [&](int32_t x, int32_t y) { return c->L[R.work[x]] > c->L[R.work[y]]; });
		for (int f0 = 0; f0 < (int)ord[r].size(); f0 += align) {
			Block b = {r, f0, std::min(align, (int)ord[r].size() - f0), 0, 0};
			for (int i = 0; i < b.n; ++i) { const int32_t sg = R.work[ord[r][f0 + i]]; b.bins += padded_len(sg); b.maxL = std::max(b.maxL, c->L[sg]); }
			if (b.bins > cap) return fail(c, PSMC_HIP_ENOMEM, "estep_batch: the tables of one block of entries do not fit the device memory (batch_bins)");
			blocks.push_back(b);
		}
	}
	const std::vector<Block> blocks_made = blocks; // replicate-major, longest first inside a replicate
	if (c->batch_sort) std::stable_sort(blocks.begin(), blocks.end(), [](const Block &x, const Block &y) { return x.maxL > y.maxL; });
	// The work list of the whole call in launch order, every block padded to `align` entries: entry -> segment, replicate (= its
	// parameter block), offset of its tables inside ITS launch, offset of its scale factors in the call-wide s table.
	std::vector<int32_t> wseg, wpar; std::vector<int64_t> wtab, wtab_s; std::vector<int> l_first; // l_first: first entry of every launch
	std::vector<std::vector<int>> ent_of(n_rep); // [replicate][work index] -> entry
	for (int r = 0; r < n_rep; ++r) ent_of[r].assign(reps[r].work.size(), -1);
	// Blocks -> launches.  Greedy from the head of the list (longest first) until the table memory (cap) or the entry slots (ent_cap) are
	// used up.  That leaves launches that are full of bins with entry slots to spare (the long trunks), launches that are out of slots
	// with memory to spare (the short ones) -- and a last launch of a few dozen short entries that lasts as long as ITS longest sweep with
	// the device empty (40 entries: 0.36 s of a 6.75 s iteration, profiles/r05_boot_schedule.txt).  When the totals fit one launch
	// fewer, the head fill stops a little below cap and the shortest blocks of the list go into the spare slots instead.
	std::vector<std::vector<int>> lblocks;
	{
		std::vector<int64_t> lbins;
		size_t next = 0;
		auto head_fill = [&](int64_t cap_head, size_t max_launches) {
			lblocks.clear(); lbins.clear(); next = 0;
			while (next < blocks.size() && lblocks.size() < max_launches) {
				lblocks.emplace_back(); lbins.push_back(0);
				while (next < blocks.size()) {
					const Block &b = blocks[next];
					if (!lblocks.back().empty() && (lbins.back() + b.bins > cap_head || (lblocks.back().size() + 1) * (size_t)align > ent_cap)) break;
					lblocks.back().push_back((int)next); lbins.back() += b.bins; ++next;
				}
			}
		};
		head_fill(cap, SIZE_MAX);
		int64_t tot_bins = 0;
		for (const Block &b : blocks) tot_bins += b.bins;
		const size_t by_bins = (size_t)((tot_bins + cap - 1) / cap), by_slots = ent_cap == SIZE_MAX ? 1 : (blocks.size() * (size_t)align + ent_cap - 1) / ent_cap;
		const size_t fewest = std::max<size_t>(1, std::max(by_bins, by_slots));
		// head fill a little below cap into `max_launches` launches, then the rest of the list -- its shortest blocks -- each into the LAST launch with room
		auto tail_fill = [&](size_t max_launches) {
			for (double keep : {0.0025, 0.005, 0.01, 0.02, 0.04, 0.08}) {
				head_fill(cap - (int64_t)(keep * (double)cap), max_launches);
				bool ok = true;
				for (size_t t = next; t < blocks.size() && ok; ++t) {
					ok = false;
					for (size_t k = lblocks.size(); k-- > 0 && !ok;)
						if ((lblocks[k].size() + 1) * (size_t)align <= ent_cap && lbins[k] + blocks[t].bins <= cap) { lblocks[k].push_back((int)t); lbins[k] += blocks[t].bins; ok = true; }
				}
				if (ok) return true;
			}
			return false;
		};
		if (c->batch_sort && c->batch_tailfill && lblocks.size() > fewest) {
			const std::vector<std::vector<int>> greedy = lblocks;
			if (!tail_fill(fewest)) lblocks = greedy;
		}
		// Replicates that complete launch by launch ("batch_major").  With every block in order of length, each replicate's short trunks sit in
		// the last launch and no replicate is complete before it.  But utils/splitfa.c cuts the trunks to ONE length (500 k bins; only the
		// chromosomes' tails differ): when half of the blocks or more share their longest length Lc, a launch of blocks <= Lc lasts as long as
		// an Lc sweep whatever else is in it -- so those blocks keep the callers' replicate order (the longer ones still go first, by length),
		// a replicate is complete a launch or two after its first block, and the caller's M-steps (`done`) run under the launches that follow.
		// Not when this order would need a launch more than the order by length does -- with or without its tail fill (round 6: it used to
		// be skipped whenever the tail fill had saved a launch; with the second table chunk the 100-replicate job fits four launches
		// EITHER way, and by length all hundred M-steps -- 0.4-0.55 s on 12 threads -- waited for the last one).
		if (c->batch_sort && c->batch_major && done && lblocks.size() > 1) {
			std::vector<int32_t> ls;
			for (const Block &b : blocks) ls.push_back(b.maxL);
			std::sort(ls.begin(), ls.end());
			int32_t Lc = 0; size_t best = 0;
			for (size_t i = 0; i < ls.size();) { size_t j = i; while (j < ls.size() && ls[j] == ls[i]) ++j; if (j - i > best) { best = j - i; Lc = ls[i]; } i = j; }
			if (best * 2 >= blocks.size()) {
				const std::vector<Block> sorted = blocks; const std::vector<std::vector<int>> by_len = lblocks;
				blocks = blocks_made;
				std::stable_sort(blocks.begin(), blocks.end(), [Lc](const Block &x, const Block &y) { return std::max(x.maxL, Lc) > std::max(y.maxL, Lc); });
				const std::vector<Block> major = blocks;
				head_fill(cap, SIZE_MAX);
				// a launch too many: the same with the few shortest blocks of the call taken out of the replicates' order and put where there is
				// room (the tail fill above; the launch of the long trunks is out of memory with slots to spare, the others out of slots)
				size_t n_over = 0;
				for (size_t k = by_len.size(); k < lblocks.size(); ++k) n_over += lblocks[k].size();
				for (size_t mult = 1; lblocks.size() > by_len.size() && c->batch_tailfill && mult <= 8; mult *= 2) {
					const size_t K = std::min(major.size(), n_over * mult);
					std::vector<int> idx(major.size());
					for (size_t i = 0; i < idx.size(); ++i) idx[i] = (int)i;
					std::stable_sort(idx.begin(), idx.end(), [&](int x, int y) { return major[x].bins < major[y].bins; });
					std::vector<char> out(major.size(), 0);
					for (size_t i = 0; i < K; ++i) out[idx[i]] = 1;
					blocks.clear();
					for (size_t i = 0; i < major.size(); ++i) if (!out[i]) blocks.push_back(major[i]);
					for (size_t i = K; i-- > 0;) blocks.push_back(major[idx[i]]); // (the longest of the short ones first)
					if (!tail_fill(by_len.size())) { blocks = major; head_fill(cap, SIZE_MAX); }
				}
				if (lblocks.size() > by_len.size()) { blocks = sorted; lblocks = by_len; }
			}
		}
	}
	int64_t worst = 0; size_t worst_entries = 0;
	bool uses2 = false; // an entry's b table sits in the second chunk (psmc_hip_reserve_batch_tables, second call)
	{
		// the second chunk as an offset from d_b, in bins (rows of S doubles): its first row at a whole number of rows from d_b
		const bool two = refwd && c->d_b2 != nullptr && c->d_b != nullptr && c->tab_bins > 128;
		const int64_t row = (int64_t)S * 8, cap1 = two ? c->tab_bins - 128 : INT64_MAX;
		int64_t delta2 = 0, cap2 = 0;
		if (two) {
			const int64_t diff = (int64_t)((intptr_t)c->d_b2 - (intptr_t)c->d_b), skew = ((diff % row) + row) % row, shift = (row - skew) % row;
			delta2 = (diff + shift) / row; cap2 = c->b2_bins - (shift ? 1 : 0);
		}
		int64_t s_run = 0;
		for (const std::vector<int> &lb : lblocks) {
			int64_t run = 0, run2 = 0; const size_t e0 = wseg.size();
			l_first.push_back((int)e0);
			for (int bi : lb) {
				const Block &b = blocks[bi];
				for (int i = 0; i < align; ++i) {
					if (i < b.n) {
						const int32_t wi = ord[b.rep][b.first + i], sg = reps[b.rep].work[wi];
						const int64_t len = padded_len(sg);
						ent_of[b.rep][wi] = (int)wseg.size();
						const bool in2 = two && (run2 > 0 || run + len > cap1);
						if (in2 && run2 + len > cap2) return fail(c, PSMC_HIP_ESTATE, "estep_batch: launch does not fit the two table chunks");
						wseg.push_back(sg); wpar.push_back(b.rep); wtab.push_back(in2 ? delta2 + run2 : run); wtab_s.push_back(s_run + run + run2);
						if (in2) { run2 += len; uses2 = true; } else run += len;
					} else { wseg.push_back(-1); wpar.push_back(b.rep); wtab.push_back(0); wtab_s.push_back(0); }
				}
			}
			worst = std::max(worst, run); worst_entries = std::max(worst_entries, wseg.size() - e0);
			s_run += run + run2;
		}
		l_first.push_back((int)wseg.size());
	}
	const int n_launches = (int)l_first.size() - 1, n_all = (int)wseg.size();
	// Without the f table the forward pass writes scale factors only (8 bytes per bin), so with several launches it runs ONCE over the
	// entries of ALL of them -- thousands of waves, issue-bound, instead of one latency-bound launch of ~1000 waves each (100
	// replicates of a genome: 1.25 s against 4 x 0.9 s per EM iteration) -- into a call-wide s table (15 GB for 1.9 G bins).  Its
	// waves start in list order: longest first.
	const bool fwd_all = refwd && (n_launches > 1 || uses2); // (the second chunk holds b rows only: the scale factors of its entries need the call-wide table)
	{
		// tables: for the largest launch when the caller fixed "batch_bins"; else for everything that fits (or all entries at once), ONCE -- a
		// hipMalloc of 250 GB takes 4-6 s on this driver (it clears the memory: profiles/r04_boot_breakdown.txt), so the size must not depend
		// on this call's launches, and psmc_hip_reserve_batch_tables lets a caller pay for it while it loads
		if ((rc = ensure_tables(c, true, c->d_b2 ? std::min<int64_t>(worst, c->tab_bins - 128) : (c->batch_bins > 0 ? worst : std::max(worst, std::min(cap, all_bins))), !refwd))) return rc;
		if ((rc = ensure_seg_outputs(c, (int)worst_entries))) return rc;
		if (c->bw_cap < (size_t)n_all) {
			if ((rc = dev_alloc(c, &c->d_bw_seg, (size_t)n_all))) return rc;
			if ((rc = dev_alloc(c, &c->d_bw_par, (size_t)n_all))) return rc;
			if ((rc = dev_alloc(c, &c->d_bw_tab, (size_t)2 * n_all))) return rc; // tables | scale factors
			c->bw_cap = (size_t)n_all;
		}
		if (c->bpar_cap < (size_t)n_rep) { if ((rc = dev_alloc(c, &c->d_bpar, (size_t)n_rep * PL))) return rc; c->bpar_cap = (size_t)n_rep; }
		if (refwd && !c->d_cu_mask && (rc = dev_alloc(c, &c->d_cu_mask, (size_t)4096))) return rc;
		// (hmm_lk's products and their offsets for the LARGEST launch, here: growing them between two launches frees the old buffer, and hipFree
		// waits for every kernel on the device -- beside psmc_boot's main run that was its 3.3 s E-step: 2.66 s of "read-back" in launch 1)
		size_t lk_worst = 0;
		for (int g = 0; g < n_launches; ++g) {
			size_t t = 0;
			for (int i = l_first[g]; i < l_first[g + 1]; ++i) t += wseg[i] >= 0 ? (size_t)(c->L[wseg[i]] / LKP_DIV + LKP_MIN) : 1;
			lk_worst = std::max(lk_worst, t);
		}
		if (c->lkp_cap < lk_worst) { if ((rc = dev_alloc(c, &c->d_lkp, lk_worst))) return rc; c->lkp_cap = lk_worst; }
		if (c->lkoff_cap < worst_entries) { if ((rc = dev_alloc(c, &c->d_lkoff, worst_entries))) return rc; c->lkoff_cap = worst_entries; }
		if (fwd_all && c->s_all_cap < (size_t)all_bins + 128) { if ((rc = dev_alloc(c, &c->d_s_all, (size_t)all_bins + 128))) return rc; c->s_all_cap = (size_t)all_bins + 128; }
	}
	static const bool dbg_t = getenv("PSMC_HIP_DEBUG_TIMES") != nullptr;
	auto now = []() { return dbg_now(); };
	{
		std::vector<double> hp((size_t)n_rep * PL);
		for (int r = 0; r < n_rep; ++r) (void)fill_params(c, a + (size_t)r * n * n, e + (size_t)r * 2 * n, a0 + (size_t)r * n, hp.data() + (size_t)r * PL);
		HIPCHK(c, hipStreamSynchronize(c->stream));
		// (asynchronous copies on the context's own stream: a blocking hipMemcpy waits for every kernel on the DEVICE -- with psmc_boot
		// --main that is the main run's 1.6 s forward sweep; measured: 11.5 instead of 7.2 s per iteration while the main run was alive)
		HIPCHK(c, hipMemcpyAsync(c->d_bw_seg, wseg.data(), sizeof(int32_t) * n_all, hipMemcpyHostToDevice, c->stream));
		HIPCHK(c, hipMemcpyAsync(c->d_bw_par, wpar.data(), sizeof(int32_t) * n_all, hipMemcpyHostToDevice, c->stream));
		HIPCHK(c, hipMemcpyAsync(c->d_bw_tab, wtab.data(), sizeof(int64_t) * n_all, hipMemcpyHostToDevice, c->stream));
		HIPCHK(c, hipMemcpyAsync(c->d_bw_tab + n_all, wtab_s.data(), sizeof(int64_t) * n_all, hipMemcpyHostToDevice, c->stream));
		HIPCHK(c, hipMemcpyAsync(c->d_bpar, hp.data(), sizeof(double) * hp.size(), hipMemcpyHostToDevice, c->stream));
		HIPCHK(c, hipStreamSynchronize(c->stream)); // (the host vectors go out of scope)
	}
	c->tables_batch = true;
	double fwd_all_s = 0.0;
	if (fwd_all) {
		const double t0 = now();
		EstepLaunch p;
		fill_common(c, p, c->stream, c->d_bpar);
		p.d_work = c->d_bw_seg; p.n_work = n_all; p.d_work_par = c->d_bw_par; p.d_work_tab = c->d_bw_tab + n_all; p.par_stride = (int64_t)PL; p.work_align = align;
		p.exact_refwd = 1; p.exact_only = 1; p.d_s = c->d_s_all;
		if (launch_exact(p) != 0) return fail(c, PSMC_HIP_EDEVICE, "launch_exact (batch, forward pass of all entries)", hipGetLastError());
		if (dbg_t) { (void)hipStreamSynchronize(c->stream); fwd_all_s = now() - t0; }
	}
	// per-entry results of the whole call on the host (34 KB each): a replicate's entries may sit in different launches
	// (34 KB per entry at 64 states, 8 MB at 1024: a hundred replicates of sixty trunks would be 48 GB of pageable memory there -- ADVICE r5)
	if ((double)n_all * (double)(S * S) * 8.0 > 16e9) {
		c->err = "estep_batch: the per-entry statistics of this call (" + std::to_string(n_all) + " entries x " + std::to_string(S) + "^2 doubles) need more than 16 GB of host memory: send fewer replicates per call at this many states";
		return PSMC_HIP_ENOMEM;
	}
	c->h_segA.resize((size_t)n_all * S * S); c->h_segE.resize((size_t)n_all * 3 * S);
	std::vector<double> lk_all((size_t)n_all, 0.0);
	// the launch after which a replicate's statistics are complete
	std::vector<int> last_launch(n_rep, 0);
	{
		std::vector<int> launch_of((size_t)n_all, 0);
		for (int g = 0; g < n_launches; ++g) for (int i = l_first[g]; i < l_first[g + 1]; ++i) launch_of[i] = g;
		for (int r = 0; r < n_rep; ++r) for (int w : ent_of[r]) if (w >= 0) last_launch[r] = std::max(last_launch[r], launch_of[w]);
	}
	// hmm_add_expect in selection order per replicate (khmm.c:346-359), he_sum starting from zeros; LL += hmm_lk (em.c:48)
	std::vector<double> sA((size_t)n * n), sE((size_t)2 * n);
	double t_sums = 0.0;
	auto finalize = [&](int r) {
		const RepSel &R = reps[r];
		std::fill(sA.begin(), sA.end(), 0.0); std::fill(sE.begin(), sE.end(), 0.0);
		double ll = 0.0;
		for (size_t i = 0; i < R.sel2work.size(); ++i) {
			const int w = ent_of[r][R.sel2work[i]];
			const double *hA = &c->h_segA[(size_t)w * S * S], *hE = &c->h_segE[(size_t)w * 3 * S];
			ll += lk_all[(size_t)w];
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
	};
	std::vector<int32_t> finished;
	for (int g = 0; g < n_launches; ++g) {
		const double t_a = now();
		const int e0 = l_first[g], nw = l_first[g + 1] - e0;
		int64_t run = 0; int32_t longest = 0;
		for (int i = e0; i < e0 + nw; ++i) if (wseg[i] >= 0) { run = std::max(run, wtab[i] + padded_len(wseg[i])); longest = std::max(longest, c->L[wseg[i]]); }
		EstepLaunch p;
		fill_common(c, p, c->stream, c->d_bpar);
		p.d_work = c->d_bw_seg + e0; p.n_work = nw; p.d_work_par = c->d_bw_par + e0; p.d_work_tab = c->d_bw_tab + e0; p.par_stride = (int64_t)PL; p.work_align = align;
		p.exact_refwd = refwd ? (c->exact_refwd == 1 ? 1 : 2) : 0; // 2: two entries per work-group (k_expect_exact_rf2), the default
		p.d_cu_mask = c->d_cu_mask;
		if (fwd_all) { p.exact_only = 2; p.d_work_tab_s = c->d_bw_tab + n_all + e0; p.d_s = c->d_s_all; }
		p.d_segA = c->d_segA; p.d_segE = c->d_segE; p.d_segA0 = c->d_segA0; p.d_chk = c->d_chk;
		const double t_b = now();
		if (launch_exact(p) != 0) return fail(c, PSMC_HIP_EDEVICE, "launch_exact (batch)", hipGetLastError());
		if (dbg_t) (void)hipStreamSynchronize(c->stream);
		const double t_c = now();
		// hmm_lk of the launch's entries (khmm.c:245-260): the running products on the device (k_lk_products: the same multiplications
		// in the same order), the platform's log() of the few numbers that are logged here -- instead of reading 8 bytes per bin back
		std::vector<int64_t> lkoff((size_t)nw);
		int64_t lk_tot = 0;
		for (int i = 0; i < nw; ++i) { lkoff[i] = lk_tot; lk_tot += wseg[e0 + i] >= 0 ? c->L[wseg[e0 + i]] / LKP_DIV + LKP_MIN : 1; }
		HIPCHK(c, hipMemcpyAsync(c->d_lkoff, lkoff.data(), sizeof(int64_t) * nw, hipMemcpyHostToDevice, c->stream));
		if (launch_lk_products(c->stream, p, fwd_all ? c->d_s_all : c->d_s, c->d_lkoff, c->d_lkp) != 0) return fail(c, PSMC_HIP_EDEVICE, "launch_lk_products", hipGetLastError());
		c->h_lkp.resize((size_t)lk_tot);
		HIPCHK(c, hipMemcpyAsync(c->h_segA.data() + (size_t)e0 * S * S, c->d_segA, sizeof(double) * nw * S * S, hipMemcpyDeviceToHost, c->stream));
		HIPCHK(c, hipMemcpyAsync(c->h_segE.data() + (size_t)e0 * 3 * S, c->d_segE, sizeof(double) * nw * 3 * S, hipMemcpyDeviceToHost, c->stream));
		HIPCHK(c, hipMemcpyAsync(c->h_lkp.data(), c->d_lkp, sizeof(double) * (size_t)lk_tot, hipMemcpyDeviceToHost, c->stream));
		HIPCHK(c, hipStreamSynchronize(c->stream)); // (lkoff stays alive until here)
		const double t_d = now();
		collect_timing(c);
		int n_over = 0;
		for (int i = 0; i < nw; ++i) {
			if (wseg[e0 + i] < 0) continue;
			const int32_t sg = wseg[e0 + i];
			const double *q = &c->h_lkp[(size_t)lkoff[i]];
			const int cnt = (int)q[0], cap_i = c->L[sg] / LKP_DIV + LKP_MIN;
			if (cnt >= 1 && cnt < cap_i) {
				double sum = 0.0;
				for (int j = 1; j <= cnt; ++j) sum += log(q[j]);
				lk_all[(size_t)e0 + i] = sum;
			} else { // more logged products than the buffer holds (scale factors around 0.03 throughout): this entry's scale factors, the host's product
				++n_over;
				c->h_s.resize((size_t)c->L[sg]);
				HIPCHK(c, hipMemcpyAsync(c->h_s.data(), (fwd_all ? c->d_s_all + wtab_s[e0 + i] : c->d_s + wtab[e0 + i]), sizeof(double) * (size_t)c->L[sg], hipMemcpyDeviceToHost, c->stream));
				HIPCHK(c, hipStreamSynchronize(c->stream));
				lk_all[(size_t)e0 + i] = host_lk(c->h_s.data(), c->L[sg]);
			}
		}
		if (dbg_t && n_over) fprintf(stderr, "[psmc_hip] batch launch %d: %d entries logged more products than their buffer holds (host hmm_lk)\n", g, n_over);
		if (dbg_t) fprintf(stderr, "[psmc_hip] batch launch %d: %d entries (longest %d bins), %.1f M table bins | prepare %.3f s, kernels %.3f (fwd %.0f bwd %.0f expect %.0f ms%s), read-back %.3f, hmm_lk %.3f\n",
		                   g, nw, (int)longest, run / 1e6, t_b - t_a, t_c - t_b, c->last_ms[1], c->last_ms[2], c->last_ms[3],
		                   fwd_all ? (g == 0 ? (std::string("; forward pass of all entries ") + std::to_string(fwd_all_s) + " s").c_str() : "; fwd: see launch 0") : "", t_d - t_c, now() - t_d);
		++c->last_batch_groups;
		// the replicates this launch completed: their sums now, and the caller hears of them before the next launch starts
		const double t_f = now();
		finished.clear();
		for (int r = 0; r < n_rep; ++r) if (last_launch[r] == g) { finalize(r); finished.push_back(r); }
		t_sums += now() - t_f;
		if (done && !finished.empty()) done(user, (int)finished.size(), finished.data());
	}
	if (dbg_t) {
		std::string per;
		for (int g = 0; g < n_launches; ++g) per += (g ? " " : "") + std::to_string(std::count(last_launch.begin(), last_launch.end(), g));
		fprintf(stderr, "[psmc_hip] batch: %d replicates, %d entries in %d launches (replicates complete after launch 0..: %s), host sums %.3f s\n", n_rep, n_all, n_launches, per.c_str(), t_sums);
	}
	return PSMC_HIP_OK;
}

// Fast mode: a single replicate already fills the device, so the replicates run one after the other -- but each
// keeps ITS OWN tile plan (tiling of its selection, learned glued runs, transfer-matrix lists) in a child context,
// so nothing is re-planned or re-learned from one EM iteration to the next.  Children share the parent's tables.
static psmc_hip_ctx *batch_child(psmc_hip_ctx *c, int r)
{
	while ((int)c->kids.size() <= r) { // (positions below r that no call has named yet get their context too: unselected, no tables of their own)
		psmc_hip_ctx *k = new (std::nothrow) psmc_hip_ctx();
		if (!k) return nullptr;
		k->n = c->n; k->ns = c->ns; k->device = c->device; k->mode = c->mode; k->parent = c;
		k->chunk = c->chunk > 0 ? c->chunk : c->share_T; // (share_T = 0 without "share_learn": its own tiling)
		k->warmup = c->warmup; k->max_rounds = c->max_rounds; k->rep_impl = c->rep_impl;
		k->n_sub = c->n_sub; k->target_waves = c->target_waves; k->overlap = c->overlap; k->warm_tol = c->warm_tol; k->struct_opt = c->struct_opt;
		k->struct_tiles_set = c->struct_tiles_set; k->struct_tiles = c->struct_tiles;
		k->two_phase = c->two_phase; k->kc_div = c->kc_div; k->kc_min = c->kc_min; k->ckpt = c->ckpt; k->fuse = c->fuse;
		k->learn = c->learn; k->group_cap = c->group_cap; k->warm_shift = c->warm_shift; k->kc_sub = c->kc_sub; k->kcol_prio = c->kcol_prio;
		k->fuse128 = c->fuse128; k->coarse = c->coarse; k->gate = c->gate; k->lanes8 = c->lanes8; k->gap_tiles = c->gap_tiles; k->exact_refwd = c->exact_refwd;
		k->merge = c->merge; k->adapt = c->adapt; k->prev_start = c->prev_start;
		k->merge1 = c->merge1; k->merge_order = c->merge_order; k->runs_late = c->runs_late; k->warm_shift_set = c->warm_shift_set; k->kc_sub_set = c->kc_sub_set;
		k->stream = c->stream; k->stream2 = c->stream2; k->stream3 = c->stream3; k->stream4 = c->stream4; k->stream5 = c->stream5;
		for (int i = 0; i < 14; ++i) k->evx[i] = c->evx[i];
		for (int i = 0; i < 10; ++i) k->ev[i] = c->ev[i];
		k->h_par = c->h_par; k->d_par = c->d_par; k->par_len = c->par_len;
		// same segments, same offsets, the parent's copy of the observations
		k->d_obs = c->d_obs; k->obs_borrowed = true; k->off = c->off; k->total = c->total;
		if (set_segments_common(k, c->n_seg, c->L.data()) != 0) { c->err = k->err; psmc_hip_destroy(k); return nullptr; } // never keep a half-built child
		c->kids.push_back(k);
	}
	return c->kids[r];
}

// A replicate whose fast E-step ends with PSMC_HIP_ECONVERGE (the verify / repair rounds of its tiles ran out) used to end the whole batch --
// a hundred-replicate job of psmc_boot -- where `psmc` repeats that one E-step with the exact kernels (host/hipbe.c; khmm.c:145-324 has no
// such failure mode).  The batch does the same: an exact twin of the context over the same observations (borrowed, already in HBM), made on
// first need, runs THAT replicate's selection; a line on stderr says so; the batch goes on in fast mode.  
static int batch_exact_once(psmc_hip_ctx *c, int r, const int32_t *idx, int n_sel, const double *a, const double *e, const double *a0,
                            double *A, double *sums, double *E, double *LL, const char *why)
{
	const int n = c->n;
	fprintf(stderr, "[psmc_hip] replicate %d: fast E-step did not converge (%s); repeating this E-step with the exact kernels\n", c->batch_first + r, why);
	int rc = 0;
	if (!c->x_twin) {
		if ((rc = psmc_hip_create(&c->x_twin, n, c->device, PSMC_HIP_MODE_EXACT))) return fail(c, rc, "estep_batch: cannot create the exact twin context");
		if ((rc = psmc_hip_load_segments_device(c->x_twin, c->n_seg, c->d_obs, c->off.data(), c->L.data()))) {
			c->err = "exact twin: " + c->x_twin->err; psmc_hip_destroy(c->x_twin); c->x_twin = nullptr; return rc;
		}
	}
	std::vector<double> Af;
	double *Ao = A;
	if (!Ao) { Af.resize((size_t)n * n); Ao = Af.data(); }
	if ((rc = psmc_hip_select(c->x_twin, n_sel, idx)) || (rc = psmc_hip_estep(c->x_twin, a, e, a0, Ao, E, nullptr, LL, nullptr))) { c->err = "exact twin: " + c->x_twin->err; return rc; }
	if (sums) { // SL | SU | DG | CL | CU of the exact counts (what psmc_hip_estep_factored returns)
		memset(sums, 0, sizeof(double) * 5 * (size_t)n);
		for (int k = 0; k < n; ++k)
			for (int l = 0; l < n; ++l) {
				const double v = Ao[(size_t)k * n + l];
				if (l < k) { sums[k] += v; sums[3 * n + l] += v; } else if (l > k) { sums[n + k] += v; sums[4 * n + l] += v; } else sums[2 * n + k] = v;
			}
	}
	++c->n_exact_fallbacks;
	return PSMC_HIP_OK;
}

static int batch_fast(psmc_hip_ctx *c, int n_rep, const double *a, const double *e, const double *a0, const int32_t *sel_off,
                      const int32_t *sel_idx, double *A, double *sums, double *E, double *LL, psmc_hip_batch_done_fn done, void *user)
{
	const int n = c->n;
	if (c->share_T == 0 && c->share_learn) {
		// One tile length for every replicate (the largest one's: the others then have fewer tiles than a round holds), so that a tile
		// is (segment, index) in all of them and what one replicate learns about the slow regions serves the next (psmc_hip_ctx.h)
		int T = c->chunk;
		if (T <= 0) {
			std::vector<RepSel> reps;
			int rc0 = batch_selections(c, n_rep, sel_off, sel_idx, reps);
			if (rc0) return rc0;
			int64_t most = 0; size_t nw = 1;
			for (const RepSel &R : reps) {
				int64_t bins = 0;
				for (int32_t sg : R.work) bins += c->L[sg];
				if (bins > most) { most = bins; nw = R.work.size(); }
			}
			T = auto_tile_len(c, most, nw, true);
		}
		c->share_T = T;
		c->sh_glue_f.assign(c->n_seg, {}); c->sh_glue_b.assign(c->n_seg, {}); c->sh_wf.assign(c->n_seg, {}); c->sh_wb.assign(c->n_seg, {});
		for (int sg = 0; sg < c->n_seg; ++sg) {
			const size_t nt = ((size_t)c->L[sg] + T - 1) / T;
			c->sh_glue_f[sg].assign(nt, 0); c->sh_glue_b[sg].assign(nt, 0); c->sh_wf[sg].assign(nt, c->warmup); c->sh_wb[sg].assign(nt, c->warmup);
		}
	}
	static const bool dbg_t = getenv("PSMC_HIP_DEBUG_TIMES") != nullptr;
	const double t_call = dbg_now();
	double t_est = 0.0;
	for (int r = 0; r < n_rep; ++r) {
		const double t_k = dbg_now();
		psmc_hip_ctx *k = batch_child(c, c->batch_first + r);
		c->dbg_acc[0] += dbg_now() - t_k;
		if (!k) return fail(c, PSMC_HIP_ENOMEM, "estep_batch: cannot create the replicate context");
		const int n_sel = sel_off[r + 1] - sel_off[r];
		if (n_sel < 1) return fail(c, PSMC_HIP_EINVAL, "estep_batch: empty selection");
		const int32_t *idx = sel_idx + sel_off[r];
		int rc = 0;
		{
			const double t_s = dbg_now();
			if ((int)k->sel.size() != n_sel || memcmp(k->sel.data(), idx, sizeof(int32_t) * n_sel) != 0) rc = psmc_hip_select(k, n_sel, idx);
			c->dbg_acc[1] += dbg_now() - t_s;
		}
		const double t_e = dbg_now();
		const double *ar = a + (size_t)r * n * n, *er = e + (size_t)r * 2 * n, *a0r = a0 + (size_t)r * n;
		if (rc == 0) {
			double *Ar = A ? A + (size_t)r * n * n : nullptr, *Sr = sums ? sums + (size_t)r * 5 * n : nullptr, *Er = E ? E + (size_t)r * 2 * n : nullptr, *Lr = LL ? LL + r : nullptr;
			if (A) rc = psmc_hip_estep(k, ar, er, a0r, Ar, Er, nullptr, Lr, nullptr);
			if (rc == 0 && sums) rc = psmc_hip_estep_factored(k, ar, er, a0r, Sr, Er, Lr);
			if (rc == PSMC_HIP_ECONVERGE) rc = batch_exact_once(c, r, idx, n_sel, ar, er, a0r, Ar, Sr, Er, Lr, k->err.c_str());
		}
		t_est += dbg_now() - t_e;
		if (rc) { c->err = "replicate " + std::to_string(r) + ": " + k->err; return rc; }
		if (done) { const int32_t rr = r; done(user, 1, &rr); } // its statistics are final: the caller's M-step can run under the next replicate's E-step
	}
	if (dbg_t) {
		const double *q = c->dbg_acc;
		fprintf(stderr, "[psmc_hip] fast batch on device %d: %d replicates %.3f s | contexts %.3f, select %.3f, E-steps %.3f (plan %.3f, items %.3f, launches %.3f, rest %.3f); "
		                "%.0f repair rounds, %.0f repaired tiles\n", c->device, n_rep, dbg_now() - t_call, q[0], q[1], t_est, q[2], q[3], q[4], t_est - q[2] - q[3] - q[4], q[6], q[7]);
		for (double &v : c->dbg_acc) v = 0.0;
	}
	c->tables_batch = true;
	c->last_batch_groups = n_rep;
	return PSMC_HIP_OK;
}

extern "C" int psmc_hip_estep_batch(psmc_hip_ctx *c, int n_rep, const double *a, const double *e, const double *a0,
                                    const int32_t *sel_off, const int32_t *sel_idx, double *A, double *sums, double *E, double *LL)
{
	return psmc_hip_estep_batch_cb(c, n_rep, a, e, a0, sel_off, sel_idx, A, sums, E, LL, nullptr, nullptr);
}

extern "C" int psmc_hip_estep_batch_cb(psmc_hip_ctx *c, int n_rep, const double *a, const double *e, const double *a0,
                                       const int32_t *sel_off, const int32_t *sel_idx, double *A, double *sums, double *E, double *LL,
                                       psmc_hip_batch_done_fn done, void *user)
{
	if (!c || n_rep < 1 || !a || !e || !a0 || !sel_off || !sel_idx || (!A && !sums)) return fail(c, PSMC_HIP_EINVAL, "estep_batch: bad argument");
	if (c->parent) return fail(c, PSMC_HIP_EINVAL, "estep_batch: not on a replicate context");
	if (c->n_seg < 1) return fail(c, PSMC_HIP_ESTATE, "estep_batch: no segments loaded");
	HIPCHK(c, hipSetDevice(c->device));
	if (c->mode == PSMC_HIP_MODE_EXACT || c->ns > 128) return batch_exact(c, n_rep, a, e, a0, sel_off, sel_idx, A, sums, E, LL, done, user); // (beyond 128 states: the wide exact kernels whatever the mode)
	return batch_fast(c, n_rep, a, e, a0, sel_off, sel_idx, A, sums, E, LL, done, user);
}

extern "C" int psmc_hip_batch_info(psmc_hip_ctx *c, int out[2])
{
	if (!c || !out) return PSMC_HIP_EINVAL;
	out[0] = c->last_batch_groups; out[1] = (int)c->kids.size();
	return PSMC_HIP_OK;
}
