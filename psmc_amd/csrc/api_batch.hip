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

// Table bins an exact batch may use: "batch_bins", or 0.9 of the device memory that is free or already in this context's tables.
// (Round 3 took 0.9 of the free memory PLUS all of what the context held: the second call saw a larger capacity than the first,
// re-planned its groups and re-allocated 250 GB of tables -- 8 s; profiles/r04_boot_breakdown.txt.)
static int batch_capacity(psmc_hip_ctx *c, int64_t *cap, bool refwd)
{
	*cap = c->batch_bins;
	if (*cap > 0) return 0;
	size_t fr = 0, tot = 0;
	HIPCHK(c, hipMemGetInfo(&fr, &tot));
	const double S = (double)c->ns, per_bin = S * 8.0 * (refwd ? 1.0 : 2.0) + 8.0;
	const double held = (double)c->tab_bins * (S * 8.0 * ((c->have_b ? 1.0 : 0.0) + (c->d_f ? 1.0 : 0.0)) + 8.0 + (c->d_sb ? 8.0 : 0.0));
	// (the call-wide scale-factor table of a batch without the f table stays allocated between calls: count it as free, or the second
	// call would see a smaller capacity than the first and cut other groups)
	*cap = (int64_t)(((double)fr + held + (double)c->s_all_cap * 8.0) * 0.9 / per_bin) - 256;
	if (refwd) *cap -= *cap / 64; // ... and leave room for it: 8 of every 520 bytes
	if (*cap < 1) return fail(c, PSMC_HIP_ENOMEM, "estep_batch: no device memory left for tables");
	return 0;
}

// Does this batch run without the f table (k_expect_exact_rf: 64 states)?  "exact_refwd" 1 / 0: yes / no.  Auto: only when the f and b
// tables of all its replicates (need_bins; <= 0: unknown) would NOT fit one launch group -- the recompute pass runs at the forward
// sweep's latency (0.83 instead of 0.45 us per bin of the longest segment), which pays when it halves the number of groups (100
// replicates of a genome: 11.9 against 13.2 s per EM iteration) and costs when one group would have done (16 replicates: 2.4 against 1.9 s).
static int batch_refwd(psmc_hip_ctx *c, int64_t need_bins, bool *refwd)
{
	*refwd = false;
	if (c->ns != 64 || c->exact_refwd == 0) return 0;
	if (c->exact_refwd >= 1) { *refwd = true; return 0; }
	int64_t cap_tab = 0;
	int rc = batch_capacity(c, &cap_tab, false);
	if (rc) return rc;
	*refwd = need_bins <= 0 || need_bins > cap_tab;
	return 0;
}

extern "C" int psmc_hip_reserve_batch_tables(psmc_hip_ctx *c, int64_t max_bins)
{
	if (!c) return PSMC_HIP_EINVAL;
	if (c->mode != PSMC_HIP_MODE_EXACT) return PSMC_HIP_OK; // fast mode keeps one replicate's tables: nothing to reserve
	HIPCHK(c, hipSetDevice(c->device));
	if (c->n_seg < 1) return fail(c, PSMC_HIP_ESTATE, "reserve_batch_tables: no segments loaded");
	int64_t cap = 0;
	bool refwd = false;
	int rc = batch_refwd(c, max_bins, &refwd);
	if (rc || (rc = batch_capacity(c, &cap, refwd))) return rc;
	return ensure_tables(c, true, max_bins > 0 ? std::min(cap, max_bins) : cap, !refwd);
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
	int64_t cap = 0, all_bins = 0; // all_bins: what ONE group of all replicates would need
	for (const RepSel &R : reps) all_bins += R.bins;
	bool refwd = false;
	if ((rc = batch_refwd(c, all_bins, &refwd)) || (rc = batch_capacity(c, &cap, refwd))) return rc;
	size_t n_entries_all = 0;
	for (const RepSel &R : reps) n_entries_all += R.work.size();
	const int align = c->ns == 128 ? (n_entries_all <= 256 ? 1 : (n_entries_all <= 512 ? 2 : 4)) : 4; // sweeps per block sharing one parameter set in LDS
	c->last_batch_groups = 0;
	// Entries per group.  An entry is one sequential sweep over its segment, and the replicates of a bootstrap are made of equal trunks
	// (utils/splitfa.c), so a launch runs in ROUNDS: the backward sweep has one wave slot per SIMD before waves share one, the recompute
	// pass (k_expect_exact_rf2) four entries per compute unit -- both 4 x CUs entries.  What memory allows beyond a whole number of
	// rounds runs as a nearly empty extra round (1088 entries on 1024 slots: two rounds, measured 1.85 s against 0.95 s for 1024), so
	// a group is cut at the last full round it can hold.
	size_t ent_cap = SIZE_MAX;
	if (refwd && c->ns == 64 && n_entries_all > 0 && all_bins > cap) {
		int cus = 0;
		if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device) != hipSuccess || cus <= 0) { (void)hipGetLastError(); cus = 256; }
		const size_t slots = (size_t)cus * (c->exact_refwd == 1 ? 2 : 4);
		const size_t fit = (size_t)((double)cap / ((double)all_bins / (double)n_entries_all)); // entries the table memory holds
		if (fit >= slots) ent_cap = fit / slots * slots;
	}
	// The work list of the whole call, replicate-major, every replicate padded to `align` entries: entry -> segment, replicate (= its
	// parameter block), offset of its tables inside ITS launch group, offset of its scale factors in the call-wide s table.  Groups =
	// as many consecutive replicates as fit the table memory: slices [g_first[g], g_first[g+1]) of that list.
	std::vector<int32_t> wseg, wpar; std::vector<int64_t> wtab, wtab_s; std::vector<int> first(n_rep + 1, 0), g_rep;
	std::vector<double> lk;
	int64_t worst = 0; size_t worst_entries = 0;
	{
		int64_t s_run = 0;
		for (int r0 = 0; r0 < n_rep;) {
			if (reps[r0].bins > cap) return fail(c, PSMC_HIP_ENOMEM, "estep_batch: the tables of one replicate do not fit the device memory (batch_bins)");
			int r1 = r0; int64_t bins = 0; size_t ents = 0;
			auto padded = [&](int r) { return (reps[r].work.size() + align - 1) / align * align; };
			while (r1 < n_rep && bins + reps[r1].bins <= cap && (r1 == r0 || ents + padded(r1) <= ent_cap)) { bins += reps[r1].bins; ents += padded(r1); ++r1; }
			g_rep.push_back(r0);
			int64_t run = 0; const size_t e0 = wseg.size();
			for (int r = r0; r < r1; ++r) {
				first[r] = (int)wseg.size();
				for (int32_t sg : reps[r].work) { wseg.push_back(sg); wpar.push_back(r); wtab.push_back(run); wtab_s.push_back(s_run + run); run += ((int64_t)c->L[sg] + 63) & ~(int64_t)63; }
				while (wseg.size() % align) { wseg.push_back(-1); wpar.push_back(r); wtab.push_back(0); wtab_s.push_back(0); }
			}
			s_run += run;
			worst = std::max(worst, run); worst_entries = std::max(worst_entries, wseg.size() - e0);
			r0 = r1;
		}
		g_rep.push_back(n_rep); first[n_rep] = (int)wseg.size();
	}
	const int n_groups = (int)g_rep.size() - 1, n_all = (int)wseg.size();
	// Without the f table the forward pass writes scale factors only (8 bytes per bin), so with several groups it runs ONCE over the
	// replicates of ALL of them -- thousands of waves, issue-bound, instead of one latency-bound launch of ~1000 waves per group
	// (100 replicates of a genome: 1.8 s against 4 x 0.9 s per EM iteration) -- into a call-wide s table (15 GB for 1.9 G bins).
	const bool fwd_all = refwd && n_groups > 1;
	{
		// tables: for the largest group when the caller fixed "batch_bins"; else for everything that fits (or all replicates at once), ONCE -- a
		// hipMalloc of 250 GB takes 4-6 s on this driver (it clears the memory: scripts/r04/malloc_probe.py), so the size must not depend
		// on this call's groups, and psmc_hip_reserve_batch_tables lets a caller pay for it while it loads
		if ((rc = ensure_tables(c, true, c->batch_bins > 0 ? worst : std::max(worst, std::min(cap, all_bins)), !refwd))) return rc;
		if ((rc = ensure_seg_outputs(c, (int)worst_entries))) return rc;
		if (c->bw_cap < (size_t)n_all) {
			if ((rc = dev_alloc(c, &c->d_bw_seg, (size_t)n_all))) return rc;
			if ((rc = dev_alloc(c, &c->d_bw_par, (size_t)n_all))) return rc;
			if ((rc = dev_alloc(c, &c->d_bw_tab, (size_t)2 * n_all))) return rc; // tables | scale factors
			c->bw_cap = (size_t)n_all;
		}
		if (c->bpar_cap < (size_t)n_rep) { if ((rc = dev_alloc(c, &c->d_bpar, (size_t)n_rep * PL))) return rc; c->bpar_cap = (size_t)n_rep; }
		if (refwd && !c->d_cu_mask && (rc = dev_alloc(c, &c->d_cu_mask, (size_t)4096))) return rc;
		if (fwd_all && c->s_all_cap < (size_t)all_bins + 128) { if ((rc = dev_alloc(c, &c->d_s_all, (size_t)all_bins + 128))) return rc; c->s_all_cap = (size_t)all_bins + 128; }
	}
	static const bool dbg_t = getenv("PSMC_HIP_DEBUG_TIMES") != nullptr;
	auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
	{
		std::vector<double> hp((size_t)n_rep * PL);
		for (int r = 0; r < n_rep; ++r) (void)fill_params(c, a + (size_t)r * n * n, e + (size_t)r * 2 * n, a0 + (size_t)r * n, hp.data() + (size_t)r * PL);
		HIPCHK(c, hipStreamSynchronize(c->stream));
		HIPCHK(c, hipMemcpy(c->d_bw_seg, wseg.data(), sizeof(int32_t) * n_all, hipMemcpyHostToDevice));
		HIPCHK(c, hipMemcpy(c->d_bw_par, wpar.data(), sizeof(int32_t) * n_all, hipMemcpyHostToDevice));
		HIPCHK(c, hipMemcpy(c->d_bw_tab, wtab.data(), sizeof(int64_t) * n_all, hipMemcpyHostToDevice));
		HIPCHK(c, hipMemcpy(c->d_bw_tab + n_all, wtab_s.data(), sizeof(int64_t) * n_all, hipMemcpyHostToDevice));
		HIPCHK(c, hipMemcpy(c->d_bpar, hp.data(), sizeof(double) * hp.size(), hipMemcpyHostToDevice));
	}
	c->tables_batch = true;
	double fwd_all_s = 0.0;
	if (fwd_all) {
		const double t0 = now();
		EstepLaunch p;
		fill_common(c, p, c->stream, c->d_bpar);
		p.d_work = c->d_bw_seg; p.n_work = n_all; p.d_work_par = c->d_bw_par; p.d_work_tab = c->d_bw_tab + n_all; p.par_stride = (int64_t)PL; p.work_align = align;
		p.exact_refwd = 1; p.exact_only = 1; p.d_s = c->d_s_all;
		if (launch_exact(p) != 0) return fail(c, PSMC_HIP_EDEVICE, "launch_exact (batch, forward pass of all replicates)", hipGetLastError());
		if (dbg_t) { (void)hipStreamSynchronize(c->stream); fwd_all_s = now() - t0; }
	}
	for (int g = 0; g < n_groups; ++g) {
		const double t_a = now();
		const int r0 = g_rep[g], r1 = g_rep[g + 1], ng = r1 - r0, e0 = first[r0], nw = first[r1] - e0;
		const int64_t s_base = fwd_all ? wtab_s[e0] : 0;
		int64_t run = 0;
		for (int i = e0; i < e0 + nw; ++i) if (wseg[i] >= 0) run = std::max(run, wtab[i] + (((int64_t)c->L[wseg[i]] + 63) & ~(int64_t)63));
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
		c->h_segA.resize((size_t)nw * S * S); c->h_segE.resize((size_t)nw * 3 * S); c->h_s.resize((size_t)run);
		HIPCHK(c, hipMemcpyAsync(c->h_segA.data(), c->d_segA, sizeof(double) * nw * S * S, hipMemcpyDeviceToHost, c->stream));
		HIPCHK(c, hipMemcpyAsync(c->h_segE.data(), c->d_segE, sizeof(double) * nw * 3 * S, hipMemcpyDeviceToHost, c->stream));
		HIPCHK(c, hipMemcpyAsync(c->h_s.data(), (fwd_all ? c->d_s_all : c->d_s) + s_base, sizeof(double) * (size_t)run, hipMemcpyDeviceToHost, c->stream));
		HIPCHK(c, hipStreamSynchronize(c->stream));
		const double t_d = now();
		collect_timing(c);
		// hmm_add_expect in selection order per replicate (khmm.c:346-359), he_sum starting from zeros; LL += hmm_lk (em.c:48)
		std::vector<double> sA((size_t)n * n), sE((size_t)2 * n);
		// hmm_lk of every (replicate, segment) entry: a running product over all of its bins with the platform log() -- 0.7 ns per bin,
		// 0.36 s per group of 28 replicates on one core; the entries are independent, so host threads share them (each value is
		// computed by one thread exactly as before: bit-identical)
		std::vector<double> lk_all((size_t)nw, 0.0);
		{
			const unsigned nt = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
			std::atomic<size_t> next(0);
			auto work = [&]() {
				for (size_t i = next.fetch_add(1); i < (size_t)nw; i = next.fetch_add(1))
					if (wseg[e0 + i] >= 0) lk_all[i] = host_lk(&c->h_s[(size_t)wtab[e0 + i]], c->L[wseg[e0 + i]]);
			};
			std::vector<std::thread> th;
			for (unsigned t = 1; t < nt; ++t) th.emplace_back(work);
			work();
			for (std::thread &t : th) t.join();
		}
		for (int r = r0; r < r1; ++r) {
			const RepSel &R = reps[r];
			const int f0 = first[r] - e0; // the replicate's first entry inside the group's launch
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
		if (dbg_t) fprintf(stderr, "[psmc_hip] batch group %d: %d replicates, %d entries, %.1f M table bins | prepare %.3f s, kernels %.3f (fwd %.0f bwd %.0f expect %.0f ms%s), read-back %.3f, host sums %.3f\n",
		                   c->last_batch_groups, ng, nw, run / 1e6, t_b - t_a, t_c - t_b, c->last_ms[1], c->last_ms[2], c->last_ms[3],
		                   fwd_all ? (g == 0 ? (std::string("; forward pass of all replicates ") + std::to_string(fwd_all_s) + " s").c_str() : "; fwd: see group 0") : "", t_d - t_c, now() - t_d);
		++c->last_batch_groups;
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
		k->chunk = c->chunk > 0 ? c->chunk : c->share_T; // (share_T = 0 without "share_learn": its own tiling)
		k->warmup = c->warmup; k->max_rounds = c->max_rounds; k->rep_impl = c->rep_impl; k->expect_impl = c->expect_impl;
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
