// api.hip -- C-ABI of libpsmc_hip.so (see include/psmc_hip.h): context, options, segment upload, parameter staging, the
// tables, the exact mode with its host-side ordered reductions, the table readers and the decoding entry points.  The fast
// mode's planner and launcher are in api_fast.hip, the bootstrap batch in api_batch.hip, the diagnostics in api_probes.hip;
// psmc_hip_ctx.h holds the context they share.
#include "psmc_hip_ctx.h"

// The fast E-step keeps four streams busy (forward chain, backward chain, counts, walks) beside the
// caller's.  HIP hands a process 4 hardware queues by default and streams beyond that share one -- a
// kernel then waits for an unrelated one.  Ask for 8 before the runtime starts (no effect, and no harm,
// if the host program initialised HIP earlier or set the variable itself).
__attribute__((constructor)) static void psmc_hip_more_queues() { setenv("GPU_MAX_HW_QUEUES", "8", 0); }

static void destroy_kids(psmc_hip_ctx *c);

// The five streams of a context, on the whole device or -- psmc_hip_set_cu_range -- masked to a range of its compute units.
static int make_streams(psmc_hip_ctx *c)
{
	hipStream_t *st[5] = {&c->stream, &c->stream2, &c->stream3, &c->stream4, &c->stream5};
	for (hipStream_t *s : st) if (*s) { (void)hipStreamSynchronize(*s); (void)hipStreamDestroy(*s); *s = nullptr; }
	if (c->cu_count <= 0) {
		for (hipStream_t *s : st) if (hipStreamCreateWithFlags(s, hipStreamNonBlocking) != hipSuccess) return PSMC_HIP_EDEVICE;
		return 0;
	}
	int cus = 0;
	if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device) != hipSuccess || cus <= 0) return PSMC_HIP_EDEVICE;
	if (c->cu_first < 0 || c->cu_first + c->cu_count > cus) return PSMC_HIP_EINVAL;
	std::vector<uint32_t> mask((size_t)(cus + 31) / 32, 0u);
	for (int i = c->cu_first; i < c->cu_first + c->cu_count; ++i) mask[(size_t)i >> 5] |= 1u << (i & 31);
	for (hipStream_t *s : st) if (hipExtStreamCreateWithCUMask(s, (uint32_t)mask.size(), mask.data()) != hipSuccess) return PSMC_HIP_EDEVICE;
	return 0;
}

extern "C" int psmc_hip_device_cus(int device)
{
	int cus = 0;
	if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) { (void)hipGetLastError(); return 0; }
	return cus;
}

extern "C" int psmc_hip_set_cu_range(psmc_hip_ctx *c, int first, int count)
{
	if (!c || first < 0 || count < 0) return fail(c, PSMC_HIP_EINVAL, "set_cu_range: bad argument");
	if (c->parent) return fail(c, PSMC_HIP_EINVAL, "set_cu_range: not on a replicate context");
	HIPCHK(c, hipSetDevice(c->device));
	destroy_kids(c); // replicate contexts share the parent's streams
	const int of = c->cu_first, oc = c->cu_count;
	c->cu_first = count > 0 ? first : 0; c->cu_count = count;
	int rc = make_streams(c);
	if (rc) { // leave a usable context behind
		c->cu_first = of; c->cu_count = oc;
		if (make_streams(c) != 0) { c->cu_first = c->cu_count = 0; (void)make_streams(c); }
		return fail(c, rc, "set_cu_range: cannot create the masked streams (range outside the device?)", hipGetLastError());
	}
	return PSMC_HIP_OK;
}

extern "C" int psmc_hip_reserve_tables(psmc_hip_ctx *c)
{
	if (!c) return PSMC_HIP_EINVAL;
	if (c->n_seg < 1) return fail(c, PSMC_HIP_ESTATE, "reserve_tables: no segments loaded");
	HIPCHK(c, hipSetDevice(c->device));
	const bool exact = c->mode == PSMC_HIP_MODE_EXACT || c->ns > 128;
	return ensure_tables(c, exact, 0, true);
}

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
	if (n_states > PSMC_HIP_MAX_STATES) return PSMC_HIP_ENOTSUP; // one thread per state in a work-group of the wide exact kernels (estep_wide.hip)
	int nd = psmc_hip_device_count();
	if (nd <= 0 || device < 0 || device >= nd) return PSMC_HIP_EDEVICE;
	if (hipSetDevice(device) != hipSuccess) return PSMC_HIP_EDEVICE;
	psmc_hip_ctx *c = new (std::nothrow) psmc_hip_ctx();
	if (!c) return PSMC_HIP_ENOMEM;
	// states padded to 64 or 128 (one or two per lane); beyond that to the next multiple of 64: the wide EXACT kernels, whatever the mode
	c->n = n_states; c->ns = n_states > 128 ? (n_states + 63) / 64 * 64 : (n_states > 64 ? 128 : 64); c->device = device; c->mode = mode;
	c->par_len = c->ns > 128 ? 2 * (size_t)c->ns * c->ns + 4 * (size_t)c->ns : psmc_hip_ctx::PAR_LEN;
	if (make_streams(c) != 0) { psmc_hip_destroy(c); return PSMC_HIP_EDEVICE; }
	for (int i = 0; i < 14; ++i)
		if (hipEventCreate(&c->evx[i]) != hipSuccess) { delete c; return PSMC_HIP_EDEVICE; }
	for (int i = 0; i < 10; ++i)
		if (hipEventCreate(&c->ev[i]) != hipSuccess) { delete c; return PSMC_HIP_EDEVICE; }
	if (hipHostMalloc((void **)&c->h_par, c->par_len * sizeof(double), hipHostMallocDefault) != hipSuccess ||
	    hipMalloc((void **)&c->d_par, c->par_len * sizeof(double)) != hipSuccess) {
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
	if (c->x_twin) { psmc_hip_destroy(c->x_twin); c->x_twin = nullptr; } // (it borrows this context's observations and copied its segment list)
	for (psmc_hip_ctx *k : c->kids) psmc_hip_destroy(k);
	c->kids.clear();
	c->share_T = 0; // the shared learning goes with the plans it was made for
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
		if (c->h_mlen) (void)hipHostFree(c->h_mlen);
		if (c->h_chunks) (void)hipHostFree(c->h_chunks);
		if (c->h_mis) (void)hipHostFree(c->h_mis);
		if (c->d_prevx) (void)hipFree(c->d_prevx);
		if (c->d_finv) (void)hipFree(c->d_finv);
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
	                c->d_bw_seg, c->d_bw_par, c->d_bw_tab, c->d_bpar, c->d_s_all, c->d_cu_mask, c->d_lkp, c->d_lkoff, c->d_b2};
	for (void *p : ptrs) if (p) (void)hipFree(p);
	if (c->h_par) (void)hipHostFree(c->h_par);
	if (c->h_cnt) (void)hipHostFree(c->h_cnt);
	if (c->h_ritems) (void)hipHostFree(c->h_ritems);
	if (c->h_mlen) (void)hipHostFree(c->h_mlen);
	if (c->h_chunks) (void)hipHostFree(c->h_chunks);
	if (c->h_mis) (void)hipHostFree(c->h_mis);
	if (c->d_prevx) (void)hipFree(c->d_prevx);
	if (c->d_finv) (void)hipFree(c->d_finv);
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
	if (k == "batch_first") { if (v < 0 || v > 1e6) return PSMC_HIP_EINVAL; c->batch_first = (int)v; return PSMC_HIP_OK; } // (names a replicate context: they all stay)
	destroy_kids(c); // replicate contexts of a batch copied the options when they were made: start them afresh
	if (k == "chunk") { if (v < 0) return PSMC_HIP_EINVAL; c->chunk = (int)v; c->plan_dirty = true; }
	else if (k == "warmup") { if (v < 0) return PSMC_HIP_EINVAL; c->warmup = (int)v; c->plan_dirty = true; }
	else if (k == "max_rounds") c->max_rounds = (int)v;
	else if (k == "structured") { c->struct_opt = v != 0 ? 1 : 0; }
	else if (k == "learn") { c->learn = v != 0 ? 1 : 0; }
	else if (k == "merge") { c->merge = v != 0 ? 1 : 0; c->plan_dirty = true; c->chunk_cap = 0; }
	else if (k == "adapt") { c->adapt = v != 0 ? 1 : 0; c->plan_dirty = true; c->chunk_cap = 0; }
	else if (k == "prev_start") { c->prev_start = v != 0 ? 1 : 0; c->prev_ok = false; c->plan_dirty = true; c->chunk_cap = 0; }
	else if (k == "warm_shift") { if (v < 0 || v > 4) return PSMC_HIP_EINVAL; c->warm_shift = (int)v; c->warm_shift_set = true; c->plan_dirty = true; }
	else if (k == "merge1") { if (v < -1 || v > 1) return PSMC_HIP_EINVAL; c->merge1 = (int)v; c->plan_dirty = true; }
	else if (k == "lanes8") { if (v < -1 || v > 1) return PSMC_HIP_EINVAL; c->lanes8 = (int)v; }
	else if (k == "gap_tiles") { c->gap_tiles = v != 0 ? 1 : 0; c->plan_dirty = true; c->items_dirty = true; }
	else if (k == "coarse") { if (v < -1 || v > 16) return PSMC_HIP_EINVAL; c->coarse = (int)v; c->plan_dirty = true; c->items_dirty = true; }
	else if (k == "kc_sub") { if (v < 1 || v > 16) return PSMC_HIP_EINVAL; c->kc_sub = (int)v; c->kc_sub_set = true; c->plan_dirty = true; c->items_dirty = true; }
	else if (k == "two_phase") { if (v != -1 && v != 0 && v != 2) return PSMC_HIP_EINVAL; c->two_phase = (int)v; c->plan_dirty = true; c->items_dirty = true; }
	else if (k == "kc_min") { if (v < -1) return PSMC_HIP_EINVAL; c->kc_min = (int)v; c->items_dirty = true; }
	else if (k == "ckpt") { c->ckpt = v != 0 ? 1 : 0; }
	else if (k == "fuse") { c->fuse = v != 0 ? 1 : 0; c->plan_dirty = true; }
	else if (k == "fuse128") { if (v != 0 && v != 2) return PSMC_HIP_EINVAL; c->fuse128 = (int)v; c->plan_dirty = true; c->items_dirty = true; }
	else if (k == "group_cap") { if (v < 0) return PSMC_HIP_EINVAL; c->group_cap = (int)v; c->items_dirty = true; }
	else if (k == "share_learn") { c->share_learn = v != 0 ? 1 : 0; }
	else if (k == "exact_refwd") { if (v < -1 || v > 2) return PSMC_HIP_EINVAL; c->exact_refwd = (int)v; c->reserved_refwd = -1; c->reserved_cap = 0; if (c->d_b2) { (void)hipFree(c->d_b2); c->d_b2 = nullptr; c->b2_bins = c->b2_alloc = 0; } }
	else if (k == "batch_sort") { c->batch_sort = v != 0 ? 1 : 0; }
	else if (k == "batch_tailfill") c->batch_tailfill = v != 0;
	else if (k == "batch_major") c->batch_major = v != 0;
	else if (k == "batch_bins") { if (v < 0) return PSMC_HIP_EINVAL; c->batch_bins = (int64_t)v; }
	else if (k == "overlap") c->overlap = v != 0 ? 1 : 0;
	else if (k == "warm_tol") c->warm_tol = v;
	else if (k == "rep_impl") c->rep_impl = v < 0 ? -1 : (v != 0 ? 1 : 0);
	else return PSMC_HIP_EINVAL;
	return PSMC_HIP_OK;
}

int set_segments_common(psmc_hip_ctx *c, int n_seg, const int32_t *L)
{
	destroy_kids(c); // batch children hold plans over the previous segments
	c->reserved_refwd = -1; c->reserved_cap = 0;
	if (c->d_b2) { (void)hipFree(c->d_b2); c->d_b2 = nullptr; c->b2_bins = c->b2_alloc = 0; }
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
bool fill_params(const psmc_hip_ctx *c, const double *a, const double *e, const double *a0, double *dst)
{
	const int n = c->n;
	if (c->ns > 128) { // wide exact kernels: a | aT | e(3) | a0, stride S (estep_wide.hip)
		const size_t S = (size_t)c->ns;
		double *pa = dst, *pt = pa + S * S, *pe = pt + S * S, *pa0 = pe + 3 * S;
		memset(pa, 0, c->par_len * sizeof(double));
		for (int k = 0; k < n; ++k) {
			for (int l = 0; l < n; ++l) { pa[k * S + l] = a[(size_t)k * n + l]; pt[l * S + k] = a[(size_t)k * n + l]; }
			pe[k] = e[k]; pe[S + k] = e[n + k];
			pa0[k] = a0[k];
		}
		for (size_t k = 0; k < S; ++k) pe[2 * S + k] = 1.0; // khmm.c:21
		return false;
	}
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

int stage_params(psmc_hip_ctx *c, const double *a, const double *e, const double *a0, hipStream_t st)
{
	HIPCHK(c, hipStreamSynchronize(st)); // previous async copy out of the pinned staging buffer
	c->use_struct = fill_params(c, a, e, a0, c->h_par);
	HIPCHK(c, hipMemcpyAsync(c->d_par, c->h_par, c->par_len * sizeof(double), hipMemcpyHostToDevice, st));
	return 0;
}

static int ensure_tables_once(psmc_hip_ctx *c, bool need_b, int64_t bins, bool need_f)
{
	int rc;
	if (c->tab_bins < bins) { // grow: everything goes, and comes back as needed
		if (c->d_f) { (void)hipFree(c->d_f); c->d_f = nullptr; }
		if (c->d_b) { (void)hipFree(c->d_b); c->d_b = nullptr; }
		if (c->d_b2) { (void)hipFree(c->d_b2); c->d_b2 = nullptr; c->b2_bins = c->b2_alloc = 0; c->reserved_cap = 0; }
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

int ensure_tables(psmc_hip_ctx *c, bool need_b, int64_t want_bins, bool need_f)
{
	if (c->parent) { // a batch child works in its parent's tables (same segments, one E-step at a time)
		int rc = ensure_tables(c->parent, need_b);
		c->d_f = c->parent->d_f; c->d_b = c->parent->d_b; c->d_s = c->parent->d_s; c->d_sb = c->parent->d_sb;
		c->tab_bins = c->parent->tab_bins; c->have_b = c->parent->have_b;
		return rc;
	}
	++c->tab_serial; // whoever asks for the tables is about to write them (api_fast.hip: is the X of the last fast E-step still there?)
	const int64_t bins = std::max(c->total, want_bins) + 128;
	int rc = ensure_tables_once(c, need_b, bins, need_f);
	if (rc == PSMC_HIP_ENOMEM && c->tab_bins > bins) {
		// The tables were sized for another layout -- an exact batch without the f table holds twice the bins of one with it -- and the
		// table this call adds does not fit beside them at that size (a single E-step, decode or get_tables after such a
		// batch).  Start over at what THIS call needs.
		if (c->d_f) { (void)hipFree(c->d_f); c->d_f = nullptr; }
		if (c->d_b) { (void)hipFree(c->d_b); c->d_b = nullptr; }
		if (c->d_b2) { (void)hipFree(c->d_b2); c->d_b2 = nullptr; c->b2_bins = c->b2_alloc = 0; }
		if (c->d_s) { (void)hipFree(c->d_s); c->d_s = nullptr; }
		if (c->d_sb) { (void)hipFree(c->d_sb); c->d_sb = nullptr; }
		c->have_b = false; c->tab_bins = 0; c->reserved_refwd = -1; c->reserved_cap = 0;
		(void)hipGetLastError();
		rc = ensure_tables_once(c, need_b, bins, need_f);
	}
	return rc;
}

// fused backward sweep + counts: structured matrices; 64 states, or 128 with "fuse128"
bool fused_counts(const psmc_hip_ctx *c) { return c->fuse && (c->ns == 64 || (c->ns == 128 && c->fuse128)); }

// the column-per-lane transfer-matrix kernel (k_kcol2_struct): 64 states.  With 65..128 states it takes 2.3x fewer vector
// instructions too, but its chain path ends later and the E-step is slower -- round 2: 30.6 vs 28.3 ms factored; round 3, with
// the run tiles in the second launch of the counts and the faster chain kernel: 45.7 vs 44.1 ms full counts, 29.2 vs 27.3
// factored, whatever kc_sub (profiles/r03_kc_min_sweep.txt) -- so that instantiation is not built.
bool kcol2_on(const psmc_hip_ctx *c) { return c->ns == 64; }

void fill_common(psmc_hip_ctx *c, EstepLaunch &p, hipStream_t st, const double *par_base)
{
	memset(&p, 0, sizeof(p));
	p.stream = st;
	p.rep_impl = c->rep_impl; p.n_states = c->n;
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
	} else if (c->ns > 128) { // a | aT | e(3) | a0 (fill_params); exact kernels only
		const size_t S = (size_t)c->ns;
		p.d_aeT = pb + S * S; p.d_e = pb + 2 * S * S; p.d_a0 = pb + 2 * S * S + 3 * S;
		p.d_re = p.d_sp = p.d_kcc = nullptr; p.structured = 0; p.fused = 0; p.ckpt = 0;
	}
	p.d_obs = c->d_obs; p.d_seg_off = c->d_seg_off; p.d_seg_len = c->d_seg_len;
	p.d_work = c->d_work; p.n_work = (int)c->work.size();
	p.d_f = c->d_f; p.d_b = c->d_b; p.d_s = c->d_s; p.d_sb = c->d_sb;
	for (int i = 0; i < 10; ++i) p.ev[i] = c->mode == PSMC_HIP_MODE_FAST || i < 5 ? c->ev[i] : nullptr;
}

void collect_timing(psmc_hip_ctx *c)
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
double host_lk(const double *s, int L)
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
int ensure_seg_outputs(psmc_hip_ctx *c, int nw)
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
	if (c->mode != PSMC_HIP_MODE_EXACT && c->ns <= 128) return fail(c, PSMC_HIP_ENOTSUP, "estep_segments: exact mode only");
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

int estep_exact(psmc_hip_ctx *c, const double *a, const double *e, const double *a0, double *A, double *E,
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

extern "C" int psmc_hip_estep(psmc_hip_ctx *c, const double *a, const double *e, const double *a0, double *A, double *E,
                              double *A0, double *LL, double *chk)
{
	if (!c || !a || !e || !a0) return fail(c, PSMC_HIP_EINVAL, "estep: bad argument");
	// beyond 128 states there is no fast path: a fast-mode context runs the wide exact kernels (inside every fast tolerance)
	if (c->mode == PSMC_HIP_MODE_EXACT || c->ns > 128) return estep_exact(c, a, e, a0, A, E, A0, LL, chk);
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
	if (c->mode != PSMC_HIP_MODE_EXACT && c->ns <= 128) return fail(c, PSMC_HIP_ENOTSUP, "decode: exact mode only");
	if (!c->d_f || !c->have_b || c->tables_batch) return fail(c, PSMC_HIP_ESTATE, "decode: no single E-step yet");
	HIPCHK(c, hipSetDevice(c->device));
	const int L = c->L[seg];
	int32_t *dp = nullptr; double *dm = nullptr;
	if (hipMalloc((void **)&dp, sizeof(int32_t) * (size_t)L) != hipSuccess) return fail(c, PSMC_HIP_ENOMEM, "hipMalloc");
	if (hipMalloc((void **)&dm, sizeof(double) * (size_t)L) != hipSuccess) { (void)hipFree(dp); return fail(c, PSMC_HIP_ENOMEM, "hipMalloc"); }
	int rc = c->ns > 128 ? launch_post_decode_wide(c->stream, c->d_f, c->d_b, c->d_s, c->off[seg], L, c->n, c->ns, dp, dm)
	                     : launch_post_decode(c->stream, c->d_f, c->d_b, c->d_s, c->off[seg], L, c->n, c->ns, dp, dm);
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
	if (c->mode != PSMC_HIP_MODE_EXACT && c->ns <= 128) return fail(c, PSMC_HIP_ENOTSUP, "posterior: exact mode only");
	if (!c->d_f || !c->have_b || c->tables_batch) return fail(c, PSMC_HIP_ESTATE, "posterior: no single E-step yet");
	HIPCHK(c, hipSetDevice(c->device));
	const int L = c->L[seg], n = c->n;
	double *dp = nullptr, *dr = nullptr;
	if (post && hipMalloc((void **)&dp, sizeof(double) * (size_t)L * n) != hipSuccess) return fail(c, PSMC_HIP_ENOMEM, "hipMalloc");
	if (recomb && hipMalloc((void **)&dr, sizeof(double) * (size_t)L) != hipSuccess) { if (dp) (void)hipFree(dp); return fail(c, PSMC_HIP_ENOMEM, "hipMalloc"); }
	const double *d_e = c->ns > 128 ? c->d_par + 2 * (size_t)c->ns * c->ns : (c->ns == 128 ? c->d_par + 32768 : c->d_par + 4 * 4096);
	int rc = c->ns > 128 ? launch_post_full_wide(c->stream, c->d_par, d_e, c->d_obs, c->d_f, c->d_b, c->d_s, c->off[seg], L, n, c->ns, dp, dr)
	                     : launch_post_full(c->stream, c->d_par, d_e, c->d_obs, c->d_f, c->d_b, c->d_s, c->off[seg], L, n, c->ns, dp, dr);
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
	if (c->mode != PSMC_HIP_MODE_EXACT && c->ns <= 128) return fail(c, PSMC_HIP_ENOTSUP, "post_counts: exact mode only");
	if (!c->d_f || !c->have_b || c->tables_batch) return fail(c, PSMC_HIP_ESTATE, "post_counts: no single E-step yet");
	HIPCHK(c, hipSetDevice(c->device));
	const int L = c->L[seg], n = c->n, min_l = L < l ? L : l;
	if (min_l == 0) return PSMC_HIP_OK;
	int32_t *d1 = nullptr; double *dc = nullptr;
	if (hipMalloc((void **)&d1, sizeof(int32_t) * (size_t)min_l * n_cnt) != hipSuccess) return fail(c, PSMC_HIP_ENOMEM, "hipMalloc");
	if (hipMalloc((void **)&dc, sizeof(double) * (size_t)n * n_cnt) != hipSuccess) { (void)hipFree(d1); return fail(c, PSMC_HIP_ENOMEM, "hipMalloc"); }
	hipError_t e0 = hipMemcpyAsync(d1, cnt1, sizeof(int32_t) * (size_t)min_l * n_cnt, hipMemcpyHostToDevice, c->stream);
	hipError_t e1 = hipMemcpyAsync(dc, cnt, sizeof(double) * (size_t)n * n_cnt, hipMemcpyHostToDevice, c->stream);
	int rc = c->ns > 128 ? launch_post_counts_wide(c->stream, c->d_f, c->d_b, c->d_s, c->off[seg], min_l, d1, n_cnt, n, c->ns, dc)
	                     : launch_post_counts(c->stream, c->d_f, c->d_b, c->d_s, c->off[seg], min_l, d1, n_cnt, n, c->ns, dc);
	hipError_t e2 = hipMemcpyAsync(cnt, dc, sizeof(double) * (size_t)n * n_cnt, hipMemcpyDeviceToHost, c->stream);
	hipError_t e3 = hipStreamSynchronize(c->stream);
	(void)hipFree(d1); (void)hipFree(dc);
	if (rc || e0 != hipSuccess || e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) return fail(c, PSMC_HIP_EDEVICE, "post_counts");
	return PSMC_HIP_OK;
}
