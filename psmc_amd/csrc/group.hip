// group.hip -- one E-step sharded over several GPUs of a node, inside the C-ABI (include/psmc_hip.h, psmc_hip_group_*).
//
// Segments are independent given the parameters (lh3/psmc em.c:36-55: the only state carried across the segment
// loop is `he_sum += he` and `LL +=`), so every device runs the E-step of its own segments and ONE exchange per EM
// iteration replaces hmm_add_expect (khmm.c:346-359):
//   fast   RCCL all-reduce (sum, f64) of the n*n + 2n + 1 doubles [A | E | LL] each device's reduction kernel left
//          in HBM -- 34 KB at n = 64: latency-bound on xGMI, any algorithm.  One process, one communicator per
//          device (ncclCommInitAll), the per-device calls fused in a ncclGroupStart / ncclGroupEnd section.
//   exact  every shard returns the per-segment `he` of its segments and the host adds them in the GLOBAL input
//          order: floating-point addition is not associative and bit-identity with the reference's serial loop is
//          the point of that mode.
// Shards = longest-processing-time-first partition of the segments by length.  librccl is opened on first use, so a
// single-GPU user of the library never loads it.  A device may be listed more than once (two shards on one GPU: how
// the tests exercise this file on a 1-GPU box); RCCL needs distinct devices, so such a group adds the shards' vectors
// on the host in shard order instead -- and so does a group in auto mode ("rccl" = -1) whose librccl cannot be opened
// or whose communicator cannot be created (the reason is kept: psmc_hip_group_selfcheck reports the path in use).
// Only "rccl" = 1 makes a missing RCCL an error.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <string.h>
#include <algorithm>
#include <string>
#include <thread>
#include <vector>
#include "psmc_hip.h"

namespace {

struct Rccl {
	void *lib = nullptr;
	ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
	ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
	ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t (*GroupStart)() = nullptr;
	ncclResult_t (*GroupEnd)() = nullptr;
	const char *(*GetErrorString)(ncclResult_t) = nullptr;
	bool open(std::string &err)
	{
		if (lib) return true;
		// PSMC_HIP_RCCL_LIB: the library to load instead (a site's own RCCL build; tests/stub_rccl: a single-process stand-in that
		// lets the multi-shard branch below run on a one-GPU box -- see the "rccl" = 2 option)
		if (const char *own = getenv("PSMC_HIP_RCCL_LIB")) {
			if (!(lib = dlopen(own, RTLD_NOW | RTLD_LOCAL))) { err = std::string("cannot open PSMC_HIP_RCCL_LIB: ") + dlerror(); return false; }
		} else {
			const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
			for (const char *nm : names) if ((lib = dlopen(nm, RTLD_NOW | RTLD_LOCAL))) break;
			if (!lib) { err = std::string("cannot open librccl: ") + dlerror(); return false; }
		}
		CommInitAll = (decltype(CommInitAll))dlsym(lib, "ncclCommInitAll");
		CommDestroy = (decltype(CommDestroy))dlsym(lib, "ncclCommDestroy");
		AllReduce = (decltype(AllReduce))dlsym(lib, "ncclAllReduce");
		GroupStart = (decltype(GroupStart))dlsym(lib, "ncclGroupStart");
		GroupEnd = (decltype(GroupEnd))dlsym(lib, "ncclGroupEnd");
		GetErrorString = (decltype(GetErrorString))dlsym(lib, "ncclGetErrorString");
		if (!CommInitAll || !CommDestroy || !AllReduce || !GroupStart || !GroupEnd || !GetErrorString) { err = "librccl lacks an expected symbol"; dlclose(lib); lib = nullptr; return false; }
		return true;
	}
};

} // namespace

struct psmc_hip_group {
	int n = 0, mode = 0, n_sh = 0, n_seg = 0;
	std::vector<psmc_hip_ctx *> sh;
	std::vector<int> dev;
	std::vector<std::vector<int32_t>> segs_of;  // global segment ids of each shard, ascending
	std::vector<int32_t> shard_of, local_of;     // per global segment
	std::vector<double *> d_stats;               // per shard: the vector the collective runs on
	std::vector<double *> h_stats;               // per shard: pinned host copy of it (asynchronous read-back on the shard's stream)
	size_t stats_len = 0;
	std::vector<hipStream_t> st;
	std::string err;
	int want_rccl = -1;                          // "rccl": -1 auto (distinct devices, more than one shard), 0 never, 1 always, 2 (tests): always, and
	                                             // a device may repeat -- real RCCL refuses that communicator, the stand-in of tests/stub_rccl does not
	bool distinct = true;
	Rccl rccl;
	std::vector<ncclComm_t> comm;
	int last_reduce = 0;                         // 0 none (one shard), 1 RCCL all-reduce, 2 host sum in shard order, 3 ordered per-segment sum (exact)
	std::string rccl_note;                       // auto mode: why RCCL is not in use (library missing, communicator refused)
};

static int gfail(psmc_hip_group *g, int code, const std::string &what) { if (g) g->err = what; return code; }

extern "C" int psmc_hip_group_create(psmc_hip_group **out, int n_states, int n_dev, const int *devices, int mode)
{
	if (!out) return PSMC_HIP_EINVAL;
	*out = nullptr;
	if (n_dev < 1 || n_dev > 64 || !devices) return PSMC_HIP_EINVAL;
	psmc_hip_group *g = new (std::nothrow) psmc_hip_group();
	if (!g) return PSMC_HIP_ENOMEM;
	g->n = n_states; g->mode = mode; g->n_sh = n_dev;
	g->dev.assign(devices, devices + n_dev);
	for (int i = 0; i < n_dev; ++i)
		for (int j = 0; j < i; ++j) if (devices[i] == devices[j]) g->distinct = false;
	g->sh.assign(n_dev, nullptr); g->d_stats.assign(n_dev, nullptr); g->h_stats.assign(n_dev, nullptr); g->st.assign(n_dev, nullptr);
	for (int i = 0; i < n_dev; ++i) {
		int rc = psmc_hip_create(&g->sh[i], n_states, devices[i], mode);
		if (rc == 0 && mode == PSMC_HIP_MODE_FAST) {
			const size_t len = std::max((size_t)n_states * n_states + 2 * (size_t)n_states + 1, (size_t)7 * n_states + 1); // [A | E | LL], or the factored statistics (longer below 5 states)
			g->stats_len = len;
			if (hipSetDevice(devices[i]) != hipSuccess || hipMalloc((void **)&g->d_stats[i], sizeof(double) * len) != hipSuccess ||
			    hipHostMalloc((void **)&g->h_stats[i], sizeof(double) * len, hipHostMallocDefault) != hipSuccess ||
			    hipStreamCreateWithFlags(&g->st[i], hipStreamNonBlocking) != hipSuccess) rc = PSMC_HIP_EDEVICE;
		}
		if (rc) { psmc_hip_group_destroy(g); return rc; }
	}
	*out = g;
	return PSMC_HIP_OK;
}

extern "C" void psmc_hip_group_destroy(psmc_hip_group *g)
{
	if (!g) return;
	for (size_t i = 0; i < g->comm.size(); ++i) if (g->comm[i]) (void)g->rccl.CommDestroy(g->comm[i]);
	for (int i = 0; i < g->n_sh; ++i) {
		if (g->dev.size() > (size_t)i) (void)hipSetDevice(g->dev[i]);
		if (g->st[i]) { (void)hipStreamSynchronize(g->st[i]); (void)hipStreamDestroy(g->st[i]); }
		if (g->d_stats[i]) (void)hipFree(g->d_stats[i]);
		if (g->h_stats[i]) (void)hipHostFree(g->h_stats[i]);
		if (g->sh[i]) psmc_hip_destroy(g->sh[i]);
	}
	delete g;
}

extern "C" const char *psmc_hip_group_last_error(const psmc_hip_group *g) { return g ? g->err.c_str() : ""; }

extern "C" int psmc_hip_group_set_option(psmc_hip_group *g, const char *key, double value)
{
	if (!g || !key) return PSMC_HIP_EINVAL;
	if (!strcmp(key, "rccl")) { g->want_rccl = value < 0 ? -1 : (value == 2 ? 2 : (value != 0 ? 1 : 0)); return PSMC_HIP_OK; }
	for (int i = 0; i < g->n_sh; ++i) {
		const int rc = psmc_hip_set_option(g->sh[i], key, value);
		if (rc) return gfail(g, rc, std::string("set_option: ") + key);
	}
	return PSMC_HIP_OK;
}

// Longest-processing-time-first over the segment lengths; a shard keeps its segments in input order.
extern "C" int psmc_hip_group_load_segments(psmc_hip_group *g, int n_seg, const uint8_t *const *seq, const int32_t *L)
{
	if (!g || n_seg < 1 || !seq || !L) return gfail(g, PSMC_HIP_EINVAL, "group_load_segments: bad argument");
	std::vector<int32_t> order(n_seg);
	for (int i = 0; i < n_seg; ++i) order[i] = i;
	std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) { return L[x] > L[y]; });
	std::vector<int64_t> load(g->n_sh, 0);
	g->segs_of.assign(g->n_sh, std::vector<int32_t>());
	for (int32_t i : order) {
		int best = 0;
		for (int s = 1; s < g->n_sh; ++s) if (load[s] < load[best]) best = s;
		g->segs_of[best].push_back(i); load[best] += L[i];
	}
	g->n_seg = n_seg;
	g->shard_of.assign(n_seg, 0); g->local_of.assign(n_seg, 0);
	for (int s = 0; s < g->n_sh; ++s) {
		std::sort(g->segs_of[s].begin(), g->segs_of[s].end());
		std::vector<const uint8_t *> p; std::vector<int32_t> l;
		for (size_t j = 0; j < g->segs_of[s].size(); ++j) {
			const int32_t i = g->segs_of[s][j];
			g->shard_of[i] = s; g->local_of[i] = (int32_t)j;
			p.push_back(seq[i]); l.push_back(L[i]);
		}
		if (p.empty()) continue; // more shards than segments: this one idles
		const int rc = psmc_hip_load_segments(g->sh[s], (int)p.size(), p.data(), l.data());
		if (rc) return gfail(g, rc, std::string("shard ") + std::to_string(s) + ": " + psmc_hip_last_error(g->sh[s]));
	}
	return PSMC_HIP_OK;
}

extern "C" int psmc_hip_group_info(psmc_hip_group *g, int *n_shards, int32_t *shard_of_seg, int *last_reduce)
{
	if (!g) return PSMC_HIP_EINVAL;
	if (n_shards) *n_shards = g->n_sh;
	if (shard_of_seg) for (int i = 0; i < g->n_seg; ++i) shard_of_seg[i] = g->shard_of[i];
	if (last_reduce) *last_reduce = g->last_reduce;
	return PSMC_HIP_OK;
}

// run f(shard) on one host thread per shard that holds segments; first error wins
template <class F> static int for_shards(psmc_hip_group *g, F f)
{
	std::vector<int> rc(g->n_sh, 0);
	std::vector<std::thread> th;
	int n_live = 0, only = -1;
	for (int s = 0; s < g->n_sh; ++s) if (!g->segs_of[s].empty()) { ++n_live; only = s; }
	if (n_live == 1) rc[only] = f(only); // one shard: the caller's thread (a thread start and join cost as much as the whole tail of an E-step)
	else
		for (int s = 0; s < g->n_sh; ++s) {
			if (g->segs_of[s].empty()) continue;
			th.emplace_back([&, s]() { rc[s] = f(s); });
		}
	for (std::thread &t : th) t.join();
	for (int s = 0; s < g->n_sh; ++s)
		if (rc[s]) return gfail(g, rc[s], std::string("shard ") + std::to_string(s) + " (device " + std::to_string(g->dev[s]) + "): " + psmc_hip_last_error(g->sh[s]));
	return 0;
}

// sum of the live shards' device vectors into host memory: RCCL all-reduce over the devices, or the host adds them
static int reduce_vectors(psmc_hip_group *g, size_t len, std::vector<double> &out, const std::vector<char> &live)
{
	out.assign(len, 0.0);
	int n_live = 0, first = -1;
	for (int s = 0; s < g->n_sh; ++s) if (live[s]) { ++n_live; if (first < 0) first = s; }
	bool use_rccl = g->want_rccl >= 1 || (g->want_rccl < 0 && g->distinct && n_live > 1 && n_live == g->n_sh); // auto: every device takes part
	if (use_rccl && !g->distinct && g->want_rccl != 2) return gfail(g, PSMC_HIP_EINVAL, "rccl = 1 needs distinct devices");
	if (use_rccl && n_live != g->n_sh) return gfail(g, PSMC_HIP_ESTATE, "RCCL all-reduce: a shard holds no segment (fewer segments than devices)");
	if (use_rccl && g->comm.empty()) {
		// auto mode falls back to the host sum when there is no usable RCCL on this node (same result, one more copy per
		// shard); "rccl" = 1 keeps the hard error
		std::string why;
		if (!g->rccl.open(why)) {
			if (g->want_rccl >= 1) return gfail(g, PSMC_HIP_EDEVICE, why);
			g->rccl_note = why; g->want_rccl = 0; use_rccl = false;
		} else {
			g->comm.assign(g->n_sh, nullptr);
			const ncclResult_t r = g->rccl.CommInitAll(g->comm.data(), g->n_sh, g->dev.data());
			if (r != ncclSuccess) {
				g->comm.clear();
				why = std::string("ncclCommInitAll: ") + g->rccl.GetErrorString(r);
				if (g->want_rccl >= 1) return gfail(g, PSMC_HIP_EDEVICE, why);
				g->rccl_note = why; g->want_rccl = 0; use_rccl = false;
			}
		}
	}
	if (use_rccl) {
		ncclResult_t r = g->rccl.GroupStart();
		for (int s = 0; s < g->n_sh && r == ncclSuccess; ++s) {
			(void)hipSetDevice(g->dev[s]);
			r = g->rccl.AllReduce(g->d_stats[s], g->d_stats[s], len, ncclDouble, ncclSum, g->comm[s], g->st[s]); // in place, on the stream the E-step wrote it on
		}
		const ncclResult_t r2 = g->rccl.GroupEnd();
		if (r != ncclSuccess || r2 != ncclSuccess) return gfail(g, PSMC_HIP_EDEVICE, std::string("ncclAllReduce: ") + g->rccl.GetErrorString(r != ncclSuccess ? r : r2));
		for (int s = 0; s < g->n_sh; ++s) { (void)hipSetDevice(g->dev[s]); if (hipStreamSynchronize(g->st[s]) != hipSuccess) return gfail(g, PSMC_HIP_EDEVICE, "stream sync after all-reduce"); }
		(void)hipSetDevice(g->dev[0]);
		if (hipMemcpyAsync(g->h_stats[0], g->d_stats[0], sizeof(double) * len, hipMemcpyDeviceToHost, g->st[0]) != hipSuccess || hipStreamSynchronize(g->st[0]) != hipSuccess)
			return gfail(g, PSMC_HIP_EDEVICE, "copy of the reduced statistics");
		memcpy(out.data(), g->h_stats[0], sizeof(double) * len);
		g->last_reduce = 1;
		return 0;
	}
	for (int s = 0; s < g->n_sh; ++s) { // every shard's read-back is queued behind its E-step on its own stream, then waited for in shard order
		if (!live[s]) continue;
		(void)hipSetDevice(g->dev[s]);
		if (hipMemcpyAsync(g->h_stats[s], g->d_stats[s], sizeof(double) * len, hipMemcpyDeviceToHost, g->st[s]) != hipSuccess)
			return gfail(g, PSMC_HIP_EDEVICE, "copy of a shard's statistics");
	}
	for (int s = 0; s < g->n_sh; ++s) {
		if (!live[s]) continue;
		(void)hipSetDevice(g->dev[s]);
		if (hipStreamSynchronize(g->st[s]) != hipSuccess) return gfail(g, PSMC_HIP_EDEVICE, "copy of a shard's statistics");
		const double *tmp = g->h_stats[s];
		if (s == first) memcpy(out.data(), tmp, sizeof(double) * len); else for (size_t i = 0; i < len; ++i) out[i] += tmp[i];
	}
	g->last_reduce = n_live > 1 ? 2 : 0;
	return 0;
}

static std::vector<char> live_shards(const psmc_hip_group *g)
{
	std::vector<char> live(g->n_sh, 0);
	for (int s = 0; s < g->n_sh; ++s) live[s] = !g->segs_of.empty() && !g->segs_of[s].empty();
	return live;
}

// First contact with a multi-GPU node, before any E-step: every shard's device answers, the exchange the E-steps will
// use comes up (RCCL communicator over the listed devices, or the host sum) and adds correctly IN STREAM ORDER -- shard
// s writes s + 1 into its vector with an asynchronous copy on its E-step stream, the all-reduce follows on the same
// stream, and every device must then hold n(n+1)/2.  A failure names the step (psmc_hip_group_last_error).
extern "C" int psmc_hip_group_selfcheck(psmc_hip_group *g, int out[4])
{
	if (!g) return PSMC_HIP_EINVAL;
	if (out) out[0] = g->n_sh, out[1] = 0, out[2] = 0, out[3] = 0;
	if (g->mode != PSMC_HIP_MODE_FAST) { if (out) out[1] = 3; return PSMC_HIP_OK; } // exact mode exchanges nothing between devices
	const int N = g->n_sh;
	std::vector<double> src(N);
	for (int s = 0; s < N; ++s) {
		src[s] = (double)(s + 1);
		if (hipSetDevice(g->dev[s]) != hipSuccess) return gfail(g, PSMC_HIP_EDEVICE, "selfcheck: hipSetDevice(" + std::to_string(g->dev[s]) + ")");
		if (hipMemcpyAsync(g->d_stats[s], &src[s], sizeof(double), hipMemcpyHostToDevice, g->st[s]) != hipSuccess)
			return gfail(g, PSMC_HIP_EDEVICE, "selfcheck: copy to device " + std::to_string(g->dev[s]));
	}
	std::vector<double> v;
	const int rc = reduce_vectors(g, 1, v, std::vector<char>(N, 1));
	if (rc) { g->err = "selfcheck: " + g->err; return rc; }
	const double want = 0.5 * N * (N + 1);
	if (v[0] != want) return gfail(g, PSMC_HIP_EDEVICE, "selfcheck: the exchange returned " + std::to_string(v[0]) + ", expected " + std::to_string(want));
	if (g->last_reduce == 1) // after an all-reduce EVERY device holds the sum
		for (int s = 0; s < N; ++s) {
			double x = 0.0;
			(void)hipSetDevice(g->dev[s]);
			if (hipMemcpy(&x, g->d_stats[s], sizeof(double), hipMemcpyDeviceToHost) != hipSuccess || x != want)
				return gfail(g, PSMC_HIP_EDEVICE, "selfcheck: device " + std::to_string(g->dev[s]) + " holds " + std::to_string(x) + " after the all-reduce, expected " + std::to_string(want));
		}
	if (out) { out[1] = g->last_reduce; out[2] = g->comm.empty() ? 0 : 1; out[3] = g->rccl_note.empty() ? 0 : 1; }
	if (!g->rccl_note.empty()) g->err = "RCCL not in use: " + g->rccl_note; // informational; the call succeeded
	return PSMC_HIP_OK;
}

extern "C" int psmc_hip_group_estep(psmc_hip_group *g, const double *a, const double *e, const double *a0, double *A, double *E,
                                    double *A0, double *LL, double *chk)
{
	if (!g || !a || !e || !a0) return gfail(g, PSMC_HIP_EINVAL, "group_estep: bad argument");
	if (g->n_seg < 1) return gfail(g, PSMC_HIP_ESTATE, "group_estep: no segments loaded");
	const int n = g->n;
	if (g->mode == PSMC_HIP_MODE_FAST) {
		int rc = for_shards(g, [&](int s) { return psmc_hip_estep_device(g->sh[s], a, e, a0, g->d_stats[s], g->st[s]); });
		std::vector<double> v;
		if (rc == PSMC_HIP_ENOTSUP && n > 64) { // (only the wide models have that fallback)
			// 65..128 states and a matrix without the PSMC form (e.g. after psmc_cap_matrix): the device-resident entry point
			// has no kernels for it, psmc_hip_estep falls back to the exact ones -- do the same per shard and add the host
			// vectors in shard order, so that a command that works on one GPU works on a device list
			const size_t len = (size_t)n * n + 2 * n + 1;
			std::vector<std::vector<double>> hv(g->n_sh, std::vector<double>(len, 0.0));
			rc = for_shards(g, [&](int s) { return psmc_hip_estep(g->sh[s], a, e, a0, hv[s].data(), hv[s].data() + (size_t)n * n, nullptr, &hv[s][len - 1], nullptr); });
			if (rc) return rc;
			v.assign(len, 0.0);
			int n_live = 0;
			for (int s = 0; s < g->n_sh; ++s) {
				if (g->segs_of[s].empty()) continue;
				if (n_live++ == 0) v = hv[s]; else for (size_t i = 0; i < len; ++i) v[i] += hv[s][i];
			}
			g->last_reduce = n_live > 1 ? 2 : 0;
			g->err.clear(); // the ENOTSUP of the first attempt is not this call's result
		} else {
			if (rc) return rc;
			if ((rc = reduce_vectors(g, (size_t)n * n + 2 * n + 1, v, live_shards(g)))) return rc;
		}
		if (A) memcpy(A, v.data(), sizeof(double) * n * n);
		if (E) memcpy(E, v.data() + (size_t)n * n, sizeof(double) * 2 * n);
		if (LL) *LL = v[(size_t)n * n + 2 * n];
		if (A0) memset(A0, 0, sizeof(double) * n);                 // fast mode does not compute A0 (unused downstream, khmm.c:321-322) ...
		if (chk) for (int i = 0; i < g->n_seg; ++i) chk[i] = 1.0;  // ... nor the khmm.c:237-238 self-check: as psmc_hip_estep in fast mode (include/psmc_hip.h)
		return PSMC_HIP_OK;
	}
	// exact: per-segment statistics from every shard, added in input order on the host (hmm_add_expect, khmm.c:346-359)
	std::vector<std::vector<double>> sA(g->n_sh), sE(g->n_sh), sA0(g->n_sh), sLL(g->n_sh), sC(g->n_sh);
	int rc = for_shards(g, [&](int s) {
		const size_t m = g->segs_of[s].size();
		sA[s].resize(m * n * n); sE[s].resize(m * 3 * n); sA0[s].resize(m * n); sLL[s].resize(m); sC[s].resize(m);
		return psmc_hip_estep_segments(g->sh[s], a, e, a0, sA[s].data(), sE[s].data(), sA0[s].data(), sLL[s].data(), sC[s].data());
	});
	if (rc) return rc;
	std::vector<double> tA((size_t)n * n, 0.0), tE((size_t)2 * n, 0.0), tA0(n, 0.0);
	double ll = 0.0;
	for (int i = 0; i < g->n_seg; ++i) {
		const int s = g->shard_of[i]; const size_t j = (size_t)g->local_of[i];
		const double *hA = &sA[s][j * n * n], *hE = &sE[s][j * 3 * n], *hA0 = &sA0[s][j * n];
		ll += sLL[s][j]; // em.c:48
		for (int k = 0; k < n; ++k) { tA0[k] += hA0[k]; for (int l = 0; l < n; ++l) tA[(size_t)k * n + l] += hA[(size_t)k * n + l]; }
		for (int b = 0; b < 2; ++b) for (int l = 0; l < n; ++l) tE[(size_t)b * n + l] += hE[(size_t)b * n + l];
		if (chk) chk[i] = sC[s][j];
	}
	if (A) memcpy(A, tA.data(), sizeof(double) * n * n);
	if (E) memcpy(E, tE.data(), sizeof(double) * 2 * n);
	if (A0) memcpy(A0, tA0.data(), sizeof(double) * n);
	if (LL) *LL = ll;
	g->last_reduce = 3;
	return PSMC_HIP_OK;
}

extern "C" int psmc_hip_group_estep_factored(psmc_hip_group *g, const double *a, const double *e, const double *a0, double *sums,
                                             double *E, double *LL)
{
	if (!g || !a || !e || !a0) return gfail(g, PSMC_HIP_EINVAL, "group_estep_factored: bad argument");
	if (g->mode != PSMC_HIP_MODE_FAST) return gfail(g, PSMC_HIP_ENOTSUP, "group_estep_factored: fast mode only");
	if (g->n_seg < 1) return gfail(g, PSMC_HIP_ESTATE, "group_estep_factored: no segments loaded");
	const int n = g->n;
	int rc = for_shards(g, [&](int s) { return psmc_hip_estep_factored_device(g->sh[s], a, e, a0, g->d_stats[s], g->st[s]); });
	if (rc) return rc;
	std::vector<double> v;
	if ((rc = reduce_vectors(g, (size_t)7 * n + 1, v, live_shards(g)))) return rc;
	if (sums) memcpy(sums, v.data(), sizeof(double) * 5 * n);
	if (E) memcpy(E, v.data() + (size_t)5 * n, sizeof(double) * 2 * n);
	if (LL) *LL = v[(size_t)7 * n];
	return PSMC_HIP_OK;
}

// after a group E-step the tables of segment i live on its shard: route the per-segment readers there
extern "C" int psmc_hip_group_route(psmc_hip_group *g, int seg, psmc_hip_ctx **ctx, int *local_seg)
{
	if (!g || seg < 0 || seg >= g->n_seg || !ctx || !local_seg) return gfail(g, PSMC_HIP_EINVAL, "group_route: bad argument");
	*ctx = g->sh[g->shard_of[seg]]; *local_seg = g->local_of[seg];
	return PSMC_HIP_OK;
}
