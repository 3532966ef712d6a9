// stub_rccl.hip -- TEST INFRASTRUCTURE: a single-process stand-in for librccl that lets the multi-shard RCCL branch of
// psmc_amd/csrc/group.hip (reduce_vectors: ncclCommInitAll, grouped in-place ncclAllReduce on each shard's own stream) run
// on a box with ONE GPU, where real RCCL refuses a communicator over a repeated device.  group.hip loads it instead of
// librccl when PSMC_HIP_RCCL_LIB points here and the group option "rccl" is 2 (tests/test_gpu_estep.py).
//
// What it reproduces is the part that can go wrong in the caller: STREAM ORDER.  Every rank's all-reduce is enqueued on the
// stream the caller passes; the sum is taken only after every rank's stream has reached its call, and no rank's stream goes
// on before its receive buffer holds the result -- exactly the guarantees of a real grouped all-reduce.  It is not a model of
// RCCL's transport, topology or performance.  Own code; the six entry points have the signatures of <rccl/rccl.h>.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <mutex>
#include <vector>

namespace {
struct World;
struct Call { const double *send; double *recv; size_t count; hipStream_t stream; int rank; };
struct World { int n = 0; std::vector<int> dev; std::vector<Call> calls; double *tmp = nullptr; size_t tmp_cap = 0; int live = 0; };
std::mutex g_mu;
int g_depth = 0;
std::vector<World *> g_pending; // worlds with calls recorded inside the open group
__global__ void k_sum(const double *const *src, int n_src, double *out, size_t count)
{
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= count) return;
	double s = src[0][i];
	for (int r = 1; r < n_src; ++r) s += src[r][i]; // rank order
	out[i] = s;
}
__global__ void k_bcast(const double *in, double *const *dst, int n_dst, size_t count)
{
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= count) return;
	for (int r = 0; r < n_dst; ++r) dst[r][i] = in[i];
}
ncclResult_t flush(World *w)
{
	if ((int)w->calls.size() != w->n) { w->calls.clear(); return ncclInvalidUsage; } // every rank of the communicator must take part
	const size_t count = w->calls[0].count;
	for (const Call &c : w->calls) if (c.count != count) { w->calls.clear(); return ncclInvalidArgument; }
	const int n = w->n;
	hipStream_t s0 = w->calls[0].stream;
	if (hipSetDevice(w->dev[w->calls[0].rank]) != hipSuccess) return ncclUnhandledCudaError;
	if (w->tmp_cap < count) { if (w->tmp) (void)hipFree(w->tmp); if (hipMalloc((void **)&w->tmp, sizeof(double) * count) != hipSuccess) return ncclUnhandledCudaError; w->tmp_cap = count; }
	// pointer tables (host-pinned, read by the kernels): freed after the stream has passed them
	const double **src; double **dst;
	if (hipHostMalloc((void **)&src, sizeof(double *) * n, hipHostMallocDefault) != hipSuccess || hipHostMalloc((void **)&dst, sizeof(double *) * n, hipHostMallocDefault) != hipSuccess) return ncclUnhandledCudaError;
	std::vector<hipEvent_t> ev(n);
	for (int r = 0; r < n; ++r) {
		src[r] = w->calls[r].send; dst[r] = w->calls[r].recv;
		if (hipEventCreateWithFlags(&ev[r], hipEventDisableTiming) != hipSuccess) return ncclUnhandledCudaError;
		(void)hipEventRecord(ev[r], w->calls[r].stream);          // rank r's stream has reached its call ...
		if (r > 0) (void)hipStreamWaitEvent(s0, ev[r], 0);          // ... before the sum is taken (on rank 0's stream)
	}
	const unsigned blocks = (unsigned)((count + 255) / 256);
	hipLaunchKernelGGL(k_sum, dim3(blocks), dim3(256), 0, s0, (const double *const *)src, n, w->tmp, count);
	hipLaunchKernelGGL(k_bcast, dim3(blocks), dim3(256), 0, s0, (const double *)w->tmp, (double *const *)dst, n, count);
	hipEvent_t done;
	if (hipEventCreateWithFlags(&done, hipEventDisableTiming) != hipSuccess) return ncclUnhandledCudaError;
	(void)hipEventRecord(done, s0);
	for (int r = 1; r < n; ++r) (void)hipStreamWaitEvent(w->calls[r].stream, done, 0); // no rank goes on before its buffer holds the sum
	// the events and tables may be released once the work is enqueued (HIP defers the destruction of recorded events); the pinned
	// tables must outlive the kernels: wait for rank 0's stream here -- a stub may block where the real library would not
	(void)hipStreamSynchronize(s0);
	for (int r = 0; r < n; ++r) (void)hipEventDestroy(ev[r]);
	(void)hipEventDestroy(done);
	(void)hipHostFree(src); (void)hipHostFree(dst);
	w->calls.clear();
	return hipGetLastError() == hipSuccess ? ncclSuccess : ncclUnhandledCudaError;
}
struct Comm { World *w; int rank; };
} // namespace

extern "C" {
ncclResult_t ncclCommInitAll(ncclComm_t *comm, int ndev, const int *devlist)
{
	if (!comm || ndev < 1) return ncclInvalidArgument;
	World *w = new World(); w->n = ndev; w->live = ndev;
	for (int i = 0; i < ndev; ++i) w->dev.push_back(devlist ? devlist[i] : i); // a device may repeat: that is the point of this stub
	for (int i = 0; i < ndev; ++i) comm[i] = (ncclComm_t) new Comm{w, i};
	return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
	Comm *c = (Comm *)comm;
	if (!c) return ncclInvalidArgument;
	std::lock_guard<std::mutex> lk(g_mu);
	if (--c->w->live == 0) { if (c->w->tmp) (void)hipFree(c->w->tmp); delete c->w; }
	delete c;
	return ncclSuccess;
}
ncclResult_t ncclGroupStart() { std::lock_guard<std::mutex> lk(g_mu); ++g_depth; return ncclSuccess; }
ncclResult_t ncclGroupEnd()
{
	std::lock_guard<std::mutex> lk(g_mu);
	if (g_depth <= 0) return ncclInvalidUsage;
	if (--g_depth > 0) return ncclSuccess;
	ncclResult_t r = ncclSuccess;
	for (World *w : g_pending) { const ncclResult_t q = flush(w); if (q != ncclSuccess) r = q; }
	g_pending.clear();
	return r;
}
ncclResult_t ncclAllReduce(const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op, ncclComm_t comm, hipStream_t stream)
{
	Comm *c = (Comm *)comm;
	if (!c || !sendbuff || !recvbuff) return ncclInvalidArgument;
	if (datatype != ncclDouble || op != ncclSum) return ncclInvalidArgument; // all group.hip asks for
	std::lock_guard<std::mutex> lk(g_mu);
	World *w = c->w;
	w->calls.push_back(Call{(const double *)sendbuff, (double *)recvbuff, count, stream, c->rank});
	if (g_depth > 0) { bool seen = false; for (World *p : g_pending) seen = seen || p == w; if (!seen) g_pending.push_back(w); return ncclSuccess; }
	return (int)w->calls.size() == w->n ? flush(w) : ncclSuccess; // ungrouped: the last rank's call completes the collective
}
const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : (r == ncclInvalidArgument ? "invalid argument (stub)" : (r == ncclInvalidUsage ? "invalid usage (stub)" : "unhandled HIP error (stub)")); }
}
