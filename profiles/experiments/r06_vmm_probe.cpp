#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("FAIL %s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void fill(double *p, size_t n, double v) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; for (; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v; }
__global__ void sum(const double *p, size_t n, double *out) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; double s = 0; for (; i < n; i += (size_t)gridDim.x * blockDim.x) s += p[i]; atomicAdd(out, s); }
int main(int argc, char **argv) {
	size_t gb1 = argc > 1 ? atol(argv[1]) : 4, gb2 = argc > 2 ? atol(argv[2]) : 2;
	CK(hipSetDevice(0));
	hipMemAllocationProp prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
	size_t gran = 0; CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
	printf("granularity %zu\n", gran);
	size_t s1 = gb1 << 30, s2 = gb2 << 30, va = s1 + s2;
	void *base = nullptr; double t0 = now();
	CK(hipMemAddressReserve(&base, va, (size_t)1 << 30, nullptr, 0)); printf("base %p\n", base);
	hipMemGenericAllocationHandle_t h1, h2;
	double t1 = now(); CK(hipMemCreate(&h1, s1, &prop, 0)); double t2 = now();
	CK(hipMemMap(base, s1, 0, h1, 0));
	hipMemAccessDesc ad = {}; ad.location = prop.location; ad.flags = hipMemAccessFlagsProtReadWrite;
	CK(hipMemSetAccess(base, s1, &ad, 1)); double t3 = now();
	printf("reserve %.3f s, create %zu GB %.3f s, map+access %.3f s\n", t1 - t0, gb1, t2 - t1, t3 - t2);
	double *out; CK(hipMalloc(&out, 8)); CK(hipMemset(out, 0, 8));
	fill<<<4096, 256>>>((double *)base, s1 / 8, 1.0); CK(hipDeviceSynchronize());
	double t4 = now(); CK(hipMemCreate(&h2, s2, &prop, 0)); CK(hipMemMap((char *)base + s1, s2, 0, h2, 0)); CK(hipMemSetAccess((char *)base + s1, s2, &ad, 1)); double t5 = now();
	printf("grow by %zu GB: %.3f s\n", gb2, t5 - t4);
	fill<<<4096, 256>>>((double *)base + s1 / 8, s2 / 8, 2.0);
	sum<<<4096, 256>>>((const double *)base, va / 8, out); CK(hipDeviceSynchronize());
	double h; CK(hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost));
	printf("sum %.0f expected %.0f\n", h, (double)(s1 / 8) + 2.0 * (s2 / 8));
	double t6 = now(); void *plain; CK(hipMalloc(&plain, s2)); double t7 = now(); printf("plain hipMalloc %zu GB %.3f s\n", gb2, t7 - t6);
	CK(hipMemUnmap(base, s1)); CK(hipMemUnmap((char *)base + s1, s2)); CK(hipMemRelease(h1)); CK(hipMemRelease(h2)); CK(hipMemAddressFree(base, va));
	printf("ok\n"); return 0;
}
