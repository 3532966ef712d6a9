#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("FAIL %s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void fill(double *p, size_t n, double v) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; for (; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v; }
int main(int argc, char **argv) {
	size_t gb1 = argc > 1 ? atol(argv[1]) : 200, gb2 = argc > 2 ? atol(argv[2]) : 31;
	CK(hipSetDevice(0));
	void *a, *b, *c;
	double t0 = now(); CK(hipMalloc(&a, gb1 << 30)); double t1 = now(); CK(hipMalloc(&b, gb2 << 30)); double t2 = now();
	printf("first malloc %zu GB %.3f s, second %zu GB %.3f s\n", gb1, t1 - t0, gb2, t2 - t1);
	fill<<<8192, 256>>>((double *)a, (gb1 << 30) / 8, 1.0); fill<<<8192, 256>>>((double *)b, (gb2 << 30) / 8, 2.0); CK(hipDeviceSynchronize());
	double t3 = now(); CK(hipFree(a)); CK(hipFree(b)); double t4 = now();
	CK(hipMalloc(&c, (gb1 + gb2) << 30)); double t5 = now();
	fill<<<8192, 256>>>((double *)c, ((gb1 + gb2) << 30) / 8, 3.0); CK(hipDeviceSynchronize()); double t6 = now();
	printf("free both %.3f s, malloc %zu GB in the same process %.3f s, first touch (fill) %.3f s\n", t4 - t3, gb1 + gb2, t5 - t4, t6 - t5);
	CK(hipFree(c)); printf("ok\n"); return 0;
}
