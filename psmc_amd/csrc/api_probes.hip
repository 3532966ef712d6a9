// api_probes.hip -- libpsmc_hip_diag.so (include/psmc_hip_diag.h): device self-test, instruction microbenchmarks, the pipe /
// placement / compute-unit-mask / HBM probes that DESIGN.md cites, and the HIP-event timing of a context's last E-step.
// Not part of an E-step and not part of the drop-in library: it links against libpsmc_hip.so.
#include "psmc_hip_ctx.h"
#include "psmc_hip_diag.h"

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

// Two streams with complementary compute-unit masks (or none), the placement probe on both at once.
extern "C" int psmc_hip_cumask_probe(int device, int n_cus_a, int n_waves_a, int n_waves_b, int steps, double *out, double *ms_out)
{
	int nd = psmc_hip_device_count();
	if (n_cus_a < 0 || n_waves_a < 1 || n_waves_b < 1 || n_waves_a > (1 << 16) || n_waves_b > (1 << 16) || steps < 4 || !out) return PSMC_HIP_EINVAL;
	if (nd <= 0 || device < 0 || device >= nd) return PSMC_HIP_EDEVICE;
	if (hipSetDevice(device) != hipSuccess) return PSMC_HIP_EDEVICE;
	int cus = 0;
	if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus <= 0) return PSMC_HIP_EDEVICE;
	if (n_cus_a >= cus) return PSMC_HIP_EINVAL;
	double *d = nullptr;
	if (hipMalloc((void **)&d, sizeof(double) * 3 * (size_t)(n_waves_a + n_waves_b)) != hipSuccess) return PSMC_HIP_ENOMEM;
	hipStream_t st[2] = {nullptr, nullptr};
	int rc = 0;
	if (n_cus_a > 0) {
		const int words = (cus + 31) / 32;
		std::vector<uint32_t> ma(words, 0u), mb(words, 0u);
		for (int i = 0; i < cus; ++i) (i < n_cus_a ? ma : mb)[i >> 5] |= 1u << (i & 31);
		if (hipExtStreamCreateWithCUMask(&st[0], (uint32_t)words, ma.data()) != hipSuccess) rc = 1;
		if (rc == 0 && hipExtStreamCreateWithCUMask(&st[1], (uint32_t)words, mb.data()) != hipSuccess) rc = 1;
	} else {
		if (hipStreamCreateWithFlags(&st[0], hipStreamNonBlocking) != hipSuccess) rc = 1;
		if (rc == 0 && hipStreamCreateWithFlags(&st[1], hipStreamNonBlocking) != hipSuccess) rc = 1;
	}
	hipEvent_t e0, e1[2];
	(void)hipEventCreate(&e0); (void)hipEventCreate(&e1[0]); (void)hipEventCreate(&e1[1]);
	for (int pass = 0; pass < 2 && rc == 0; ++pass) { // pass 0 warms clocks and the instruction cache
		(void)hipDeviceSynchronize();
		(void)hipEventRecord(e0, st[0]);
		(void)hipStreamWaitEvent(st[1], e0, 0);
		rc = run_place_probe(st[0], d, n_waves_a, 1, steps & ~3); (void)hipEventRecord(e1[0], st[0]);
		if (rc == 0) { rc = run_place_probe(st[1], d + 3 * (size_t)n_waves_a, n_waves_b, 1, steps & ~3); (void)hipEventRecord(e1[1], st[1]); }
	}
	if (rc == 0 && hipDeviceSynchronize() == hipSuccess && hipMemcpy(out, d, sizeof(double) * 3 * (size_t)(n_waves_a + n_waves_b), hipMemcpyDeviceToHost) == hipSuccess) {
		float a = 0, b = 0;
		(void)hipEventElapsedTime(&a, e0, e1[0]); (void)hipEventElapsedTime(&b, e0, e1[1]);
		if (ms_out) *ms_out = std::max(a, b);
	} else rc = 1;
	(void)hipEventDestroy(e0); (void)hipEventDestroy(e1[0]); (void)hipEventDestroy(e1[1]);
	for (int k = 0; k < 2; ++k) if (st[k]) (void)hipStreamDestroy(st[k]);
	(void)hipFree(d);
	return rc ? PSMC_HIP_EDEVICE : PSMC_HIP_OK;
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
