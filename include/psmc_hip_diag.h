/* psmc_hip_diag.h -- C-ABI of libpsmc_hip_diag.so: the lab bench of the MI355X PSMC E-step (device self-test,
 * instruction microbenchmarks, pipe / placement / HBM probes, per-kernel HIP-event timing of the last E-step).
 *
 * NOT part of the drop-in boundary: a maintainer of lh3/psmc links libpsmc_hip.so (include/psmc_hip.h) only.  This
 * library exists for tests/, bench.py's roofline figures and the measurements DESIGN.md cites; it links against
 * libpsmc_hip.so and is built from the same tree (psmc_amd/csrc/Makefile), so it may look inside a context.
 * Every function returns 0 (or what it documents) or a negative PSMC_HIP_E* code of psmc_hip.h.
 */
#ifndef PSMC_HIP_DIAG_H
#define PSMC_HIP_DIAG_H
#include "psmc_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Built-in check of the cross-lane primitives on the device (row replication
 * variants, DPP broadcasts, f64 MFMA layout).  Returns 0 when all agree;
 * a positive bitmask of failed primitives otherwise. */
int psmc_hip_selftest(int device);

/* Diagnostic: shader cycles per operation of the FP64 building blocks (dependent
 * and independent v_fmac_f64_dpp chains, row replication, f64 MFMA ...), one
 * wave; see psmc_amd/csrc/microbench.hip for the meaning of out[0..13]. */
int psmc_hip_microbench(int device, double *out, int n);

/* Diagnostic: do the f64 matrix instructions of one wave overlap with the f64 vector instructions of another wave
 * on the same SIMD?  One work-group on one CU, waves go to its four SIMDs round robin; a "matrix wave" issues 64
 * v_mfma_f64_16x16x4 per round, a "vector wave" 1024 v_fma_f64 (8 chains) -- ~4100 cycles of issue either way.
 * out[8*c + w] = shader cycles per round of wave w in configuration c (0 where the configuration has no wave w):
 *   c=0: 4 matrix waves (one per SIMD)      c=1: 4 vector waves         c=2: 8 matrix waves (two per SIMD)
 *   c=3: 8 vector waves                     c=4: waves 0-3 matrix, 4-7 vector (one of each per SIMD)
 *   c=5: even waves matrix, odd waves vector (SIMDs 0 and 2 hold two matrix waves, 1 and 3 two vector waves).
 * Separate pipes would give c=4 the times of c=0 / c=1; one shared pipe gives it their sum.  n >= 48. */
#define PSMC_HIP_PIPE_PROBE_CONFIGS 6
int psmc_hip_pipe_probe(int device, double *out, int n);

/* Diagnostic, second edition of the pipe probe: which instructions of one wave overlap with another wave's
 * v_mfma_f64 on the same SIMD?  One work-group of up to 8 waves on one CU (wave w -> SIMD w % 4); kinds8[w] says what
 * wave w issues per round (~4 k cycles of issue when alone): 0 idle, 1 v_mfma_f64_16x16x4 x 64, 2 v_fma_f64 x 1024,
 * 3 v_mov_b32_dpp x 1024, 4 DPP scan levels (2 v_mov_b32_dpp + v_add_f64, as the sweeps' row scans) ~ 1024 in all,
 * 5 ds_read_b128 x 512, 6 s_load_dwordx4 x 256 + v_readlane_b32 x 512, 7 v_add_u32 x 1024, 8 v_fma_f32 x 1024,
 * 9 v_add_f64 x 1024.  out8[w] = shader cycles per round of wave w (0 for idle waves). */
int psmc_hip_pipe_probe2(int device, const int *kinds8, int rounds, double *out8);

/* Diagnostic: where do the waves of a launch smaller than the device land?  n_kernels (1..4) launches of n_waves waves
 * of the structured sweep step (no memory traffic), in work-groups of waves_per_block (1..4) waves, side by side on
 * streams of their own.  out[3*(k*n_waves_padded + w) + 0..2] = shader cycles per step of wave w of launch k, its
 * HW_ID register (SIMD bits 5:4, CU 11:8, SH 12, SE 15:13) and its XCC_ID; n_waves_padded = n_waves rounded up to a
 * multiple of waves_per_block.  *ms_out = the slowest launch.  A shard-sized E-step has fewer waves than the device
 * has SIMDs: if they are stacked on the same SIMDs, every step costs a multiple of its latency. */
int psmc_hip_place_probe(int device, int n_waves, int waves_per_block, int n_kernels, int steps, double *out, double *ms_out);

/* Diagnostic: an 8-byte-per-lane streaming copy (reads and writes 8*n_doubles bytes, 5
 * launches) to calibrate the rocprofv3 FETCH_SIZE / WRITE_SIZE counters for the access
 * width the kernels use; *ms_out = average duration of one launch. */
int psmc_hip_stream_probe(int device, long long n_doubles, double *ms_out);

/* Diagnostic: what plain streaming kernels reach on this device with 16-byte accesses over two
 * buffers of `bytes` each: gbps_out[0] fill, [1] read, [2] copy (read + write), [3] the store
 * pattern of the structured sweeps (four 512-byte-per-step streams per wave).  GB/s. */
int psmc_hip_hbm_probe(int device, long long bytes, double *gbps_out);

/* Diagnostic: the structured sweep step (no memory traffic) on n_waves wavefronts at once, `steps`
 * steps each: out[0] kernel ms, [1] mean / [2] max shader cycles per step of a wave, [3] mean shader
 * clock in MHz the waves saw -- how far FP64 issue and clocks hold up when the whole device is busy.
 */
int psmc_hip_load_probe(int device, int n_waves, int steps, double *out);

/* Diagnostic: psmc_hip_load_probe with the table stores of a forward sweep: each wave appends 512 bytes per tile and
 * step during the last `store_steps` of its `steps` steps.  mode 1: two 16-byte stores per lane and step (what
 * k_fwd_struct does), 2: the same bytes written as 2 KB per tile every 4th step.  out[0..3] as psmc_hip_load_probe,
 * out[4] = GB/s of the stores over the whole kernel. */
int psmc_hip_load_probe_st(int device, int n_waves, int steps, int store_steps, int mode, double *out);

/* Wall time in ms of the last E-step measured with HIP events on the streams the
 * kernels ran on.  Exact mode: [0] total, [1] forward, [2] backward, [3] expect,
 * [4] host-copy tail.  Fast mode: [0] total, [1] both sweep chains (speculate +
 * repair rounds; forward and backward run concurrently), [2] LL + redo of the
 * counts after the chains, [3] the full expect kernel alone (fused back half: its one or two
 * launches, summed), [4] reductions, [5] the
 * speculative forward sweep kernel alone, [6] the speculative backward sweep alone. */
int psmc_hip_last_timing(psmc_hip_ctx *ctx, double ms[7]);

/* Diagnostic: do HIP's per-stream compute-unit masks (hipExtStreamCreateWithCUMask) partition the device?  Stream A gets
 * the first n_cus_a compute units of the mask's bit order, stream B the others; n_waves_a / n_waves_b one-wave work-groups of
 * the structured sweep step (no memory traffic) run on them at the same time.  out[3*w + 0..2] = shader cycles per step,
 * HW_ID and XCC_ID of wave w (A's waves first, then B's, as psmc_hip_place_probe); *ms_out = the slower launch.
 * n_cus_a = 0: no masks (both streams see the whole device).  psmc_boot keeps its main run out of the bootstrap
 * batch's way like this (psmc_amd/host/boot.c). */
int psmc_hip_cumask_probe(int device, int n_cus_a, int n_waves_a, int n_waves_b, int steps, double *out, double *ms_out);

#ifdef __cplusplus
}
#endif
#endif
