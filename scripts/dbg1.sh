PSMC_HIP_DEBUG_FLAGGED=1 timeout 600 python scripts/estep_trace.py 8 "$1" 2> gpurun_out/flag.err | cut -c1-170
grep "flagged fwd" gpurun_out/flag.err | tail -4 | cut -c1-1500
