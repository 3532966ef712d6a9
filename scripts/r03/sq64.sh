cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/pmc
BARGS="--cpu-sample 0 --exact-extra 0 --n128-extra 0 --boot-extra 0 --shard-extra 0 --group-extra 0"
CMD="python $R/bench.py --steps 2 --warmup 1 $BARGS"
echo "rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU -- python bench.py --steps 2 --warmup 1 $BARGS" > $R/gpurun_out/pmc/n64_SQ.cmd
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d $R/gpurun_out/pmc -o n64_SQ -- $CMD > $R/gpurun_out/pmc/n64_SQ.json 2> $R/gpurun_out/pmc/n64_SQ.err; echo "sq n64 rc=$?"
cd $R; python scripts/sq_summary.py r03 n64
ls -la gpurun_out/pmc/n64_SQ*
