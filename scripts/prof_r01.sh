# Round-1 profiling recipe (run on the GPU box via gpurun).  Kernel-trace stats first, then PMC passes in their
# own runs (never combined with sys/runtime/hip tracing).  Summaries are copied into profiles/ by hand.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_r01
rm -rf $OUT && mkdir -p $OUT
CMD="python bench.py --steps 5 --warmup 2 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o r01 -- $CMD > $OUT/trace.log 2>&1
tail -1 $OUT/trace.log > $OUT/bench_line_under_trace.json
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o f -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o w -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_sq1 -o p1 -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/pmc_sq1.log 2>&1
rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq2 -o p2 -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/pmc_sq2.log 2>&1
python bench.py --steps 20 --warmup 3 > $OUT/bench_full.json 2> $OUT/bench_full.err
tail -1 $OUT/bench_full.json
