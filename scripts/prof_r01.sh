# Profiling recipe for round 1 (run on the GPU box via gpurun): kernel-trace stats, then PMC passes (separate runs).
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_r01
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o r1 -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/trace.log 2>&1
rocprofv3 -L > $OUT/counters.txt 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc1 -o p1 -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $OUT/pmc2 -o p2 -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/pmc2.log 2>&1
find $OUT -name "*.csv" | head -20
