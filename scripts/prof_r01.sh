# Round-1 profiling recipe (run on the GPU box via gpurun).  Kernel-trace stats first, then PMC passes in their
# own runs (never combined with sys/runtime/hip tracing).  Summaries are copied into profiles/ by hand
# (scripts/summarize_profiles.py).
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_r01
rm -rf $OUT && mkdir -p $OUT
# (1) the command whose dominant-kernel duration bench.py reports as `roofline` (one forward at a time)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/single -o s -- python bench.py --inflight 1 --steps 8 --warmup 3 --no-cpu-baseline --no-saturated > $OUT/single.log 2>&1
# (2) the default command (phase A single stream + phase B six forwards in flight)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/default -o d -- python bench.py --no-cpu-baseline > $OUT/default.log 2>&1
# (3) PMC passes on the single-stream command
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE"; do
  tag=$(echo $c | cut -d' ' -f1)
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$tag -o p -- python bench.py --inflight 1 --steps 2 --warmup 1 --no-cpu-baseline --no-saturated > $OUT/pmc_$tag.log 2>&1
done
# (4) HBM traffic of one forward as the timed region runs it (sub-band scans at 16 rows per workgroup, fused-input layer 2)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/job_$c -o p -- python bench.py --no-phase-a --inflight 1 --rpw 4,16 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/job_$c.log 2>&1
done
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -1 $OUT/bench_default.json
