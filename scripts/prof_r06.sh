# Round-6 profiling recipe (run on the GPU box via gpurun).  Kernel-trace stats first, then PMC passes in their own runs (never
# combined with sys/runtime/hip tracing).  scripts/summarize_profiles_r06.py turns the raw outputs into profiles/r06_*.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_r06
rm -rf $OUT && mkdir -p $OUT
B="python bench.py --no-cpu-baseline"
# (1) the strict configuration: one forward at a time (the engine's default schedule for a forward alone)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/single -o s -- $B --sequential --steps 8 --warmup 3 --no-phase-a > $OUT/single.log 2>&1
# (1b) the same without the chunk overlap: every scan is ONE whole-sequence launch (the durations the roofline figures use)
SFSN_OVERLAP_CHUNKS=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/single_whole -o s -- $B --sequential --steps 8 --warmup 3 --no-phase-a > $OUT/single_whole.log 2>&1
# (2) the default command (phases S, K and the timed region with 12 forwards in flight)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/default -o d -- $B > $OUT/default.log 2>&1
# (3) PMC passes, whole-sequence launches, one forward at a time
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE"; do
  tag=$(echo $c | cut -d' ' -f1)
  SFSN_OVERLAP_CHUNKS=0 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$tag -o p -- $B --sequential --steps 2 --warmup 1 --no-phase-a > $OUT/pmc_$tag.log 2>&1
done
# (4) HBM traffic of one forward in the timed region's geometry (full-band stack at 8, sub-band scans at 16 rows per workgroup, fused input products)
for c in FETCH_SIZE WRITE_SIZE; do
  SFSN_OVERLAP_CHUNKS=0 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/job_$c -o p -- $B --sequential --rpw 8,16 --steps 3 --warmup 1 --no-phase-a > $OUT/job_$c.log 2>&1
done
# (4b) the mode the live recipe runs -- no fp32 spike tensors (bench.py --no-layer-outputs = layer_outputs "none"; "counts" moves the same bytes):
#      HBM traffic of one forward in the strict schedule and in the timed region's geometry
for c in FETCH_SIZE WRITE_SIZE; do
  SFSN_OVERLAP_CHUNKS=0 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/nl_$c -o p -- $B --sequential --no-layer-outputs --steps 2 --warmup 1 --no-phase-a > $OUT/nl_$c.log 2>&1
  SFSN_OVERLAP_CHUNKS=0 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/nljob_$c -o p -- $B --sequential --no-layer-outputs --rpw 8,16 --steps 3 --warmup 1 --no-phase-a > $OUT/nljob_$c.log 2>&1
done
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python bench.py --streaming --no-cpu-baseline > $OUT/bench_streaming.json 2>> $OUT/bench_default.err
python bench.py --streaming --waveform --host-io > $OUT/bench_streaming_waveform_host.json 2>> $OUT/bench_default.err
# (5) the training step: the line at the recipe's batch, kernel stats of one step at B = 16
python bench.py --training --batch 64 > $OUT/bench_training.json 2>> $OUT/bench_default.err
python bench.py --training --batch 16 > $OUT/bench_training_b16.json 2>> $OUT/bench_default.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/training -o t -- python bench.py --training --batch 16 --steps 2 --warmup 1 > $OUT/training.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/training64 -o t -- python bench.py --training --batch 64 --steps 2 --warmup 1 > $OUT/training64.log 2>&1
# what goes back is capped at 64 MiB: the big per-dispatch traces are not needed (the stats files are)
rm -f $OUT/default/d_kernel_trace.csv $OUT/training/t_kernel_trace.csv $OUT/training64/t_kernel_trace.csv $OUT/single_whole/s_kernel_trace.csv
python -c "from spiking_fullsubnet_amd import _lib; print(_lib.source_hash())" > $OUT/source_hash.txt
tail -c 600 $OUT/bench_default.json
