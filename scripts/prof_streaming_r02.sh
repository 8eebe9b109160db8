# round 2: streaming lines (one launch per hop / graph replay) + kernel stats of the one-launch run (run on the GPU box)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_stream_r02
rm -rf $OUT && mkdir -p $OUT
python bench.py --streaming > $OUT/streaming_one_launch.json 2> $OUT/err1.log
python bench.py --streaming --no-one-launch > $OUT/streaming_graph.json 2> $OUT/err2.log
python bench.py --streaming --batch 4 --hop 2 > $OUT/streaming_b4_hop2.json 2> $OUT/err3.log
python bench.py --streaming --waveform > $OUT/streaming_waveform.json 2> $OUT/err4.log
python bench.py --streaming --waveform --host-io > $OUT/streaming_waveform_host.json 2> $OUT/err5.log
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -o s -- python bench.py --streaming --steps 300 --warmup 50 > $OUT/log.txt 2>&1
cat $OUT/streaming_one_launch.json $OUT/streaming_graph.json $OUT/streaming_b4_hop2.json $OUT/streaming_waveform.json $OUT/streaming_waveform_host.json | cut -c1-1200
head -5 $OUT/t/s_kernel_stats.csv | cut -c1-200
