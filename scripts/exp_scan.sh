# Bottleneck attribution for the scan kernels (wrong-result debug variants, built with -DSFSN_TIMING_EXPERIMENTS).
cd $GRAFT_REPO_ROOT/spiking_fullsubnet_amd/csrc && cp libsfsn_hip.so /tmp/keep.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I../../include -DSFSN_TIMING_EXPERIMENTS -shared -o libsfsn_hip.so sfsn_kernels.hip sfsn_fft.hip sfsn_pack.cpp
cd $GRAFT_REPO_ROOT
for r in 16 8 4; do
for v in 3 51; do
  SFSN_SCAN_RPW=$r SFSN_SCAN_DEBUG_OUT=$v python bench.py --steps 4 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('RPW=$r OUT=$v', 'sb', d['roofline']['launch_ms'], 'fb', d['roofline']['other_kernels_ms']['scan:fb'], 'total', d['ms_per_step'])"
done; done
cp /tmp/keep.so spiking_fullsubnet_amd/csrc/libsfsn_hip.so
