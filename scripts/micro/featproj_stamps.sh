# phase ledger of featproj_kernel: build sfsn_featproj.hip alone with stamps, run it (GPU box)
set -e
cd "$(dirname "$0")"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I../../include -I../../spiking_fullsubnet_amd/csrc -DFP_STAMPS $EXTRA -shared -o libfp_stamps.so ../../spiking_fullsubnet_amd/csrc/sfsn_featproj.hip
python featproj_stamps.py
