# phase ledger of input_proj_bf3_kernel: build sfsn_kernels.hip alone with stamps, run it (GPU box)
set -e
cd "$(dirname "$0")"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I../../include -I../../spiking_fullsubnet_amd/csrc -DIP_STAMPS $EXTRA -shared -o libip_stamps.so ../../spiking_fullsubnet_amd/csrc/sfsn_kernels.hip
python inproj_stamps.py
