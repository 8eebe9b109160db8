# a build of the library BESIDE the product's (git-ignored; the sources are copied: the product's objects stay untouched):
#   scripts/build_exp_lib.sh                      -> spiking_fullsubnet_amd/csrc_exp/   with -DSFSN_EXPERIMENTS (per-workgroup stamps,
#                                                    per-wave stall counters, timing switches; scripts/exp_wgtimes_r05.py, exp_beside_r05.py)
#   scripts/build_exp_lib.sh NAME "-DX=1 -DY=2"   -> spiking_fullsubnet_amd/csrc_NAME/  with those flags (kernel variants for A/B runs:
#                                                    SFSN_LIB_PATH=.../csrc_NAME/libsfsn_hip.so)
set -e
cd "$(dirname "$0")/.."
name=${1:-exp}
flags=${2--DSFSN_EXPERIMENTS}
rm -rf spiking_fullsubnet_amd/csrc_$name && mkdir spiking_fullsubnet_amd/csrc_$name
cp spiking_fullsubnet_amd/csrc/*.hip spiking_fullsubnet_amd/csrc/*.h spiking_fullsubnet_amd/csrc/*.cpp spiking_fullsubnet_amd/csrc/Makefile spiking_fullsubnet_amd/csrc_$name/
make -C spiking_fullsubnet_amd/csrc_$name -j8 EXTRA="$flags"
