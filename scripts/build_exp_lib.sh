# an EXPERIMENTS build of the library (per-workgroup stamps, timing switches) BESIDE the product's: spiking_fullsubnet_amd/csrc_exp/
# (git-ignored; scripts/exp_wgtimes_r05.py loads it when it exists).  The sources are copied: the product's objects stay untouched.
set -e
cd "$(dirname "$0")/.."
rm -rf spiking_fullsubnet_amd/csrc_exp && mkdir spiking_fullsubnet_amd/csrc_exp
cp spiking_fullsubnet_amd/csrc/*.hip spiking_fullsubnet_amd/csrc/*.h spiking_fullsubnet_amd/csrc/*.cpp spiking_fullsubnet_amd/csrc/Makefile spiking_fullsubnet_amd/csrc_exp/
make -C spiking_fullsubnet_amd/csrc_exp -j8 EXTRA=-DSFSN_EXPERIMENTS
