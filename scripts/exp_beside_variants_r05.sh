cd $GRAFT_REPO_ROOT
for v in ${VARIANTS:-product dp8 dp8pf6 dg6}; do
  echo "=== $v"
  if [ $v = product ]; then NO_EXP=1 timeout 300 python scripts/exp_beside_r05.py 2>&1 | grep -v "^recorded\|amdgpu.ids"
  else SFSN_LIB_PATH=$GRAFT_REPO_ROOT/spiking_fullsubnet_amd/csrc_$v/libsfsn_hip.so timeout 300 python scripts/exp_beside_r05.py 2>&1 | grep -v "^recorded\|amdgpu.ids"; fi
done
echo "=== stall ledger (EXPERIMENTS build of the product's sources)"
timeout 300 python scripts/exp_beside_r05.py 2>&1 | grep -v "amdgpu.ids"
