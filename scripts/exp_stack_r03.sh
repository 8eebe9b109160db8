# round 3: the wide stack (IO-wave scans + PROJ roles with a loader wave), H=224, B=64 rows; SFSN_STACK_EXP = wrong-result timing experiments
# (the switches are compiled in only with: make -C spiking_fullsubnet_amd/csrc clean all EXTRA=-DSFSN_EXPERIMENTS)
cd /root/repo
export ROWS=128 SFSN_STACK_DEBUG=1
SFSN_STACK_EXP=32 python scripts/exp_stack_direct.py 224 2 8 16
SFSN_STACK_EXP=96 python scripts/exp_stack_direct.py 224 2 8 16
SFSN_STACK_EXP=97 python scripts/exp_stack_direct.py 224 2 8 16
SFSN_STACK_EXP=33 python scripts/exp_stack_direct.py 224 2 8 16
