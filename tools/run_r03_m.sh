# GPU box, round 3 step M: timing-only bounds of the reduce pass (no record loop / no Adam phase)
set -u
R=$GRAFT_REPO_ROOT
cd $R
L=latent-nerf-test_amd/lib
bash tools/ab_kernels.sh $L/liblnerf_hip.so $L/liblnerf_hip_norec.so $L/liblnerf_hip_noadam.so > gpurun_out/r03m_abk.txt 2>&1; cat gpurun_out/r03m_abk.txt | grep -v "k_mlp\|k_grid_forward"
exit 0
