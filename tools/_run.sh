cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
echo new $(python tools/dbg_ray_only.py 40 2>&1 | tail -1)
echo old $(TSDF_HIP_LIB=$GRAFT_REPO_ROOT/build/variants/rayold/libtsdf_hip.so python tools/dbg_ray_only.py 40 2>&1 | tail -1)
done | tee gpurun_out/r04t/ray_ab.txt
