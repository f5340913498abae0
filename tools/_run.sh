cd $GRAFT_REPO_ROOT
for i in 1 2 3; do for n in w6 w7; do
echo $n $(TSDF_HIP_LIB=$GRAFT_REPO_ROOT/build/variants/$n/libtsdf_hip.so python tools/dbg_integrate_only.py 2>&1 | tail -1)
done; done
