cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
echo base $(python tools/bench_bilateral.py 2>&1 | tail -1)
echo doubles $(TSDF_HIP_LIB=$GRAFT_REPO_ROOT/build/variants/bild/libtsdf_hip.so python tools/bench_bilateral.py 2>&1 | tail -1)
done
