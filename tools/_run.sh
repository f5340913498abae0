cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_parity_integrate.py tests/test_weight_storage.py -m gpu -x -q 2>&1 | tail -2
for i in 1 2 3 4; do
echo new $(python tools/dbg_integrate_only.py 2>&1 | tail -1)
echo old $(TSDF_HIP_LIB=$GRAFT_REPO_ROOT/build/variants/intold/libtsdf_hip.so python tools/dbg_integrate_only.py 2>&1 | tail -1)
done
