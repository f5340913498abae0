cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_parity_integrate.py -m gpu -x -q 2>&1 | tail -2
bash tools/stats_cmd.sh r05c "python $GRAFT_REPO_ROOT/tools/dbg_integrate_only.py" 8 | grep "cull\|integrate_packed"
TSDF_HIP_LIB=$GRAFT_REPO_ROOT/build/variants/bilold/libtsdf_hip.so bash tools/stats_cmd.sh r05c_old "python $GRAFT_REPO_ROOT/tools/dbg_integrate_only.py" 8 | grep "cull\|integrate_packed"
