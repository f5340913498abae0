cd $GRAFT_REPO_ROOT
for r in 1 2; do for n in base prologue noblend nomark nofallback; do
echo $n $(TSDF_HIP_LIB=$GRAFT_REPO_ROOT/build/variants/d_$n/libtsdf_hip.so python tools/dbg_integrate_only.py 2>&1 | tail -1)
done; done
for n in base prologue noblend nomark nofallback; do
export TSDF_HIP_LIB=$GRAFT_REPO_ROOT/build/variants/d_$n/libtsdf_hip.so
echo "#### $n"
bash tools/pmc_cmd.sh d_$n "python $GRAFT_REPO_ROOT/tools/dbg_integrate_only.py" "integrate_packed" "insts:SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVES"
cd $GRAFT_REPO_ROOT
done
