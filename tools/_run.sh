cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do 
TAG=cz32 timeout 300 python tools/dbg_integrate_only.py 2>&1 | tail -1
for n in cz16 cz64; do
TAG=$n TSDF_HIP_LIB=$GRAFT_REPO_ROOT/build/variants/$n/libtsdf_hip.so timeout 300 python tools/dbg_integrate_only.py 2>&1 | tail -1
done; done | tee gpurun_out/r04q/chunkz.txt
