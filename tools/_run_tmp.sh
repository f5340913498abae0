for g in 512 256; do for v in tight shell512 shell256 shell128 tight shell256; do
  echo -n "$v "; TSDF_HIP_LIB=$GRAFT_REPO_ROOT/build/variants/$v/libtsdf_hip.so python tools/dbg_ray_cells.py 40 $g 2>&1 | tail -1
done; done
