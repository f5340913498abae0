( timeout 600 python -m pytest tests/test_parity_raycast.py -m gpu -q -k "tiny_images" ) 2>&1 | tail -2
( TSDF_HIP_LIB=$GRAFT_REPO_ROOT/build/variants/bug/libtsdf_hip.so timeout 600 python -m pytest tests/test_parity_raycast.py -m gpu -q -k "tiny_images" ) 2>&1 | tail -3 | cut -c1-200
