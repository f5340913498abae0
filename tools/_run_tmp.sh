#!/bin/bash
cd $GRAFT_REPO_ROOT
TSDF_RAY_CELLS=2 timeout 600 python -m pytest tests/test_parity_raycast.py tests/test_fuzz_parity.py -m gpu -x -q -k "cell_parallel or fuzz or not_rigid or counted_again or tiny" -p no:cacheprovider 2>&1 | tail -3
timeout 600 python -m pytest tests/test_pipeline.py tests/test_multi_slab.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -3
for rep in 1 2; do python tools/dbg_ray_cells.py 40 512 2>&1 | tail -1; done
TSDF_RAY_CELLS=2 python tools/dbg_ray_cells.py 40 1024 inside 2>&1 | tail -1
