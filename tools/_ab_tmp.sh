cd /root/repo
python -m pytest tests/test_parity_raycast.py tests/test_fuzz_parity.py tests/test_multi_slab.py -m gpu -x -q 2>&1 | tail -3
TSDF_DEBUG_WAVES=1 python tools/dbg_ray_only.py 24 2>&1 | grep -E "tsdf:|segments" | tail -8
python tools/dbg_ray_only.py 24 2>&1 | tail -1
for seg in 4 5 6 8; do for b in 18 24 32; do TSDF_RAY_SEGMENTS=$seg TSDF_RAY_TRIP_BUDGET=$b python tools/dbg_ray_only.py 24 2>&1 | tail -1; done; done
