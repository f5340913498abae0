mkdir -p gpurun_out/r06u
( echo "# on the round's final sources (tools/extra_fuzz.py: 5 tests a seed -- random / smooth fields, random scenes, axis-aligned cameras, slab splits -- against the oracle, bit for bit)"
echo -n "TSDF_RAY_CELLS=2 (sorted list for views from inside), seeds 61000..61399: "; TSDF_RAY_CELLS=2 timeout 1500 python tools/extra_fuzz.py 61000 61400 2>&1 | tail -1
echo -n "TSDF_RAY_CELLS=2 TSDF_RAY_CELLS_SORT=2 (every list sorted), seeds 62000..62249: "; TSDF_RAY_CELLS=2 TSDF_RAY_CELLS_SORT=2 timeout 1200 python tools/extra_fuzz.py 62000 62250 2>&1 | tail -1
echo -n "defaults, seeds 63000..63249: "; timeout 1200 python tools/extra_fuzz.py 63000 63250 2>&1 | tail -1
echo -n "TSDF_RAY_CELLS=1 TSDF_RAY_CHOOSER=2 (trials of the other cast every few casts), seeds 64000..64199: "; TSDF_RAY_CHOOSER=2 timeout 1200 python tools/extra_fuzz.py 64000 64200 2>&1 | tail -1
) > gpurun_out/r06u/extra_fuzz.txt 2>&1
cat gpurun_out/r06u/extra_fuzz.txt
