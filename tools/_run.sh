cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04j
timeout 1200 python -m pytest tests/test_parity_icp.py -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r04j/pytest2.log
cat gpurun_out/r04j/pytest2.log
for rep in 1 2; do for m in 0 1 2; do
  TSDF_ICP_PERSISTENT=$m timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --repeats 1 > gpurun_out/r04j/b_icp${m}_$rep.json 2> gpurun_out/r04j/b_icp${m}_$rep.err
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04j/b_icp*.json")):
    try:
        d=json.load(open(f)); print(f, "icp ms/frame", d["icp"]["ms_per_frame"], "tracking ms/frame", d["tracking"]["ms_per_frame"], d["tracking"]["max_translation_error_mm"])
    except Exception as e: print(f, "ERR", e)
PY
