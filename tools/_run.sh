cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05b
for rep in 1 2; do for n in 0 64 -64 96 128 32; do
TSDF_PIPE_SIDE_CUS=$n timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --path-only --repeats 3 > gpurun_out/r05b/c3_cu${n}_$rep.json 2>gpurun_out/r05b/err_$n.txt
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05b/*.json")):
    try:
        d=json.load(open(f)); ro=d["roofline_other"] if "integrate" in d["roofline"]["kernel"] else d["roofline"]
        print(f, d["ms_per_step"], d.get("ms_per_step_runs"), d.get("last_frame_vertex_bits"))
    except Exception as e: print(f, "ERR", e)
PY
tail -2 gpurun_out/r05b/err_64.txt
