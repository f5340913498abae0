"""bench.py's output contract on the GPU box: one strict-JSON line with the fields the driver reads, the roofline and
cpu_baseline objects, and a parity gate that passed."""
import json
import math
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*args):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], check=True, cwd=ROOT, timeout=900,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True).stdout.strip().splitlines()
    assert len(out) == 1, "bench.py must print exactly one line on stdout, got %d" % len(out)

    def no_constants(name):     # NaN / Infinity are not JSON
        raise ValueError("non-finite number in the JSON line: " + name)
    return json.loads(out[0], parse_constant=no_constants)


def test_default_shaped_run_prints_the_contract_line():
    d = run_bench("--gpus", "1", "--steps", "20", "--warmup", "2", "--cpu-budget-s", "3")
    for key, kind in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                      ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str),
                      ("config", dict), ("roofline", dict), ("cpu_baseline", dict)):
        assert key in d and isinstance(d[key], kind), key
    assert "vs_baseline" in d and d["vs_baseline"] is None          # BASELINE.md publishes no number for this metric
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 2 and d["higher_is_better"] is True
    assert d["unit"] == "Mvoxels/s" and "workload" in d["config"] and "model" not in d["config"]
    # value is the whole step's throughput: grid voxels * steps / wall time
    n_vox = d["config"]["grid"][0] * d["config"]["grid"][1] * d["config"]["grid"][2]
    assert math.isclose(d["value"], n_vox / (d["ms_per_step"] * 1e-3) / 1e6, rel_tol=2e-3)
    # the timed region is repeated; the line carries every repetition and reports their median
    runs = d["ms_per_step_runs"]
    assert len(runs) >= 5 and math.isclose(sorted(runs)[len(runs) // 2], d["ms_per_step"], rel_tol=1e-3) and d["picture_bits_equal_across_runs"] is True
    # the parity gate ran on the benchmarked grid, first and last timed frame
    assert d["parity"]["grid"] == d["config"]["grid"][0] and d["parity"]["rays_compared"] >= 2 * 120 * 640 and d["parity"]["dist_bit_mismatch"] == 0
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert math.isclose(r["frac"], r["achieved"] / r["peak"], rel_tol=1e-3) and 0.05 < r["frac"] < 1.0
    assert r["launches_timed"] >= 1 and r["avg_launch_ms"] > 0 and r["algorithmic_bytes"] > 0
    # the same kernels as they run in the two-stream step (what a kernel trace of the command shows): never faster than alone by more than noise
    for rr in (r, d["roofline_other"]):
        assert rr["avg_launch_ms_pipelined"] > 0.8 * rr["avg_launch_ms"]
    # the kernel the line names is the one the pipelined step waits for, both carry the counters' fields, the spread is over >= 100 steps
    assert "dominant_by" in r and all(k in rr for rr in (r, d["roofline_other"]) for k in ("valu_busy", "lanes_active", "frac", "traffic"))
    pip = {rr["kernel"]: rr["avg_launch_ms_pipelined"] for rr in (r, d["roofline_other"])}
    assert pip[r["kernel"]] == max(pip.values())
    assert d["step_ms_spread"]["steps"] >= 100 and len(d["step_ms_spread"]["worst_steps"]) == 5
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == d["unit"] and c["sample"]
    assert d["parity"]["pass"] is True
    # BASELINE configs[4] runs inside the bench at 512^3: its outcome is asserted, not just reported
    t = d["tracking"]
    assert t["frames"] >= 24 and t["max_translation_error_mm"] < 10.0 and t["max_rotation_error_rad"] < 0.006
    assert t["mesh"]["same_as_host"] is True and t["mesh"]["triangles"] > 500_000


def test_few_steps_and_no_sampled_events_still_give_strict_json():
    d = run_bench("--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-parity", "--event-period", "0", "--repeats", "1", "--spread-steps", "0")
    assert d["steps"] == 3 and d["value"] > 0 and len(d["ms_per_step_runs"]) == 1


def test_more_ranks_than_gpus_is_refused_with_a_reason():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None); env.pop("TSDF_BENCH_SHARE_GPU", None)
    import torch
    n = torch.cuda.device_count() + 1
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1"], cwd=ROOT, timeout=600,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
    assert p.returncode != 0 and "this node shows" in p.stderr and p.stdout.strip() == ""


def test_two_ranks_on_one_gpu_walk_the_sharded_path_and_agree_with_one_volume():
    """N > 1 cannot run over RCCL on a one-GPU box; TSDF_BENCH_SHARE_GPU=1 puts both ranks on device 0 and exchanges over gloo.
    Everything else is the N = 2 path: measured slab plan, slab integrate + slab ray cast, all-gather of the hit records,
    min-k merge, next frame's filter + integrate overlapped with the exchange, and rank 0's parity replay against ONE volume."""
    env = dict(os.environ, TSDF_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    # started plainly, the way the driver starts N = 1: bench.py becomes its own torch.distributed.run launcher (round 4)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--plan-rounds", "1", "--repeats", "3", "--spread-steps", "0"]
    p = subprocess.run(cmd, cwd=ROOT, timeout=900, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, "rank 0 prints exactly one JSON line, got %d" % len(lines)
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["parallelism"] == "zslab2"
    assert len(d["ms_per_step_runs"]) == 3 and d["picture_bits_equal_across_runs"] is True
    assert d["parity"]["pass"] is True and d["parity"]["merged_picture_equals_single_volume_replay"] is True
    assert len(d["slabs"]) == 2 and d["slabs"][0][0] == 0 and d["slabs"][0][1] == d["slabs"][1][0] and d["slabs"][1][1] == 512
    for name in ("integrate", "raycast", "exchange", "integrate_kernel", "process_ray_kernel"):
        assert len(d["per_rank_ms"][name]) == 2 and all(x > 0 for x in d["per_rank_ms"][name]), name


def test_one_rank_walks_the_sharded_path_over_rccl():
    """What a one-GPU box can check of the multi-GPU path on the real backend: a world of ONE rank initialises nccl (= RCCL),
    builds a communicator on the box's GPU and pushes the hit records through all_gather_into_tensor on device memory, between the
    slab ray cast and the merge kernel of every step; the merged picture must equal a single-volume replay."""
    d = run_bench("--gpus", "1", "--one-rank-slab-path", "--steps", "6", "--warmup", "2", "--plan-rounds", "1", "--no-cpu-baseline", "--validate-merge", "--spread-steps", "0")
    assert d["config"]["collective_backend"] == "nccl" and d["config"]["ranks"] == 1 and d["config"]["parallelism"] == "zslab1"
    # the line says how many ranks the communicator itself holds (ncclCommCount), and mode B -- the all-gathered distance slab cast
    # the single-volume way -- gives the merged picture bit for bit
    assert d["collective"]["ranks_seen"] == 1 and d["collective"]["world"] == 1
    assert d["collective"]["validate_merge"]["pass"] is True and d["collective"]["validate_merge"]["differing_words_max_over_ranks"] == 0
    assert d["parity"]["pass"] is True and d["parity"]["merged_picture_equals_single_volume_replay"] is True
    assert d["slabs"] == [[0, 512]]
    assert d["per_rank_ms"]["exchange"][0] > 0 and d["value"] > 0
    # the collective was RCCL's, called on the step's own stream (no fallback to torch's)
    assert d["config"]["collective"].startswith("ncclAllGather on the step's stream"), d["config"]["collective"]
    t = run_bench("--gpus", "1", "--one-rank-slab-path", "--torch-collective", "--steps", "6", "--warmup", "2", "--plan-rounds", "0",
                  "--no-cpu-baseline", "--spread-steps", "0")
    assert t["config"]["collective"].startswith("torch.distributed") and t["parity"]["pass"] is True
    assert t["last_frame_vertex_checksum"] == d["last_frame_vertex_checksum"]
