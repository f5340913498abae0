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
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert math.isclose(r["frac"], r["achieved"] / r["peak"], rel_tol=1e-3) and 0.05 < r["frac"] < 1.0
    assert r["launches_timed"] >= 1 and r["avg_launch_ms"] > 0 and r["algorithmic_bytes"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == d["unit"] and c["sample"]
    assert d["parity"]["pass"] is True
    # BASELINE configs[4] runs inside the bench at 512^3: its outcome is asserted, not just reported
    t = d["tracking"]
    assert t["frames"] >= 24 and t["max_translation_error_mm"] < 10.0 and t["max_rotation_error_rad"] < 0.006
    assert t["mesh"]["same_as_host"] is True and t["mesh"]["triangles"] > 500_000


def test_few_steps_and_no_sampled_events_still_give_strict_json():
    d = run_bench("--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-parity", "--event-period", "0")
    assert d["steps"] == 3 and d["value"] > 0
