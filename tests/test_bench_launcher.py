"""bench.py --gpus N started plainly (no torchrun, WORLD_SIZE unset) becomes its own launcher (round 4).
CPU check of the command it would exec; the run itself is covered on the GPU box by tests/test_bench_contract.py."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plain_multi_gpu_invocation_re_executes_under_torch_distributed_run():
    env = dict(os.environ, BENCH_LAUNCH_DRYRUN="1")
    env.pop("WORLD_SIZE", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "20", "--warmup", "5"], cwd=ROOT, timeout=120,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
    assert p.returncode == 0, p.stderr[-1000:]
    lines = p.stdout.strip().splitlines()
    assert len(lines) == 1
    cmd = json.loads(lines[0])
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    tail = cmd[cmd.index(os.path.join(ROOT, "bench.py")):]
    assert tail[1:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]
