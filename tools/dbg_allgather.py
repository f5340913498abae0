"""What one all_gather_into_tensor of the hit records costs on the stream (world of one rank, nccl = RCCL):
    python tools/dbg_allgather.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
W, H = 640, 480
mine = torch.zeros((H * W, 4), dtype=torch.float32, device="cuda")
allb = torch.empty((1, H * W, 4), dtype=torch.float32, device="cuda")
x = torch.zeros((1 << 20,), device="cuda")
s = torch.cuda.current_stream()
for _ in range(5):
    dist.all_gather_into_tensor(allb.view(-1), mine.view(-1))
torch.cuda.synchronize()
for label, n_between in (("back to back", 0), ("with a kernel before and after each", 1)):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record(s)
    for _ in range(50):
        if n_between: x.add_(1.0)
        dist.all_gather_into_tensor(allb.view(-1), mine.view(-1))
        if n_between: x.add_(1.0)
    e1.record(s); torch.cuda.synchronize()
    print("%-40s %.1f us per call on the stream, %.1f us wall" % (label, e0.elapsed_time(e1) * 1e3 / 50, (time.perf_counter() - t0) * 1e6 / 50), file=sys.stderr)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(s)
for _ in range(50):
    x.add_(1.0); allb.view(-1).copy_(mine.view(-1)); x.add_(1.0)
e1.record(s); torch.cuda.synchronize()
print("%-40s %.1f us per call on the stream" % ("a plain copy between the same kernels", e0.elapsed_time(e1) * 1e3 / 50), file=sys.stderr)
dist.destroy_process_group()
