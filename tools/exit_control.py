"""Control for crashes inside exit() under a profiler: torch alone (an allocation, a kernel, copies both ways, a pinned
buffer, a second stream), no libmpc.  usage: python3 tools/exit_control.py"""
import torch
a = torch.ones(1 << 20, device="cuda")
h = torch.empty(1 << 20, pin_memory=True)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    b = (a * 2).to("cpu", non_blocking=True)
    h.copy_(a, non_blocking=True)
torch.cuda.synchronize()
print("control done", float(h[0]), float(b[0]))
