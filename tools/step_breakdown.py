"""Developer tool: host-side timeline of bench steps (config 3) without extra synchronisation: where the host is
between the milestones of encode_batch / decode_batch, next to the sum of the kernels' own durations."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import compression_amd as ca
from compression_amd import manynets_compressor as mc, ops
from tests import synth

torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
model = ca.init_model("ManyPriors", dev, out_channel_N=192, out_channel_M=256, num_distributions=64,
                      analysis_precision="f16x3", synthesis_precision="fp16")
model.load_state_dict(synth.manypriors_state_dict(234)); model.eval()
x = synth.rand_image(1000, 24, 512, 768).to(dev)


def step():
    streams, idx, h, w = model.encode_batch(x)
    recon = model.decode_batch(streams, idx, h, w)
    mc._mark("dec:returned")
    xhat, sse = ops.clamp_sse_per_image(recon, x)
    return xhat


for _ in range(3):
    step()
torch.cuda.synchronize()
acc, n = {}, 8
for it in range(n):
    mc.HOST_TRACE = tr = []
    t0 = time.perf_counter()
    step()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    prev = t0
    for label, t in tr:
        acc[label] = acc.get(label, 0.0) + (t - prev)
        prev = t
    acc["final sync"] = acc.get("final sync", 0.0) + (t1 - prev)
    acc["TOTAL"] = acc.get("TOTAL", 0.0) + (t1 - t0)
mc.HOST_TRACE = None
for k, v in acc.items():
    print(f"{k:22s} {v / n * 1e3:8.3f} ms (host time up to this milestone)")
