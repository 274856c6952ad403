"""Developer tool: wall-clock breakdown of one bench step (synchronising between phases)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import compression_amd as ca
from compression_amd import ops
from tests import synth

torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
model = ca.init_model("ManyPriors", dev, out_channel_N=192, out_channel_M=256, num_distributions=64,
                      analysis_precision="f16x3", synthesis_precision="fp16")
model.load_state_dict(synth.manypriors_state_dict(234)); model.eval()
x = synth.rand_image(1000, 24, 512, 768).to(dev)
T = {}
def tick(name, t0):
    torch.cuda.synchronize(); T[name] = T.get(name, 0.0) + time.perf_counter() - t0; return time.perf_counter()
for it in range(6):
    if it == 1: T.clear()
    t = time.perf_counter()
    model._ensure_tables()
    x4 = ops.nchw_to_nhwc(x, 4); t = tick("nchw_to_nhwc", t)
    _, yhat = model.Encoder.forward_nhwc(x4, round_out=True, keep_unrounded=False); t = tick("g_a", t)
    sym = model._symbols(yhat); t = tick("symbols(+sync)", t)
    idx, minbits, _ = ops.prior_argmin(sym, model._lut_ln); t = tick("argmin", t)
    est = float(ops.sum_f32(minbits).item()) / 8 / 24; t = tick("sum+item", t)
    streams, geo = model._encode_symbols(sym, idx, 24, est); t = tick("rans_encode+D2H", t)
    rec = model.decode_batch(streams, idx.view(24, -1), 32, 48); t = tick("decode_batch", t)
    xhat, sse = ops.clamp_sse(rec, x); t = tick("clamp_sse", t)
tot = sum(T.values())
for k, v in T.items():
    print(f"{k:20s} {v / 5 * 1e3:8.3f} ms")
print(f"{'total':20s} {tot / 5 * 1e3:8.3f} ms")
