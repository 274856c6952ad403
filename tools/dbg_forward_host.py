"""Host-side profile of the eval-mode forward in throughput mode (ForwardPipeline): where the host's time per batch goes.
usage: python tools/dbg_forward_host.py [B H W]"""
import cProfile, os, pstats, sys, time, torch
sys.path.insert(0, os.getcwd())
import compression_amd as ca
from compression_amd.pipeline import ForwardPipeline
from tests import synth
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
B, H, W = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (16, 256, 256)
m = ca.init_model("ManyPriors", dev, out_channel_N=192, out_channel_M=256, num_distributions=64, analysis_precision="f16x3")
m.load_state_dict(synth.manypriors_state_dict(234))
m.eval()
x = synth.rand_image(1000, B, H, W).to(dev)
pipe = ForwardPipeline(m, dev, depth=2)
def run(n):
    for _ in range(n):
        r = pipe.submit(x)
        if r is not None:
            float(r[0]["bpp"])
    for r in pipe.flush():
        float(r[0]["bpp"])
    pipe.synchronize()
run(6)
t0 = time.perf_counter()
run(100)
print(f"{B}x{H}x{W}: {(time.perf_counter() - t0) / 100 * 1e3:.3f} ms per forward (pipelined)", flush=True)
# host cost of the enqueue alone: begin without finishing (GPU far behind does not matter for the host's clock)
torch.cuda.synchronize()
t0 = time.perf_counter()
jobs = [m.forward_begin(x) for _ in range(20)]
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"forward_begin: {(t1 - t0) / 20 * 1e3:.3f} ms of host time per call", flush=True)
t0 = time.perf_counter()
outs = [m.forward_finish(j) for j in jobs]
print(f"forward_finish (GPU idle): {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms of host time per call", flush=True)
pr = cProfile.Profile()
pr.enable()
run(100)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(30)
