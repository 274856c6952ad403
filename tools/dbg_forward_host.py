"""Host-side profile of the eval-mode forward: where the time between kernels goes."""
import cProfile, os, pstats, sys, time, torch
sys.path.insert(0, os.getcwd())
import compression_amd as ca
from tests import synth
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
B, H, W = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (16, 256, 256)
m = ca.init_model("ManyPriors", dev, out_channel_N=192, out_channel_M=256, num_distributions=64, analysis_precision="f16x3")
m.load_state_dict(synth.manypriors_state_dict(234))
m.eval()
x = synth.rand_image(1000, B, H, W).to(dev)
for _ in range(3):
    m(x)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    out = m(x)
torch.cuda.synchronize()
print(f"{B}x{H}x{W}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per forward", flush=True)
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    out = m(x)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
