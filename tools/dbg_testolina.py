import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import compression_amd as ca
from compression_amd import manynets_compressor as mc
from tests import synth
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
arch = sys.argv[1] if len(sys.argv) > 1 else "TestolinaManyPriors"
m = ca.init_model(arch, dev, out_channel_N=192, out_channel_M=256, num_distributions=64, analysis_precision="f16x3")
m.load_state_dict(synth.manypriors_state_dict(234, 192, 256, 64, spread=True, testolina=arch.startswith("Testolina")))
m.eval()
x = synth.noisy_image(synth.smooth_image(5, 8, 1024, 1024), 6).to(dev)
for rep in range(5):
    torch.cuda.synchronize()
    mc.HOST_TRACE = []
    a0 = torch.cuda.memory_stats()["num_device_alloc"]
    t0 = time.perf_counter()
    streams, idx, h, w = m.encode_batch(x)
    t1 = time.perf_counter()
    rec = m.decode_batch(streams, idx, h, w)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    tr = mc.HOST_TRACE
    print(arch, f"enc {1e3*(t1-t0):.2f} dec {1e3*(t2-t1):.2f} device allocs {torch.cuda.memory_stats()['num_device_alloc'] - a0}",
          " ".join(f"{l}:{1e3*(t-t0):.2f}" for l, t in tr), flush=True)
