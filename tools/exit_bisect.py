"""Bisects a crash inside exit() under rocprofv3 --memory-copy-trace: runs ONE of bench.py's side activities.
usage: python3 tools/exit_bisect.py hbm|events|gcfreeze|msssim|mark"""
import os, sys, torch
sys.path.insert(0, os.getcwd())
what = sys.argv[1]
dev = torch.device("cuda:0")
if what == "hbm":
    import bench
    print(bench.hbm_ceiling(dev))
elif what == "events":
    e = [torch.cuda.Event(enable_timing=True) for _ in range(200)]
    x = torch.ones(1 << 20, device=dev)
    for a in e:
        a.record(); x.mul_(1.0)
    torch.cuda.synchronize()
    print(e[0].elapsed_time(e[-1]))
elif what == "gcfreeze":
    import gc
    x = torch.ones(1 << 20, device=dev); gc.collect(); gc.freeze(); print(float(x.sum()))
elif what == "msssim":
    from compression_amd import pt_ms_ssim
    a = torch.rand(2, 3, 256, 256, device=dev)
    print(pt_ms_ssim.ms_ssim(a, a.clone(), data_range=1.0, size_average=False))
elif what == "mark":
    from compression_amd import ops
    ops.profile_mark(dev, 1); torch.cuda.synchronize(); print("marked")
elif what == "sysexit":
    x = torch.ones(1 << 20, device=dev); print(float(x.sum())); sys.exit(0)
elif what in ("codec", "codec_stats", "forward"):
    import compression_amd as ca
    from compression_amd import ops
    from compression_amd.distributed import StatsAccumulator
    from tests import synth
    torch.set_grad_enabled(False)
    m = ca.init_model("ManyPriors", dev, out_channel_N=192, out_channel_M=256, num_distributions=64, analysis_precision="f16x3", synthesis_precision="fp16")
    m.load_state_dict(synth.manypriors_state_dict(234)); m.eval()
    x = synth.rand_image(1000, 2, 256, 256).to(dev)
    if what == "forward":
        print(float(m(x)["bpp"]))
    else:
        s, idx, h, w = m.encode_batch(x)
        job = m.decode_batch_begin(s, idx, h, w)
        m.decode_batch_synthesis(job, score_against=x)
        xh = m.decode_batch_finish(job)
        if what == "codec_stats":
            acc = StatsAccumulator(device=dev)
            acc.add_batch_device(job["sse"], 3 * 256 * 256, bits=8.0, pixels=2 * 256 * 256, nbytes=1.0, seconds=0.1, reset=True)
            print(acc.all_reduce().result())
        print(float(xh.mean()))
elif what == "manycopies":                                  # torch alone: many small copies both ways on two streams
    h = torch.empty(1 << 16, pin_memory=True)
    d = torch.ones(1 << 16, device=dev)
    ss = [torch.cuda.Stream(), torch.cuda.Stream()]
    for i in range(int(sys.argv[2]) if len(sys.argv) > 2 else 2000):
        with torch.cuda.stream(ss[i & 1]):
            h.copy_(d, non_blocking=True); d.copy_(h, non_blocking=True); d.mul_(1.0)
    torch.cuda.synchronize(); print("copies done")
elif what == "bigalloc":                                    # torch alone: two 2 GiB tensors copied and filled, then dropped
    a = torch.empty(1 << 31, dtype=torch.uint8, device=dev); b = torch.empty(1 << 31, dtype=torch.uint8, device=dev)
    for _ in range(5):
        b.copy_(a); a.zero_()
    torch.cuda.synchronize(); del a, b; torch.cuda.empty_cache(); print("big done")
