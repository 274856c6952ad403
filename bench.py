#!/usr/bin/env python3
"""Benchmark of the Manypriors codec inference path on MI355X.

Workload (the configuration the encode+decode metric is quoted on): the full codec
on a batch of 24 synthetic Kodak-shaped 768x512 images per GPU with the standard model (N=192, M=256, K=64,
random-init transforms, spread priors): one STEP =
    encode:  g_a -> round -> symbols -> prior competition -> MRS1 rANS -> byte strings on the host
    decode:  byte strings -> rANS decode -> latent -> g_s -> clamp -> x_hat on the device
with the input batch already resident in HBM.  value = megapixels (B*H*W of the input) per second over all
ranks.  Images are independent, so N GPUs run N independent batches (weak scaling, no data-path collective);
the only collective is one all-reduce of a 16-double statistics vector per step.

Extra objects on the JSON line: "roofline" for the dominant kernel (the implicit-GEMM conv with fused GDN of the
analysis transform; achieved = algorithmic FLOP / HIP-event time, traffic = bytes per launch from the committed
counter passes), "cpu_baseline" (the CPU oracle timed on this host on a bounded sample), "kernels" (per-kernel achieved
rates of the other launches, including the materialised CDF build that the fused coder does not need).
"""
import argparse
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

PEAK_F32_MFMA_TFLOPS = 157.3     # dense fp32-input MFMA peak, MI355X
PEAK_F16_MFMA_TFLOPS = 2500.0    # dense fp16/bf16 MFMA peak, MI355X
PEAK_HBM_GBS = 8000.0


def kernel_peak(name):
    """(peak TFLOP/s of the MFMA type the kernel issues, MFMA instructions issued per algorithmic product)."""
    if name.startswith("conv_igemm_x3"):
        return PEAK_F16_MFMA_TFLOPS, 3        # a = hi + lo: three fp16 MFMAs per product
    if name.startswith("conv_igemm_f16"):
        return PEAK_F16_MFMA_TFLOPS, 1
    return PEAK_F32_MFMA_TFLOPS, 1


def pmc_traffic(kernel):
    """HBM-side bytes per launch of `kernel` from the committed rocprofv3 counter passes (profiles/r1_pmc_counters.csv,
    made by tools/pmc_run.sh: one --pmc pass for FETCH_SIZE, one for WRITE_SIZE; FETCH_SIZE doubled as gfx950 tallies
    its 128-byte read requests at 64 bytes).  None when the table has no row for the kernel."""
    import csv
    path = os.path.join(REPO, "profiles", "r1_pmc_counters.csv")
    if not os.path.exists(path):
        return None
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            if kernel in r["Kernel"] and r.get("FETCH_SIZE") and r.get("WRITE_SIZE"):
                return (2.0 * float(r["FETCH_SIZE"]) + float(r["WRITE_SIZE"])) * 1024.0
    return None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=24)
    ap.add_argument("--height", type=int, default=512)
    ap.add_argument("--width", type=int, default=768)
    ap.add_argument("--data", default="rand", choices=["rand", "smooth"])
    ap.add_argument("--analysis-precision", default="f16x3", choices=["fp32", "f16x3"],
                    help="g_a arithmetic: exact f32 MFMA, or split-precision fp16 MFMA (fp32-grade, same symbols)")
    ap.add_argument("--synthesis-precision", default="fp16", choices=["fp32", "fp16"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-images", type=int, default=48)
    return ap.parse_args()


def cpu_baseline(n_images, H, W):
    """The CPU oracle (PyTorch-CPU fp32 restatement + C range coder) on n_images images, all host threads
    torch was given.  Timed: g_a, round, LUT competition, range encode, range decode, g_s."""
    import numpy as np
    from oracle import coders, manypriors_oracle as O
    from tests import synth
    torch.set_grad_enabled(False)
    sd = synth.manypriors_state_dict(234)
    tf = O.entropy_table_float(sd, 256).contiguous()
    table = O.renorm_cast_cdf(tf, 16, O.ORDER_ROUND_THEN_ADD).numpy()
    x = synth.rand_image(99, n_images, H, W)
    O.analysis(x[:1, :, :64, :64], sd)                      # warm the thread pool
    t0 = time.perf_counter()
    for i in range(n_images):
        yhat = O.quantise(O.analysis(x[i:i + 1], sd))
        _, M, h, w = yhat.shape
        _, idx = O.prior_bits_lut_encode_layout(yhat, tf).min(0)
        sym = O.symbols_from_latent(yhat).flatten().numpy()
        idx32 = idx.reshape(-1).numpy().astype(np.int32)
        data = coders.range_encode_table(table, idx32, sym)
        dec = coders.range_decode_table(table, idx32, data)
        lat = O.latent_from_symbols(torch.from_numpy(dec).view(1, -1), M, h, w)
        O.synthesis(lat, sd).clamp(0, 1)
    dt = time.perf_counter() - t0
    return {"value": n_images * H * W / 1e6 / dt, "unit": "megapixels/s", "cores": torch.get_num_threads(),
            "kind": "port", "sample": f"{n_images} of the {H}x{W} images, encode+decode, {dt:.1f} s of CPU time"}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)
    torch.set_grad_enabled(False)

    import compression_amd as ca
    from compression_amd import ops
    from tests import synth

    model = ca.init_model("ManyPriors", dev, out_channel_N=192, out_channel_M=256, num_distributions=64,
                          analysis_precision=args.analysis_precision, synthesis_precision=args.synthesis_precision)
    model.load_state_dict(synth.manypriors_state_dict(234))
    model.eval()
    gen = synth.rand_image if args.data == "rand" else synth.smooth_image
    x = gen(1000 + rank, args.batch, args.height, args.width).to(dev)
    B, _, H, W = x.shape
    mp_per_step = B * H * W / 1e6
    stats = torch.zeros(16, dtype=torch.float64, device=dev)

    def step():
        streams, idx, h, w = model.encode_batch(x)
        recon = model.decode_batch(streams, idx, h, w)
        xhat, sse = ops.clamp_sse(recon, x)
        stats.zero_()
        stats[0] = sum(len(s) for s in streams) * 8.0      # bits
        stats[1] = B * H * W                                # pixels
        stats[2:3] = sse
        stats[3] = B
        if dist is not None:
            dist.all_reduce(stats)
        return streams, xhat

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    prof = ops.PROFILER = ops.Profiler()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        streams, xhat = step()
    barrier()
    dt = time.perf_counter() - t0
    ops.PROFILER = None
    kern = prof.summary()
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # the materialised per-symbol CDF rows ("CDF build"): measured on its own, the fused coder never needs it
    n_loc_probe = min(4096, B * (H // 16) * (W // 16))
    idx_probe = torch.randint(0, 64, (n_loc_probe,), dtype=torch.int32, device=dev)
    ops.gather_cdf(model.entropy_table, idx_probe)
    p2 = ops.PROFILER = ops.Profiler()
    for _ in range(5):
        ops.gather_cdf(model.entropy_table, idx_probe)
    ops.PROFILER = None
    kern.update(p2.summary())

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return
    bits = float(stats[0].item())
    value = world * mp_per_step * args.steps / dt
    dom = max((k for k in kern if k.startswith("conv_igemm")), key=lambda k: kern[k]["ms"])
    d = kern[dom]
    achieved = d["flops"] / (d["ms"] * 1e-3) / 1e12
    peak, passes = kernel_peak(dom)
    kernels = {}
    for k, v in kern.items():
        e = {"calls_per_step": v["calls"] / (5 if k == "gather_cdf" else args.steps), "avg_ms": v["ms"] / v["calls"]}
        if v["flops"]:
            e["tflops"] = v["flops"] / (v["ms"] * 1e-3) / 1e12
        if v["bytes"]:
            e["gb_per_s"] = v["bytes"] / (v["ms"] * 1e-3) / 1e9
        kernels[k] = e
    out = {
        "metric": "encode+decode megapixels/sec", "value": value, "unit": "megapixels/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": {"fp32": "f32", "f16x3": "f16 hi/lo split x3 (fp32-grade), f32 accumulate"}[args.analysis_precision]
                 + " (g_a); " + {"fp32": "f32", "fp16": "f16, f32 accumulate"}[args.synthesis_precision] + " (g_s)",
        "data": "synthetic",
        "config": {"workload": f"full codec (encode -> bytes -> decode) on {B} x {W}x{H} images per GPU, std Manypriors "
                               f"N=192 M=256 K=64, random-init weights with spread priors, {args.data} input",
                   "images_per_gpu": B, "height": H, "width": W, "bpp": bits / (world * B * H * W),
                   "parallelism": f"image-sharded x{world}, stats all-reduce only"},
        "roofline": {"bound": "mfma", "kernel": dom, "achieved": achieved, "peak": peak,
                     "unit": "TFLOP/s", "frac": achieved / peak, "traffic": pmc_traffic(dom),
                     "mfma_passes_per_product": passes, "issued_tflops": achieved * passes,
                     "avg_launch_ms": d["ms"] / d["calls"], "launches_per_step": d["calls"] / args.steps},
        "kernels": kernels,
    }
    if "gather_cdf" in kernels:
        out["cdf_build"] = {"bound": "hbm", "achieved": kernels["gather_cdf"]["gb_per_s"], "peak": PEAK_HBM_GBS,
                            "unit": "GB/s", "frac": kernels["gather_cdf"]["gb_per_s"] / PEAK_HBM_GBS}
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.cpu_images, H, W)
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
