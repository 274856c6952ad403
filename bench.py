#!/usr/bin/env python3
"""Benchmark of the Manypriors / JDC codec inference path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 1..5]

Workloads (--config, default 3 = the configuration the encode+decode metric is quoted on):
    1  std Manypriors, 1 x 768x512, eval forward (x_hat + bpp estimate)            forward MP/s
    2  training shape, 16 x 256x256 crops, eval forward                           forward MP/s
    3  full codec on 24 x 768x512 per GPU                                         encode+decode MP/s
    4  full codec on one 3840x2160 image per GPU per step                         encode+decode MP/s
    5  JDC: ManyPriors_DC (or --arch TestolinaManyPriors) on 8 noisy 1024x1024 per GPU, full codec
    6  hyperprior codec: TestolinaHP (or --arch Balle2018PTTFExp) on 8 x 768x512 per GPU, full codec (not a BASELINE
       configuration: the reference's hyperprior encode/decode path, abstract_model.py:370-500)
One codec STEP =
    encode:  g_a -> round -> symbols -> prior competition -> MRS1 rANS -> byte strings on the host
    decode:  byte strings -> rANS decode -> latent -> g_s -> clamp -> x_hat on the device
    + per-image squared error and the batch's statistics vector (one 16-double all-reduce when N > 1)
with the input batch already resident in HBM.  value = megapixels (B*H*W of the input) per second over all ranks.
The codec workloads keep TWO batches in flight on two HIP streams (compression_amd/pipeline.py: step n's coders run
beside the other batch's transforms; every batch still goes encode -> bytes on the host -> decode).  The timed region
starts with nothing in flight and ends when all of its K batches are decoded.  --serial: one batch after the other.

N GPUs: images are independent, every rank codes its own batch (weak scaling, no data-path collective).  Started as
`python bench.py --gpus N` the script starts N child processes itself (one per GPU, RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_* set per child) before it touches any GPU, and relays rank 0's JSON line; started by torch.distributed.run it
takes its rank from the environment.  --dry-run rehearses the launcher and the statistics path on the CPU (gloo).

Extra objects on the JSON line: "roofline" for the dominant kernel (achieved = algorithmic FLOP / its average launch
time by HIP events round each of its launches INSIDE the timed region, on the stream it runs on; the kernel is picked
in the warm-up, where every launch carries events; traffic = bytes per launch from the committed counter passes, null
if they were taken on other kernel sources), "kernels" (per-launch averages of all kernels: the dominant one from the
timed region, the others from up to 20 steps run after it one batch at a time with events round every
launch -- ~40 events a step inside the timed region cost 0.6 ms of it and more, growing with the number of steps),
"cdf_build" (the materialised per-symbol CDF rows at the workload's size, which the fused coder never needs),
"hbm_ceiling" (device copy / fill rates measured in this run), "cpu_baseline" (the CPU oracle timed on this host),
"stats" (the all-reduced evaluation statistics of one step), "device_allocs_in_timed_region" (hipMalloc calls of torch's
allocator while the clock ran: 0 once the warm-up has reached the steady state).
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

PEAK_F32_MFMA_TFLOPS = 157.3     # dense fp32-input MFMA peak, MI355X
PEAK_F16_MFMA_TFLOPS = 2500.0    # dense fp16/bf16 MFMA peak, MI355X
PEAK_HBM_GBS = 8000.0

CONFIGS = {
    1: dict(arch="ManyPriors", batch=1, height=512, width=768, mode="forward", data="rand",
            what="std Manypriors forward + bpp estimate on one 768x512 image"),
    2: dict(arch="ManyPriors", batch=16, height=256, width=256, mode="forward", data="rand",
            what="training-shape eval forward on 16 x 256x256 crops"),
    3: dict(arch="ManyPriors", batch=24, height=512, width=768, mode="codec", data="rand",
            what="full codec (encode -> bytes -> decode) on 24 x 768x512 images"),
    4: dict(arch="ManyPriors", batch=1, height=2160, width=3840, mode="codec", data="rand",
            what="full codec on one 3840x2160 image"),
    5: dict(arch="ManyPriors_DC", batch=8, height=1024, width=1024, mode="codec", data="noisy",
            what="JDC full codec on 8 noisy 1024x1024 images"),
    6: dict(arch="TestolinaHP", batch=24, height=512, width=768, mode="hp_codec", data="smooth",
            what="hyperprior full codec (encode -> bytes -> decode, main stream under per-element Laplace rows) "
                 "on 24 x 768x512 images"),
}
# sources whose change invalidates the committed counter passes of the dominant kernel
TRAFFIC_SOURCES = ("compression_amd/csrc/conv5s2_patch_x3.hip", "compression_amd/csrc/conv_x3_epilogue.h",
                   "compression_amd/csrc/conv_common.h")
PMC_TABLE = os.path.join(REPO, "profiles", "r3_pmc_counters.csv")
PMC_META = os.path.join(REPO, "profiles", "r3_pmc_meta.json")
NUM_SIMD = 256 * 4               # SIMDs of the chip
NUM_XCD = 8


def kernel_peak(name):
    """(peak TFLOP/s of the MFMA type the kernel issues, MFMA instructions issued per algorithmic product)."""
    if name.startswith("conv_igemm_x3") or name.startswith("conv5s2_patch_x3") or name.startswith("conv1_rgb_patch_x3"):
        return PEAK_F16_MFMA_TFLOPS, 3        # a = hi + lo: three fp16 MFMAs per product
    if name.startswith("conv_igemm_f16") or name.startswith("deconv5s2_patch_f16") or name.startswith("deconv_rgb16"):
        return PEAK_F16_MFMA_TFLOPS, 1
    return PEAK_F32_MFMA_TFLOPS, 1


def sources_digest():
    h = hashlib.sha256()
    for rel in TRAFFIC_SOURCES:
        with open(os.path.join(REPO, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def pmc_traffic(kernel):
    """-> (HBM-side bytes per launch of `kernel`, commit the counters were taken at) from the committed rocprofv3
    counter passes (tools/pmc_run.sh: one --pmc pass for FETCH_SIZE, one for WRITE_SIZE; FETCH_SIZE doubled: gfx950
    tallies its 128-byte read requests at 64 bytes, checked against a copy of known size in the same pass).  (None, ...)
    when the table has no row for the kernel or was taken on other sources of that kernel than the ones here."""
    import csv
    if not (os.path.exists(PMC_TABLE) and os.path.exists(PMC_META)):
        return None, None
    meta = json.load(open(PMC_META))
    if meta.get("sources_sha256_16") != sources_digest():
        return None, meta.get("commit")
    with open(PMC_TABLE, newline="") as f:
        for r in csv.DictReader(f):
            if kernel in r["Kernel"] and r.get("FETCH_SIZE") and r.get("WRITE_SIZE"):
                return (2.0 * float(r["FETCH_SIZE"]) + float(r["WRITE_SIZE"])) * 1024.0, meta.get("commit")
    return None, meta.get("commit")


def pmc_mfma(kernel):
    """-> {"mfma_busy", "clock_ghz", "profiled_at"} of `kernel` from the committed counter passes, or None: the fraction
    of the chip's SIMD cycles in which the matrix pipe was busy (SQ_VALU_MFMA_BUSY_CYCLES over GRBM_GUI_ACTIVE / 8 XCDs x
    1024 SIMDs) and the clock the kernel ran at under the counters (GRBM_GUI_ACTIVE / 8 over its duration)."""
    import csv
    if not (os.path.exists(PMC_TABLE) and os.path.exists(PMC_META)):
        return None
    meta = json.load(open(PMC_META))
    with open(PMC_TABLE, newline="") as f:
        for r in csv.DictReader(f):
            if kernel in r["Kernel"] and r.get("SQ_VALU_MFMA_BUSY_CYCLES") and r.get("GRBM_GUI_ACTIVE"):
                cyc = float(r["GRBM_GUI_ACTIVE"]) / NUM_XCD
                if cyc <= 0:
                    return None
                return {"mfma_busy": float(r["SQ_VALU_MFMA_BUSY_CYCLES"]) / (cyc * NUM_SIMD),
                        "clock_ghz": cyc / float(r["AvgNs"]), "profiled_at": meta.get("commit"),
                        "same_kernel_sources": meta.get("sources_sha256_16") == sources_digest()}
    return None


def parse(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100,
                    help="timed steps (with two batches in flight the region includes filling and draining the pipeline, "
                         "about half a step's time: 5 %% of 10 steps, 0.5 %% of 100)")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, default=3, choices=sorted(CONFIGS))
    ap.add_argument("--arch", default=None, help="model class (default: the config's; e.g. TestolinaManyPriors for config 5)")
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--data", default=None, choices=["rand", "smooth", "noisy"])
    ap.add_argument("--analysis-precision", default="f16x3", choices=["fp32", "f16x3"],
                    help="g_a arithmetic: exact f32 MFMA, or split-precision fp16 MFMA (fp32-grade, same symbols)")
    ap.add_argument("--synthesis-precision", default="fp16", choices=["fp32", "fp16"])
    ap.add_argument("--serial", action="store_true",
                    help="codec workloads: one batch after the other on one stream (default: two batches in flight on two streams)")
    ap.add_argument("--forward-depth", type=int, default=3, help="forward workloads: batches in flight (ForwardPipeline)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-images", type=int, default=40, help="images of the CPU baseline's sample at the larger thread count")
    ap.add_argument("--launch-timeout", type=float, default=3600.0,
                    help="--gpus N launcher: seconds after which ranks that are still running are ended")
    ap.add_argument("--fail-rank", type=int, default=None,
                    help="dry run only: this rank exits with an error before it joins the process group (launcher test)")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU: gloo ranks on the CPU with a stub step, to rehearse the launcher and the statistics path")
    args = ap.parse_args(argv)
    cfg = CONFIGS[args.config]
    for k in ("arch", "batch", "height", "width", "data"):
        if getattr(args, k) is None:
            setattr(args, k, cfg[k])
    args.mode = cfg["mode"]
    return args


# ---- launcher -----------------------------------------------------------------------------------------------
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def child_env(rank, world, port, base=None):
    """Environment of rank `rank` of `world` single-node ranks."""
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL over xGMI needs it on this driver
    return env


def spawn_ranks(args, argv):
    """Parent of `python bench.py --gpus N`: starts N fresh children (this process never initialises a GPU), waits,
    relays rank 0's JSON line.  Returns the exit code."""
    if not args.dry_run:
        n = torch.cuda.device_count()                      # counting devices does not initialise the GPU
        if n < args.gpus:
            print(f"bench.py: --gpus {args.gpus} but only {n} GPU(s) are visible", file=sys.stderr)
            return 2
    import tempfile
    port = free_port()
    cmd = [sys.executable, os.path.abspath(__file__)] + list(argv)
    procs = []
    # rank 0 writes to a file, not a pipe: nobody has to drain it while the ranks are polled
    with tempfile.TemporaryFile(mode="w+") as out0:
        for r in range(args.gpus):
            env = child_env(r, args.gpus, port)
            if args.fail_rank == r:
                env["MPC_BENCH_FAIL_EARLY"] = "1"
            procs.append(subprocess.Popen(cmd, env=env, stdout=out0 if r == 0 else subprocess.DEVNULL, text=True))
        # All ranks are polled together.  The first one that exits with an error ends the job: the others may be sitting
        # in the process group's rendezvous or in a barrier that can never complete, holding their GPUs.
        deadline = time.monotonic() + args.launch_timeout
        failed = None
        while failed is None and any(p.poll() is None for p in procs):
            failed = next((p for p in procs if p.poll() not in (None, 0)), None)
            if failed is None and time.monotonic() > deadline:
                print(f"bench.py: ranks still running after {args.launch_timeout:.0f} s, giving up", file=sys.stderr)
                failed = next(p for p in procs if p.poll() is None)
            if failed is None:
                time.sleep(0.05)
        if failed is not None:
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            t_kill = time.monotonic() + 10.0
            for p in procs:
                try:
                    p.wait(timeout=max(0.1, t_kill - time.monotonic()))
                except subprocess.TimeoutExpired:
                    p.kill()
                    p.wait()
        rcs = [p.returncode for p in procs]
        out0.seek(0)
        lines = [l for l in out0.read().splitlines() if l.startswith("{")]
    if failed is not None or any(rcs) or not lines:
        print(f"bench.py: rank exit codes {rcs}", file=sys.stderr)
        return next((rc for rc in rcs if rc and rc > 0), 1)
    print(lines[-1])
    return 0


def rank_env():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def device_for(local_rank, dry_run):
    """Rank -> device: one process per GPU, cuda:LOCAL_RANK (the CPU in a dry run)."""
    return torch.device("cpu") if dry_run else torch.device("cuda", local_rank)


def init_distributed(args, dev, world):
    """Process group of the run (None for one rank): RCCL ("nccl") on GPUs, gloo in a dry run."""
    if world == 1:
        return None
    import torch.distributed as dist
    if args.dry_run:
        dist.init_process_group("gloo")
    else:
        dist.init_process_group("nccl", device_id=dev)
    if dist.get_world_size() != world or (args.gpus != world):
        raise RuntimeError(f"--gpus {args.gpus}, WORLD_SIZE {world}, process group of {dist.get_world_size()} ranks disagree")
    return dist


# ---- CPU baseline ----------------------------------------------------------------------------------------------
def host_cpu():
    model, phys, logical = "unknown", set(), 0
    try:
        pid = cid = None
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "model name":
                model = v
            elif k == "processor":
                logical += 1
            elif k == "physical id":
                pid = v
            elif k == "core id":
                cid = v
                phys.add((pid, cid))
    except OSError:
        pass
    return {"model": model, "physical_cores": len(phys) or None, "logical_cpus": logical or os.cpu_count()}


def cpu_baseline_run(n_images, H, W, threads):
    """The CPU oracle (PyTorch-CPU fp32 restatement + C range coder) on n_images images with `threads` host threads,
    per stage under the reference's timing_analysis keys (manynets_compressor.py:394-401)."""
    import numpy as np
    from oracle import coders, manypriors_oracle as O
    from tests import synth
    torch.set_grad_enabled(False)
    torch.set_num_threads(threads)
    sd = synth.manypriors_state_dict(234)
    tf = O.entropy_table_float(sd, 256).contiguous()
    table_t = O.renorm_cast_cdf(tf, 16, O.ORDER_ROUND_THEN_ADD)
    table = table_t.numpy()
    x = synth.rand_image(99, n_images, H, W)
    O.analysis(x[:1, :, :64, :64], sd)                      # warm the thread pool
    st = dict.fromkeys(("nnenc", "getindices_lut", "make_cdf_tbl", "torchac_enc", "torchac_dec", "nndec"), 0.0)
    t_all = time.perf_counter()
    for i in range(n_images):
        t = time.perf_counter()
        yhat = O.quantise(O.analysis(x[i:i + 1], sd))
        st["nnenc"] += time.perf_counter() - t
        _, M, h, w = yhat.shape
        t = time.perf_counter()
        _, idx = O.prior_bits_lut_encode_layout(yhat, tf).min(0)
        st["getindices_lut"] += time.perf_counter() - t
        t = time.perf_counter()
        rows = O.gather_cdf(table_t, idx)                   # the materialised per-symbol rows of the reference's encode
        st["make_cdf_tbl"] += time.perf_counter() - t
        del rows
        sym = O.symbols_from_latent(yhat).flatten().numpy()
        idx32 = idx.reshape(-1).numpy().astype(np.int32)
        t = time.perf_counter()
        data = coders.range_encode_table(table, idx32, sym)
        st["torchac_enc"] += time.perf_counter() - t
        t = time.perf_counter()
        dec = coders.range_decode_table(table, idx32, data)
        st["torchac_dec"] += time.perf_counter() - t
        t = time.perf_counter()
        lat = O.latent_from_symbols(torch.from_numpy(dec).view(1, -1), M, h, w)
        O.synthesis(lat, sd).clamp(0, 1)
        st["nndec"] += time.perf_counter() - t
    dt = time.perf_counter() - t_all
    return {"value": n_images * H * W / 1e6 / dt, "unit": "megapixels/s", "cores": threads, "images": n_images,
            "seconds": dt, "stage_ms_per_image": {k: 1e3 * v / n_images for k, v in st.items()}}


def cpu_baseline(n_images, H, W):
    """Two bounded samples of the workload's images: at the host threads this process was given (the GPU box's CPU share;
    the machine's core counts are reported beside it) and at 8 threads."""
    share = torch.get_num_threads()
    runs = [cpu_baseline_run(n_images, H, W, share)]
    if share != 8:
        runs.append(cpu_baseline_run(max(2, n_images // 2), H, W, min(8, share)))
    torch.set_num_threads(share)
    main = runs[0]
    return {"value": main["value"], "unit": "megapixels/s", "cores": main["cores"], "kind": "port",
            "sample": f"{main['images']} of the {H}x{W} images, encode+decode incl. the materialised CDF rows, "
                      f"{main['seconds']:.1f} s at {main['cores']} threads" +
                      (f"; {runs[1]['images']} images, {runs[1]['seconds']:.1f} s at {runs[1]['cores']} threads" if len(runs) > 1 else ""),
            "host": host_cpu(), "runs": runs}


# ---- measured ceilings -------------------------------------------------------------------------------------------
def hbm_ceiling(dev, nbytes=1 << 31):
    """Device copy and fill rates of this GPU (GB/s of bytes moved), next to the 8 TB/s datasheet figure."""
    a = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    b = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    out = {}
    for name, fn, moved in (("copy", lambda: b.copy_(a), 2 * nbytes), ("fill", lambda: a.zero_(), nbytes)):
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out[name + "_gb_per_s"] = moved * 5 / (e0.elapsed_time(e1) * 1e-3) / 1e9
    return out


# ---- the runs -------------------------------------------------------------------------------------------------------
def run_dry(args, rank, world):
    """Launcher / statistics rehearsal on the CPU: every rank "codes" args.batch images with made-up but rank- and
    image-dependent numbers through StatsAccumulator and the all-reduce, and rank 0 prints the reduced sums."""
    from compression_amd.distributed import StatsAccumulator
    dev = device_for(int(os.environ.get("LOCAL_RANK", "0")), True)
    if os.environ.get("MPC_BENCH_FAIL_EARLY"):              # (launcher test: a rank that dies before the rendezvous)
        print(f"bench.py: rank {rank} fails on request", file=sys.stderr)
        return 3
    dist = init_distributed(args, dev, world)
    H, W = args.height, args.width
    t0 = time.perf_counter()
    for _ in range(args.warmup + args.steps):
        acc = StatsAccumulator()
        for i in range(args.batch):
            g = rank * args.batch + i
            acc.add_image(bits=1000.0 * (g + 1), pixels=H * W, sq_err=0.5 * (g + 1), nbytes=125 * (g + 1), seconds=0.01,
                          msssim=0.9 + 0.001 * g)
        acc.all_reduce()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if rank == 0:
        print(json.dumps({"metric": "dry run (no codec work)", "value": 0.0, "unit": "megapixels/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / max(args.steps, 1) * 1e3,
                          "device": str(dev), "backend": "gloo" if dist is not None else None,
                          "env": {k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR")},
                          "stats": acc.result()}))
    if dist is not None:
        dist.destroy_process_group()
    return 0


def make_inputs(args, rank, dev):
    """-> (model input, the image the reconstruction is scored against), both resident on the device."""
    from tests import synth
    B, H, W = args.batch, args.height, args.width
    if args.data == "noisy":
        clean = synth.smooth_image(1000 + rank, B, H, W)
        return synth.noisy_image(clean, 2000 + rank).to(dev), clean.to(dev)
    gen = synth.rand_image if args.data == "rand" else synth.smooth_image
    x = gen(1000 + rank, B, H, W).to(dev)
    return x, x


def run_rank(args):
    rank, world, local = rank_env()
    if args.dry_run:
        return run_dry(args, rank, world)
    dev = device_for(local, False)
    torch.cuda.set_device(dev)
    dist = init_distributed(args, dev, world)
    torch.set_grad_enabled(False)

    import compression_amd as ca
    from compression_amd import ops, pt_ms_ssim
    from compression_amd.distributed import StatsAccumulator
    from tests import synth

    testolina = args.arch.startswith("Testolina")
    if args.mode == "hp_codec":
        # the hyperprior models: g_a / g_s arithmetic as for the Manypriors workloads (h_a reads g_a's unrounded fp32 latent)
        model = ca.init_model(args.arch, dev, out_channel_N=192, out_channel_M=256, analysis_precision=args.analysis_precision,
                              synthesis_precision=args.synthesis_precision)
        sd = synth.transform_weights(237, 192, 256, testolina=testolina)
        sd.update(synth.hyperprior_weights(239, 192, 256))
        model.load_state_dict(sd)
    else:
        kw = dict(out_channel_N=192, out_channel_M=256, num_distributions=64, analysis_precision=args.analysis_precision,
                  synthesis_precision=args.synthesis_precision)
        model = ca.init_model("ManyPriors", dev, std_suffix="DC", **kw) if args.arch == "ManyPriors_DC" else ca.init_model(args.arch, dev, **kw)
        model.load_state_dict(synth.manypriors_state_dict(234, testolina=testolina))
    model.eval()
    x, target = make_inputs(args, rank, dev)
    B, _, H, W = x.shape
    mp_per_step = B * H * W / 1e6
    acc = StatsAccumulator(device=dev)
    state = {"seconds": 0.0}

    def serial_codec_step():
        streams, idx, h, w = model.encode_batch(x)
        job = model.decode_batch_begin(streams, idx, h, w)
        model.decode_batch_synthesis(job, score_against=target)
        xhat, sse = model.decode_batch_finish(job), job["sse"]
        nbytes = float(sum(len(s) for s in streams))
        acc.add_batch_device(sse, 3 * H * W, bits=8.0 * nbytes, pixels=B * H * W, nbytes=nbytes, seconds=state["seconds"], reset=True)
        return xhat

    def hyperprior_codec_step():
        streams = model.encode_batch(x)                     # [main string, z string, H, W] per image, bytes on the host
        recon = model.decode_batch(streams)
        xhat, sse = ops.clamp_sse_per_image(recon, target)
        nbytes = float(sum(len(s[0]) + len(s[1]) for s in streams))
        acc.add_batch_device(sse, 3 * H * W, bits=8.0 * nbytes, pixels=B * H * W, nbytes=nbytes, seconds=state["seconds"], reset=True)
        return xhat

    def forward_step():
        out = model(x)                                      # reads bpp back on the host, like the reference's test_dir
        xhat = out["reconstructed_image"]
        _, sse = ops.clamp_sse_per_image(xhat, target)
        bits = float(out["bpp_feature"]) * B * H * W
        acc.add_batch_device(sse, 3 * H * W, bits=bits, pixels=B * H * W, nbytes=bits / 8, seconds=state["seconds"],
                             side_bits=float(out["bpp_sidestring"]) * B * H * W, reset=True)
        return xhat

    # Throughput mode (default for the codec workloads): two batches in flight on two streams, see
    # compression_amd/pipeline.py.  Each batch's statistics are formed behind its g_s on its own stream.
    accs = {}

    def after_synthesis(job, strings, tag):
        xhat, sse = job["recon"], job["sse"]                # clamp and squared error come from the last layer's epilogue
        nbytes = float(sum(len(s) if isinstance(s, bytes) else len(s[0]) + len(s[1]) for s in strings))   # (hyperprior: [main, z, H, W])
        key = torch.cuda.current_stream().cuda_stream
        a = accs.get(key)
        if a is None:
            a = accs[key] = StatsAccumulator(device=dev)
        a.add_batch_device(sse, 3 * H * W, bits=8.0 * nbytes, pixels=B * H * W, nbytes=nbytes, seconds=state["seconds"], reset=True)
        state["acc"] = a
        return xhat

    if os.environ.get("MPC_BENCH_SYNTHESIS_BEHIND_LAYER"):    # (A/B switch, see manynets_compressor.SYNTHESIS_MAY_START_BEHIND_LAYER)
        from compression_amd import manynets_compressor
        manynets_compressor.SYNTHESIS_MAY_START_BEHIND_LAYER = int(os.environ["MPC_BENCH_SYNTHESIS_BEHIND_LAYER"])
    pipe = None
    if args.mode in ("codec", "hp_codec") and not args.serial:
        from compression_amd.pipeline import CodecPipeline
        pipe = CodecPipeline(model, dev, after_synthesis=after_synthesis,
                             early_first_layer=os.environ.get("MPC_BENCH_EARLY_FIRST_LAYER", "0") == "1",
                             decoder_behind=os.environ.get("MPC_BENCH_DECODER_BEHIND", "synthesis"))
    fpipe = None
    if args.mode == "forward" and not args.serial:
        from compression_amd.pipeline import ForwardPipeline
        fpipe = ForwardPipeline(model, dev, depth=args.forward_depth)

    def forward_stats(out, stream):
        """Statistics of one finished forward, formed on the stream that holds its tensors."""
        with torch.cuda.stream(stream):
            xhat = out["reconstructed_image"]
            _, sse = ops.clamp_sse_per_image(xhat, target)
            bits = out["host_bits"]["feature"]                # (host numbers: no blocking read of a device scalar)
            key = stream.cuda_stream
            a = accs.get(key)
            if a is None:
                a = accs[key] = StatsAccumulator(device=dev)
            a.add_batch_device(sse, 3 * H * W, bits=bits, pixels=B * H * W, nbytes=bits / 8, seconds=state["seconds"],
                               side_bits=out["host_bits"]["sidestring"], reset=True)
            state["acc"] = a
        return xhat

    def pipelined_forward_step():
        r = fpipe.submit(x)
        return None if r is None else forward_stats(r[0], fpipe.result_stream)

    def pipelined_codec_step():
        r = pipe.submit(x, target=target)
        return None if r is None else r[2]

    def drain():
        """-> x_hat of the batch still in flight, if any (the pipeline is empty afterwards)."""
        if fpipe is not None:
            last = None
            for out, _ in fpipe.flush():
                last = forward_stats(out, fpipe.result_stream)
            fpipe.synchronize()
            return last
        if pipe is None:
            return None
        r = pipe.flush()
        pipe.synchronize()
        return None if r is None else r[2]

    def keep_last(new, old):
        return old if new is None else new

    if args.mode == "hp_codec":
        step, serial_step = (hyperprior_codec_step if args.serial else pipelined_codec_step), hyperprior_codec_step
    elif args.mode == "forward":
        step, serial_step = (forward_step if args.serial else pipelined_forward_step), forward_step
    else:
        step, serial_step = (serial_codec_step if args.serial else pipelined_codec_step), serial_codec_step

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # Warm-up, with every launch between HIP events: this is where the dominant transform kernel is picked.  In the
    # timed region only that kernel's launches carry events (two per launch on the kernels' own stream): with events
    # round all ~20 launches of a step the host falls behind the GPU and the step grows by 0.6 ms and more.
    def conv_names(k):
        return [n for n in k if n.startswith("conv") or n.startswith("deconv5s2")]

    xhat = None
    wprof = ops.PROFILER = ops.Profiler()
    n_warm = max(args.warmup, 4 if (pipe is not None or fpipe is not None) else 1)   # (two batches per stream: torch's allocator has every block it will need)
    for _ in range(n_warm):
        t = time.perf_counter()
        xhat = keep_last(step(), xhat)                      # (kept, as in the timed loop: the allocator reaches its steady state here)
        torch.cuda.synchronize()
        state["seconds"] = time.perf_counter() - t
    xhat = keep_last(drain(), xhat)                                  # the timed region starts with nothing in flight
    ops.PROFILER = None
    wk = wprof.summary()
    dom = max(conv_names(wk), key=lambda n: wk[n]["ms"])
    prof = ops.PROFILER = ops.Profiler(only={dom})
    # A full pass of Python's cyclic collector over the ~10^6 objects that importing torch leaves behind takes 35-45 ms,
    # and one fell into most timed regions (a single step of 43 ms among steps of 8.2 ms).  Collect now, then move
    # everything that exists to the permanent generation: later passes only look at what the steps allocate.
    import gc
    gc.collect()
    gc.freeze()
    barrier()
    allocs0 = torch.cuda.memory_stats(dev).get("num_device_alloc", 0)
    ops.profile_mark(dev, 1)                                # (a one-thread kernel: the start of the timed region in a kernel trace)
    t0 = time.perf_counter()
    marks = [t0]
    for _ in range(args.steps):
        xhat = keep_last(step(), xhat)
        marks.append(time.perf_counter())                   # (host time when the step's call returned: shows a stalled step)
    xhat = keep_last(drain(), xhat)                                  # ... and ends with all of its K batches decoded
    # the job's one collective: the statistics vector of the last batch, summed over the ranks (the steps themselves
    # exchange nothing, so no rank waits for another inside the loop)
    state.get("acc", acc).all_reduce()
    barrier()
    dt = time.perf_counter() - t0
    ops.profile_mark(dev, 2)                                # ... and its end
    last = state.get("acc", acc).result()                   # (pipelined: the accumulator of the last batch's stream)
    device_allocs = torch.cuda.memory_stats(dev).get("num_device_alloc", 0) - allocs0   # hipMalloc calls inside the timed region
    ops.PROFILER = None
    timed_dom = prof.summary()[dom]
    # the other kernels' averages: the same steps once more, outside the timed region, every launch between events
    # (one batch at a time here: with two in flight a kernel's event pair also spans what it waited for or ran beside)
    aprof = ops.PROFILER = ops.Profiler()
    post_steps = min(args.steps, 20)
    for _ in range(post_steps):
        serial_step()
    ops.PROFILER = None
    kern = aprof.summary()
    for v in kern.values():
        v["steps"] = post_steps
    timed_dom["steps"] = args.steps
    kern[dom] = timed_dom
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # evaluation statistics of one step with MS-SSIM (outside the timed region: the codec step does not need it)
    state["seconds"] = dt / args.steps
    stats = None
    try:
        ms = pt_ms_ssim.ms_ssim(xhat, target, data_range=1.0, size_average=False).to(torch.float64).contiguous()
    except AssertionError:
        ms = None                                           # image too small for the 5 scales
    _, sse = ops.clamp_sse_per_image(xhat, target)
    rep = StatsAccumulator(device=dev)
    torch.cuda.synchronize()
    rep.add_batch_device(sse, 3 * H * W, bits=last["bits"] / world, pixels=B * H * W, nbytes=last["bytes"] / world,
                         seconds=state["seconds"], side_bits=last["side_bits"] / world, msssim=ms, reset=True)
    stats = rep.all_reduce().result()

    # the materialised per-symbol CDF rows ("CDF build") at the workload's size: measured on its own, the fused coder
    # never needs this tensor
    n_loc_probe = min(36864, B * (H // 16) * (W // 16))
    if args.mode != "hp_codec":                              # (the hyperprior models have no table of priors to gather from)
        idx_probe = torch.randint(0, 64, (n_loc_probe,), dtype=torch.int32, device=dev)
        model._ensure_tables()
        rows = ops.gather_cdf(model.entropy_table, idx_probe)
        p2 = ops.PROFILER = ops.Profiler()
        for _ in range(5):
            ops.gather_cdf(model.entropy_table, idx_probe, out=rows)
        ops.PROFILER = None
        kern.update(p2.summary())
        del rows

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return 0
    value = world * mp_per_step * args.steps / dt
    d = kern[dom]
    achieved = d["flops"] / (d["ms"] * 1e-3) / 1e12
    peak, passes = kernel_peak(dom)
    traffic, profiled_at = pmc_traffic(dom)
    kernels = {}
    for k, v in kern.items():
        e = {"calls_per_step": v["calls"] / (5 if k == "gather_cdf" else v.get("steps", args.steps)), "avg_ms": v["ms"] / v["calls"]}
        if v["flops"]:
            e["tflops"] = v["flops"] / (v["ms"] * 1e-3) / 1e12
        if v["bytes"]:
            e["gb_per_s"] = v["bytes"] / (v["ms"] * 1e-3) / 1e9
        kernels[k] = e
    metric = "eval forward megapixels/sec" if args.mode == "forward" else "encode+decode megapixels/sec"
    busy = pmc_mfma(dom) or {}
    out = {
        "metric": metric, "value": value, "unit": "megapixels/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": {"fp32": "f32", "f16x3": "f16 hi/lo split x3 (fp32-grade), f32 accumulate"}[args.analysis_precision]
                 + " (g_a); " + {"fp32": "f32", "fp16": "f16, f32 accumulate"}[args.synthesis_precision] + " (g_s)",
        "data": "synthetic",
        "config": {"workload": f"config {args.config}: {CONFIGS[args.config]['what']} per GPU, {args.arch} N=192 M=256 K=64, "
                               f"random-init weights with spread priors, {args.data} input",
                   "images_per_gpu": B, "height": H, "width": W, "bpp": stats["bpp"],
                   "parallelism": f"image-sharded x{world}, statistics all-reduce only",
                   "batches_in_flight": (args.forward_depth if fpipe is not None else 2) if (pipe is not None or fpipe is not None) else 1, "warmup_steps_run": n_warm,
                   "post_pass_steps": post_steps},
        "roofline": {"bound": "mfma", "kernel": dom, "achieved": achieved, "peak": peak,
                     "unit": "TFLOP/s", "frac": achieved / peak, "traffic": traffic, "traffic_profiled_at": profiled_at,
                     "mfma_passes_per_product": passes, "issued_tflops": achieved * passes,
                     "avg_launch_ms": d["ms"] / d["calls"], "launches_per_step": d["calls"] / args.steps,
                     # from the committed counter passes (profiles/): share of the chip's SIMD cycles with the matrix pipe
                     # busy, and the clock the kernel held while the counters ran
                     "mfma_busy": busy.get("mfma_busy"), "clock_ghz": busy.get("clock_ghz"),
                     "mfma_busy_profiled_at": busy.get("profiled_at"),
                     "mfma_busy_same_kernel_sources": busy.get("same_kernel_sources")},
        "kernels": kernels,
        "stats": stats,
    }
    if "gather_cdf" in kernels:
        out["cdf_build"] = {"bound": "hbm", "achieved": kernels["gather_cdf"]["gb_per_s"], "peak": PEAK_HBM_GBS,
                            "unit": "GB/s", "frac": kernels["gather_cdf"]["gb_per_s"] / PEAK_HBM_GBS,
                            "locations": n_loc_probe, "bytes": n_loc_probe * 256 * 257 * 2}
    out["hbm_ceiling"] = hbm_ceiling(dev)
    out["device_allocs_in_timed_region"] = int(device_allocs)
    out["host_ms_between_steps"] = [round((b - a) * 1e3, 3) for a, b in zip(marks, marks[1:])]
    if world == 1 and not args.no_cpu_baseline and args.mode != "hp_codec":     # (the CPU oracle's codec is the Manypriors one)
        out["cpu_baseline"] = cpu_baseline(args.cpu_images, 512, 768) if args.config == 3 else \
            cpu_baseline(max(2, min(args.cpu_images, int(16 * 393216 / (H * W)) or 2)), H, W)
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()
    return 0


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args, argv)
    return run_rank(args)


if __name__ == "__main__":
    sys.exit(main())
