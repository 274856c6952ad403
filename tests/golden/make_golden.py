"""Generates tests/golden/*.npz by running the REFERENCE (imported from /root/reference, which exists
only in the build container) on the deterministic weights/inputs of tests/synth.py.

Run from the repo root:   python tests/golden/make_golden.py
The reference's sources never leave /root/reference; only inputs/outputs are stored here.

Import notes: the reference imports a few modules at module top that are absent from this image and
are never reached on the inference path (torchvision, ptflops, cv2, png, pytorch_msssim, colorspacious,
and its own missing common.extlibs.DISTS_pt).  They are replaced by empty import-only stubs.  `torchac`
is absent too, so the reference cannot produce int16 tables or bitstreams itself: the int16 tables
stored here are produced by the reference's in-repo `_renorm_cast_cdf_` (abstract_model.py:59-73) and,
for torchac's rounding order, by the published formula applied to the reference's float table.
"""
import hashlib
import os
import sys
import tempfile
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from tests import synth  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def import_reference():
    tv = _stub("torchvision")
    tv.datasets = _stub("torchvision.datasets")
    tv.transforms = _stub("torchvision.transforms")
    tv.utils = _stub("torchvision.utils", save_image=lambda *a, **k: None)
    for n in ("ptflops", "cv2", "png"):
        _stub(n)
    _stub("colorspacious", cspace_convert=None)

    class _M(torch.nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

    _stub("pytorch_msssim", MS_SSIM=_M, SSIM=_M)
    sys.path.insert(0, "/root/reference/src")
    import common.extlibs  # noqa: F401
    _stub("common.extlibs.DISTS_pt", DISTS=_M)
    os.chdir(tempfile.mkdtemp())          # pt_helpers creates ./tmp on import
    from compression.models import manynets_compressor, abstract_model, Balle2018PT_compressor, \
        testolina_compressor
    from common.libs import utilities
    utilities.compress_png = lambda *a, **k: False   # pypng absent: PNG side-string term disabled
    return manynets_compressor, abstract_model, Balle2018PT_compressor, testolina_compressor


def sha(t):
    a = t.detach().cpu().contiguous().numpy() if torch.is_tensor(t) else np.ascontiguousarray(t)
    return hashlib.sha256(a.tobytes()).hexdigest()


def load(model, sd):
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all(k.startswith("bitEstimator_z") for k in missing) or not missing, missing
    return model.train(False)


def metric_inputs(case):
    """(reconstruction, ground truth) pairs of the image-statistics fixture: an even and an odd size (the odd one
    exercises the padded 2x pooling), three images' worth of planes."""
    if case == 0:
        gt = synth.smooth_image(21, 2, 192, 256)
    else:
        gt = synth.smooth_image(22, 1, 211, 177)
    g = torch.Generator().manual_seed(100 + case)
    rec = (gt + 0.03 * torch.randn(gt.shape, generator=g)).clamp(0, 1)
    return rec, gt


def metrics_fixture():
    """MS-SSIM / SSIM / PSNR of the reference (src/common/extlibs/pt_ms_ssim.py, tests.py:393-412)."""
    import_reference()
    from common.extlibs import pt_ms_ssim
    torch.set_grad_enabled(False)
    out = {}
    for case in (0, 1):
        rec, gt = metric_inputs(case)
        win = pt_ms_ssim._fspecial_gauss_1d(11, 1.5).repeat(3, 1, 1, 1)
        s_pc, cs = pt_ms_ssim._ssim(rec, gt, data_range=1.0, win=win, size_average=False)
        mse = torch.nn.functional.mse_loss(rec, gt)
        out[f"ssim_pc{case}"] = s_pc.numpy()
        out[f"cs{case}"] = cs.numpy()
        out[f"ssim{case}"] = np.float32(pt_ms_ssim.ssim(rec, gt, data_range=1.0))
        out[f"msssim{case}"] = np.float32(pt_ms_ssim.ms_ssim(rec, gt, data_range=1.0, size_average=True))
        out[f"psnr{case}"] = np.float64((10 * (torch.log(1.0 / mse) / np.log(10))).item())
    np.savez_compressed(os.path.join(OUT, "image_metrics.npz"), **out)
    print({k: (float(v) if np.ndim(v) == 0 else v.shape) for k, v in out.items()})


TRAIN_CASES = (("pt", True, 71), ("noisy", False, 72))     # (tag, passthrough_ae, seed)


def train_inputs(seed):
    """Input batch and the idle-prior counters of the training-mode fixture: priors 3, 17 and 40 have been idle for 60
    calls, so the forward must force them onto the most expensive locations."""
    x = synth.smooth_image(seed, 2, 64, 96)
    last_use = np.zeros(64, dtype=int)
    last_use[[3, 17, 40]] = 60
    return x, last_use


def train_fixture():
    """Training-mode forward of the reference (manynets_compressor.py:514-530, 586-602) on seeded noise."""
    mc, am, b18, tc = import_reference()
    cpu = torch.device("cpu")
    torch.set_grad_enabled(False)
    N, M, K = 192, 256, 64
    sd = synth.manypriors_state_dict(234, N, M, K, spread=True)
    out = {}
    for tag, passthrough, seed in TRAIN_CASES:
        m = load(mc.Balle2017ManyPriors_ImageCompressor(out_channel_N=N, out_channel_M=M, num_distributions=K,
                                                        device=cpu, passthrough_ae=passthrough), sd)
        m.train(True)
        x, last_use = train_inputs(seed)
        m.dists_last_use[:] = last_use
        # the forward's first draw from torch's generator is the quantisation noise: the same seed and the same call
        # reproduce it
        torch.manual_seed(seed)
        noise = torch.nn.init.uniform_(torch.zeros(x.shape[0], M, x.shape[2] // 16, x.shape[3] // 16), -0.5, 0.5)
        torch.manual_seed(seed)
        np.random.seed(seed)
        res = m(x)
        out.update({
            f"{tag}_noise": noise.numpy(), f"{tag}_xhat": res["reconstructed_image"].numpy(),
            f"{tag}_visual_loss": float(res["visual_loss"]), f"{tag}_bpp_feature": float(res["bpp_feature"]),
            f"{tag}_bpp_sidestring": float(res["bpp_sidestring"]), f"{tag}_bpp": float(res["bpp"]),
            f"{tag}_used_dists": np.array(res["used_dists"]), f"{tag}_num_forced": np.array(res["num_forced_dists"]),
            f"{tag}_last_use_after": m.dists_last_use.copy(),
        })
    np.savez_compressed(os.path.join(OUT, "forward_train.npz"), **out)
    print({k: (float(v) if np.ndim(v) == 0 else v.shape) for k, v in out.items()})


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "metrics":
        return metrics_fixture()
    if len(sys.argv) > 1 and sys.argv[1] == "train":
        return train_fixture()
    mc, am, b18, tc = import_reference()
    cpu = torch.device("cpu")
    torch.set_grad_enabled(False)
    N, M, K = 192, 256, 64

    # ---------------- Manypriors standard model ----------------
    sd = synth.manypriors_state_dict(234, N, M, K, spread=True)
    m = load(mc.Balle2017ManyPriors_ImageCompressor(out_channel_N=N, out_channel_M=M, num_distributions=K,
                                                    device=cpu, passthrough_ae=True), sd)
    m.update_entropy_table()                      # returns False (no torchac) but sets entropy_table_float
    Tf = m.entropy_table_float.contiguous()       # [K,M,257]
    T_inrepo = am._renorm_cast_cdf_(Tf.unsqueeze(0).clone(), 16).squeeze(0)
    T_torchac = (Tf * (65536 - 257 + 1)).round().to(torch.int16) + torch.arange(257, dtype=torch.int16)
    m.entropy_table = m.entropy_table_cpu = T_torchac
    rows = torch.arange(0, K * M, 37)             # subsample of (k,c) rows kept verbatim
    tables = {
        "rows": rows.numpy(),
        "table_f_rows": Tf.view(K * M, 257)[rows].numpy(),
        "table_torchac_rows": T_torchac.view(K * M, 257)[rows].numpy(),
        "table_inrepo_rows": T_inrepo.view(K * M, 257)[rows].numpy(),
        "sha_table_f": sha(Tf), "sha_table_torchac": sha(T_torchac), "sha_table_inrepo": sha(T_inrepo),
    }
    np.savez_compressed(os.path.join(OUT, "tables_std.npz"), **tables)

    fw = {}
    for tag, x in (("a", synth.rand_image(1, 1, 64, 96)), ("b", synth.smooth_image(2, 2, 128, 128))):
        y = m.Encoder(x)
        out = m(x)
        yhat = torch.round(y)
        probs = m.bitEstimators(yhat + 0.5) - m.bitEstimators(yhat - 0.5)
        total = torch.sum(torch.clamp(-torch.log2(probs + 1e-10), 0, 50), dim=2)
        minbits, idx = total.min(0)
        recon = m.Decoder(yhat)
        fw.update({
            f"{tag}_y": y.numpy(), f"{tag}_total_bits": total.numpy(), f"{tag}_minbits": minbits.numpy(),
            f"{tag}_idx": idx.numpy().astype(np.int32), f"{tag}_recon": recon.numpy(),
            f"{tag}_xhat": out["reconstructed_image"].numpy(),
            f"{tag}_bpp_feature": float(out["bpp_feature"]), f"{tag}_bpp_sidestring": float(out["bpp_sidestring"]),
            f"{tag}_bpp": float(out["bpp"]), f"{tag}_visual_loss": float(out["visual_loss"]),
            f"{tag}_used_dists": np.array(out["used_dists"]),
        })
        if x.shape[0] == 1:
            sym, eidx, h, w = m.encode(x, entropy_coding=False)
            fw.update({f"{tag}_enc_sym": sym.numpy(), f"{tag}_enc_idx": eidx.numpy().astype(np.int32),
                       f"{tag}_enc_hw": np.array([h, w])})
    # per-layer activations of g_a / g_s on the small image (to localise a failing layer)
    x = synth.rand_image(1, 1, 64, 96)
    t = x
    for i in (1, 2, 3):
        t = getattr(m.Encoder, f"gdn{i}")(getattr(m.Encoder, f"conv{i}")(t))
        fw[f"a_enc_l{i}"] = t.numpy()
    t = torch.round(m.Encoder(x))
    for i in (1, 2, 3):
        t = getattr(m.Decoder, f"igdn{i}")(getattr(m.Decoder, f"deconv{i}")(t))
        fw[f"a_dec_l{i}"] = t.numpy()
    np.savez_compressed(os.path.join(OUT, "forward_std.npz"), **fw)

    # synthetic latents straight into the competition
    yl = synth.synthetic_latent(5, 2, M, 16, 16)
    probs = m.bitEstimators(yl + 0.5) - m.bitEstimators(yl - 0.5)
    total = torch.sum(torch.clamp(-torch.log2(probs + 1e-10), 0, 50), dim=2)
    minbits, idx = total.min(0)
    np.savez_compressed(os.path.join(OUT, "competition_latent.npz"), total_bits=total.numpy(),
                        minbits=minbits.numpy(), idx=idx.numpy().astype(np.int32),
                        n_used=np.array(idx.unique().numel()))

    # Kodak-shaped digest (config 1): hashes + statistics instead of tensors
    x = synth.rand_image(234, 1, 512, 768)
    out = m(x)
    y = m.Encoder(x)
    yhat = torch.round(y)
    sym, eidx, h, w = m.encode(x, entropy_coding=False)
    frac = (y - torch.floor(y) - 0.5).abs()
    np.savez_compressed(
        os.path.join(OUT, "kodak_digest.npz"),
        sha_sym=sha(sym), sha_idx=sha(eidx.to(torch.int32)), hw=np.array([h, w]),
        sym=sym.numpy(), idx=eidx.numpy().astype(np.int32),
        near_tie_1e4=float((frac < 1e-4).float().mean()),
        bpp_feature=float(out["bpp_feature"]), bpp_sidestring=float(out["bpp_sidestring"]),
        xhat_mean=float(out["reconstructed_image"].mean()),
        xhat_crop=out["reconstructed_image"][0, :, 100:132, 200:232].numpy(),
        visual_loss=float(out["visual_loss"]), n_used=np.array(len(out["used_dists"])))

    # degenerate default-init priors (one prior wins everywhere)
    sd0 = synth.manypriors_state_dict(234, N, M, K, spread=False)
    m0 = load(mc.Balle2017ManyPriors_ImageCompressor(out_channel_N=N, out_channel_M=M, num_distributions=K,
                                                     device=cpu, passthrough_ae=True), sd0)
    x = synth.rand_image(1, 1, 64, 96)
    out = m0(x)
    np.savez_compressed(os.path.join(OUT, "forward_default_init.npz"),
                        used_dists=np.array(out["used_dists"]), bpp_feature=float(out["bpp_feature"]),
                        bpp_sidestring=float(out["bpp_sidestring"]))

    # ---------------- Testolina decoder ----------------
    sdt = synth.manypriors_state_dict(235, N, M, K, spread=True, testolina=True)
    mt = load(tc.TestolinaManyPriors_ImageCompressor(out_channel_N=N, out_channel_M=M, num_distributions=K,
                                                     device=cpu, passthrough_ae=True), sdt)
    x = synth.smooth_image(3, 1, 64, 64)
    out = mt(x)
    np.savez_compressed(os.path.join(OUT, "forward_testolina.npz"), y=mt.Encoder(x).numpy(),
                        xhat=out["reconstructed_image"].numpy(), bpp_feature=float(out["bpp_feature"]))

    # ---------------- Hyperprior model ----------------
    sdh = synth.hyperprior_state_dict(236, N, M)
    mh = load(b18.Balle2018PTTFExp_ImageCompressor(out_channel_N=N, out_channel_M=M, device=cpu), sdh)
    x = synth.smooth_image(4, 1, 128, 192)
    out = mh(x)
    y = mh.Encoder(x)
    z = mh.priorEncoder(y)
    sigma = mh.priorDecoder(torch.round(z))
    mh.update_entropy_table()
    np.savez_compressed(os.path.join(OUT, "forward_hyperprior.npz"), y=y.numpy(), z=z.numpy(),
                        sigma=sigma.numpy(), xhat=out["reconstructed_image"].numpy(),
                        bpp_feature=float(out["bpp_feature"]), bpp_sidestring=float(out["bpp_sidestring"]),
                        bpp=float(out["bpp"]), z_table=mh.z_entropy_table.numpy(),
                        z_table_f=mh.z_entropy_table_float.numpy())
    fy, fz = mh.encode(x, entropy_coding=False)
    np.savez_compressed(os.path.join(OUT, "encode_hyperprior.npz"), y_sym=fy.numpy(), z_sym=fz.numpy())
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)) // 1024, "KiB")


if __name__ == "__main__":
    main()
