"""Pipelined codec on one batch, N times: every result against the one-batch-at-a-time result.  Reports which images'
byte strings / reconstructions differ and where the decoded symbols differ."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.getcwd())
import compression_amd as ca
from compression_amd import ops
from compression_amd.pipeline import CodecPipeline
from tests import synth
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
serial = len(sys.argv) > 2 and sys.argv[2] == "serial"
if len(sys.argv) > 2 and sys.argv[2] == "stress":
    from compression_amd import manynets_compressor
    manynets_compressor.SYNTHESIS_MAY_START_BEHIND_LAYER = 2
m = ca.init_model("ManyPriors", dev, out_channel_N=192, out_channel_M=256, num_distributions=64, analysis_precision="f16x3")
m.load_state_dict(synth.manypriors_state_dict(234)); m.eval()
x = synth.rand_image(1000, 24, 512, 768).to(dev)
ref_s, idx, h, w = m.encode_batch(x)
ref_r = m.decode_batch(ref_s, idx, h, w).clone()
ref_idx = idx.clone()
ref_sym = ops.rans_decode_from_bytes(m.entropy_table, idx.reshape(-1).to(torch.int32), ref_s).clone()
torch.cuda.synchronize()
bad = 0
def check(it, strings, recon):
    global bad
    ds = [i for i in range(len(strings)) if strings[i] != ref_s[i]]
    dr = (recon != ref_r).flatten(1).any(1).nonzero().flatten().tolist()
    if ds or dr:
        bad += 1
        msg = f"step {it}: strings differ in images {ds}, recon differs in images {dr}"
        for i in ds[:2]:
            try:
                sym = ops.rans_decode_from_bytes(m.entropy_table, ref_idx[i].reshape(-1).to(torch.int32), [strings[i]])
                d = (sym.view(-1) != ref_sym[i].view(-1)).nonzero().flatten()
                loc = (d // 256).unique()
                msg += f"; image {i}: {d.numel()} symbols differ at {loc.numel()} locations {[(int(l) // w, int(l) % w) for l in loc[:8]]} len {len(strings[i])} vs {len(ref_s[i])}"
            except Exception as e:
                msg += f"; image {i}: decode with the reference indices failed ({e})"
        for i in dr[:2]:
            d = (recon[i] != ref_r[i]).nonzero()
            msg += f"; recon image {i}: {d.shape[0]} values, rows {int(d[:,1].min())}-{int(d[:,1].max())} cols {int(d[:,2].min())}-{int(d[:,2].max())} max {float((recon[i]-ref_r[i]).abs().max()):.3g}"
        print(msg, flush=True)
if serial:
    for it in range(N):
        s, i2, _, _ = m.encode_batch(x)
        r = m.decode_batch(s, i2, h, w)
        check(it, s, r)
else:
    pipe = CodecPipeline(m, dev)
    for it in range(N):
        r = pipe.submit(x, tag=it)
        if r is not None:
            pipe.synchronize()
            check(r[3], r[0], r[1])
        if it % 1000 == 999:
            print("step", it + 1, "differing so far:", bad, flush=True)   # (a long run must not look hung)
    r = pipe.flush(); pipe.synchronize(); check(r[3], r[0], r[1])
print("steps that differ:", bad, "of", N, flush=True)
