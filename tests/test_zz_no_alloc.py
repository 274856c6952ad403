"""The allocation-free decode entry point (mpc_rans_decode_noalloc) against mpc_rans_decode, through the C ABI:
two calls in a row on one GPU, the second with a smaller input and another table, so that anything the first call
left in the library's static scratch would show."""
import pytest
import torch

from oracle import manypriors_oracle as O
from tests import synth

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
M, K = 256, 64


def _case(seed, n_img, h, w, dev, prior_seed):
    from compression_amd import ops
    sd = synth.prior_weights(prior_seed, K, M, spread=True)
    tf = O.entropy_table_float(sd, M).contiguous()
    table = O.renorm_cast_cdf(tf, 16, 0).to(dev)
    yl = synth.synthetic_latent(seed, n_img, M, h, w)
    _, idx = O.competition(O.prior_bits_lut(yl, tf))
    idx = idx.reshape(-1).to(torch.int32).to(dev)
    sym = O.symbols_from_latent(yl).reshape(-1).to(dev)
    n_loc = h * w
    n_streams, lps = ops.choose_streams(n_loc, M, 0.9 * n_loc * M)
    out, nbytes, status = ops.rans_encode(table, idx, sym, n_img, n_streams, lps)
    assert status.cpu().tolist() == [0] * n_img
    return table, idx, sym, out, nbytes, n_img, n_streams, lps


def test_noalloc_decode_matches_allocating_decode():
    assert torch.cuda.is_available(), "this test needs the MI355X"
    from compression_amd import MpcError, _lib, ops
    from compression_amd._lib import ptr
    dev = torch.device("cuda:0")
    # A context of its own, which has never run an allocating decode: its scratch is empty, so a variant that went
    # through the context's scratch after all would have to grow it, and that hipMalloc would show below.
    fresh = _lib.Context(0, torch.cuda.current_stream(dev).cuda_stream)
    for seed, n_img, h, w, prior_seed in ((21, 3, 24, 20, 235), (22, 1, 6, 10, 777)):       # second call: smaller, other table
        table, idx, sym, out, nbytes, n_img, n_streams, lps = _case(seed, n_img, h, w, dev, prior_seed)
        got = torch.full((n_img, h * w * M), -1, dtype=torch.int16, device=dev)
        got_st = torch.full((n_img,), 99, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        # free device memory as the driver sees it (hipMemGetInfo): covers hipMalloc calls inside the library, which
        # torch's own counters do not see
        free_before = torch.cuda.mem_get_info(dev)[0]
        torch_before = torch.cuda.memory_allocated(dev)
        Kt, Mt, Lt = table.shape
        fresh.call("mpc_rans_decode_noalloc", ptr(table), Kt, Mt, Lt, ptr(idx), ptr(out), out.shape[-1], ptr(nbytes),
                   n_img, h * w, n_streams, lps, ptr(got), ptr(got_st))
        torch.cuda.synchronize()
        assert torch.cuda.mem_get_info(dev)[0] == free_before
        assert torch.cuda.memory_allocated(dev) == torch_before
        # the allocating entry point, afterwards, for the comparison
        want, want_st = ops.rans_decode(table, idx, out, nbytes, n_img, n_streams, lps)
        assert got_st.cpu().tolist() == want_st.cpu().tolist() == [0] * n_img
        assert torch.equal(got, want) and torch.equal(got.view(-1), sym)
        # and through the wrapper on the shared context
        got.fill_(-1)
        ops.rans_decode_noalloc(table, idx, out, nbytes, n_img, n_streams, lps, got, got_st)
        assert torch.equal(got, want)
    fresh.close()
    # beyond the static scratch: refused with a clear error, nothing launched
    big = torch.zeros((K + 1, M, 257), dtype=torch.int16, device=dev)
    with pytest.raises(MpcError, match="static scratch"):
        ops.rans_decode_noalloc(big, idx, out, nbytes, n_img, n_streams, lps, got, got_st)
    with pytest.raises(MpcError, match="257"):
        ops.rans_decode_noalloc(table[:, :, :200].contiguous(), idx, out, nbytes, n_img, n_streams, lps, got, got_st)
