"""Round-trip and size checks of the C oracle coders (range coder restatement and the MRS1 rANS
container).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import coders, manypriors_oracle as O
from tests import synth

torch.set_grad_enabled(False)
M, K = 256, 64


@pytest.fixture(scope="module")
def setup():
    sd = synth.prior_weights(235, K, M, spread=True)
    tf = O.entropy_table_float(sd, M)
    table = O.renorm_cast_cdf(tf, 16, O.ORDER_ROUND_THEN_ADD).numpy()
    yl = synth.synthetic_latent(7, 1, M, 12, 10)
    total = O.prior_bits_lut(yl, tf)
    minbits, idx = O.competition(total)
    sym = O.symbols_from_latent(yl).flatten().numpy()
    return table, idx.reshape(-1).numpy().astype(np.int32), sym, float(minbits.sum())


def test_range_coder_roundtrip_rows_and_table(setup):
    table, idx, sym, est_bits = setup
    rows = O.gather_cdf(torch.from_numpy(table), torch.from_numpy(idx)).numpy().reshape(-1, 257)
    a = coders.range_encode_rows(rows, sym)
    b = coders.range_encode_table(table, idx, sym)
    assert a == b                                   # materialised rows and table+idx are the same coder
    assert np.array_equal(coders.range_decode_rows(rows, a), sym)
    assert np.array_equal(coders.range_decode_table(table, idx, b), sym)
    ideal = coders.ideal_bits(table, idx, sym)
    assert 0 <= len(a) * 8 - ideal < 64             # a range coder ends within a few bytes of the ideal
    assert abs(ideal - est_bits) / est_bits < 0.01  # int16 tables track the float estimate


@pytest.mark.parametrize("n_streams", [1, 2, 7, 120])
def test_mrs1_roundtrip_and_overhead(setup, n_streams):
    table, idx, sym, _ = setup
    n_loc = idx.size
    lps = -(-n_loc // n_streams)
    data = coders.mrs1_encode(table, idx, sym, n_streams, lps * M)
    assert np.array_equal(coders.mrs1_decode(table, idx, data), sym)
    ideal = coders.ideal_bits(table, idx, sym)
    overhead = len(data) - ideal / 8
    assert 0 <= overhead <= 16 + n_streams * 6 + 0.001 * ideal / 8 + 2


def test_mrs1_rejects_bad_input(setup):
    table, idx, sym, _ = setup
    bad = sym.copy()
    bad[5] = 300
    with pytest.raises(RuntimeError):
        coders.mrs1_encode(table, idx, bad, 4, 32 * M)
    data = bytearray(coders.mrs1_encode(table, idx, sym, 4, 32 * M))
    data[0] ^= 0xFF
    with pytest.raises(RuntimeError):
        coders.mrs1_decode(table, idx, bytes(data))


def test_edge_symbols_roundtrip():
    """Symbols 0 and 255 (whose upper edge is the implicit 2^16) and a near-deterministic row."""
    tf = torch.zeros(1, 2, 257)
    tf[0, 0] = torch.linspace(0, 1, 257)
    tf[0, 1, 129:] = 1.0                             # all mass on symbol 128
    table = O.renorm_cast_cdf(tf, 16, O.ORDER_ROUND_THEN_ADD).numpy()
    idx = np.zeros(6, np.int32)
    sym = np.array([0, 128, 255, 128, 17, 128, 255, 128, 0, 128, 3, 128], np.int16)
    for enc, dec in ((coders.range_encode_table, coders.range_decode_table),):
        assert np.array_equal(dec(table[:, :2], idx, enc(table[:, :2], idx, sym)), sym)
    data = coders.mrs1_encode(table[:, :2], idx, sym, 3, 4)
    assert np.array_equal(coders.mrs1_decode(table[:, :2], idx, data), sym)
