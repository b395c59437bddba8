"""Pin the explicit GRU/LSTM recurrences (oracle/rnn_ref.py) against torch.nn.GRU / LSTM."""
import pytest
import torch

from oracle import rnn_ref


@pytest.mark.parametrize("layers", [1, 2])
def test_gru_matches_torch(layers):
    torch.manual_seed(0)
    m = torch.nn.GRU(6, 5, num_layers=layers, batch_first=True, bidirectional=True)
    x = torch.randn(3, 7, 6)
    ref, _ = m(x)
    got = rnn_ref.gru(dict(m.state_dict()), "", x, num_layers=layers, bidirectional=True)
    assert torch.allclose(got, ref, atol=1e-6)


def test_lstm_matches_torch():
    torch.manual_seed(0)
    m = torch.nn.LSTM(6, 9, batch_first=True, bidirectional=True)
    x = torch.randn(4, 3, 6)
    ref, _ = m(x)
    got = rnn_ref.lstm(dict(m.state_dict()), "", x, num_layers=1, bidirectional=True)
    assert torch.allclose(got, ref, atol=1e-6)


def test_fast_gru_equals_loops():
    torch.manual_seed(1)
    m = torch.nn.GRU(6, 4, num_layers=2, batch_first=True, bidirectional=True)
    x = torch.randn(2, 9, 6)
    P = dict(m.state_dict())
    a = rnn_ref._gru_loops(P, "", x, 2, True)
    b = rnn_ref.gru_fast(P, "", x, 2, True)
    assert torch.allclose(a, b, atol=1e-6)
