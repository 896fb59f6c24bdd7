"""GraphedForward: the inference forward replayed as one hipGraph gives the same bits as the eager forward, for new inputs too, and
refuses what it was not captured for."""
import pytest
import torch

pytestmark = pytest.mark.gpu

import wav2sleep_amd as W  # noqa: E402
from oracle import wav2sleep_oracle as O  # noqa: E402  (input generator only)

SM = {'ABD': 'ABD', 'THX': 'THX', 'ECG': 'ECG', 'PPG': 'PPG'}


def _model():
    torch.manual_seed(3)
    return W.Wav2Sleep(W.SignalEncoders(SM, 128, 'gelu', norm='instance', causal=False, chunk_causal=False),
                       W.MultiModalAttentionEmbedder(128, layers=2, dropout=0.1, dim_ff=512, nhead=8),
                       W.SequenceCNN(128, dropout=0.1, norm='layer', causal=False, num_layers=2, kernel_size=7, num_dilations=6), 4).to('cuda').eval()


def test_graphed_forward_replays_the_eager_forward_bit_for_bit():
    model = _model()
    cfg = O.ModelConfig(signal_map=SM, num_classes=4)
    x, _ = O.make_inputs(cfg, 2, 40, seed=1, missing={'PPG': [1]})
    x = {k: v.cuda() for k, v in x.items()}
    fwd = W.GraphedForward(model, x)
    with torch.no_grad():
        assert torch.equal(fwd(x), model(x))
        for seed, missing in ((2, None), (3, {'ECG': [0], 'ABD': [1]})):
            x2, _ = O.make_inputs(cfg, 2, 40, seed=seed, missing=missing)
            x2 = {k: v.cuda() for k, v in x2.items()}
            assert torch.equal(fwd(x2), model(x2))
    with pytest.raises(ValueError):
        fwd({k: v[:1] for k, v in x.items()})            # another batch size
    with pytest.raises(ValueError):
        fwd({'ECG': x['ECG']})                            # another signal set
    with torch.no_grad():
        next(model.parameters()).mul_(1.5)
    with pytest.raises(RuntimeError):
        fwd(x)                                            # weights changed: the packed copies in the graph are stale
    fwd.recapture()
    with torch.no_grad():
        assert torch.equal(fwd(x), model(x))
    with pytest.raises(ValueError):
        W.GraphedForward(model.train(), x)
