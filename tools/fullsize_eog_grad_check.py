"""BASELINE configs[3] at full size (EOG-L + EOG-R at 4096 samples per epoch: ten-block encoders, 3.9 M samples per recording, 5 classes),
B = 1: loss and every gradient tensor against the oracle's autograd, and bit-reproducibility of two runs.  (A stand-alone check, not
part of the pytest suite: `python tools/fullsize_eog_grad_check.py` on the GPU box; result quoted in DESIGN.md section 1.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import wav2sleep_amd as W
from oracle import wav2sleep_oracle as O
DEV = 'cuda'
sm = {'EOG-L': 'EOG-L', 'EOG-R': 'EOG-R'}
torch.manual_seed(42)
model = W.Wav2Sleep(W.SignalEncoders(sm, 128, 'gelu', norm='instance', causal=False, chunk_causal=False),
                    W.MultiModalAttentionEmbedder(128, layers=2, dropout=0.0, dim_ff=512, nhead=8),
                    W.SequenceCNN(128, dropout=0.0, norm='layer', causal=False, num_layers=2, kernel_size=7, num_dilations=6), 5).to(DEV).train()
cfg = O.ModelConfig(signal_map=sm, num_classes=5)
sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
x, y = O.make_inputs(cfg, 1, 960, seed=321)
runs = []
for _ in range(2):
    model.zero_grad(set_to_none=True)
    logits = model({k: v.to(DEV) for k, v in x.items()})
    loss = torch.nn.functional.cross_entropy(logits.reshape(-1, 5), y.to(DEV).reshape(-1).long(), ignore_index=-1)
    loss.backward()
    torch.cuda.synchronize()
    runs.append(model._flat_grad.clone())
print('bit-reproducible:', torch.equal(runs[0], runs[1]))
want_loss, _, want = O.loss_and_grads(sd, cfg, x, y)
worst = ('', 0.0)
for name, p in model.named_parameters():
    rel = float((p.grad.detach().cpu() - want[name]).norm() / (want[name].norm() + 1e-20))
    worst = (name, rel) if rel > worst[1] else worst
print(f'full-size gradients EOG pair B=1: loss {float(loss):.6f} vs oracle {want_loss:.6f}; worst tensor {worst[0]} rel-L2 {worst[1]:.2e}')
