import os, sys, json, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from ctc_asr_amd import hip, split_gemm
from ctc_asr_amd.model import CTCModel, ModelConfig
hip.load()
# 1. kernel check incl. strided out
x = torch.randn(300, 64, device='cuda') * torch.logspace(-20, 10, 300, device='cuda')[:, None]
s = hip.split_bf16(x, (0, 1, 2))
a1 = x.bfloat16(); r = x - a1.float(); a2 = r.bfloat16(); r2 = r - a2.float(); a3 = r2.bfloat16()
assert torch.equal(s[:, 0], a1) and torch.equal(s[:, 1], a2) and torch.equal(s[:, 2], a3)
rec = s.float().sum(1)
print('reconstruction max rel err', float(((rec - x).abs() / x.abs().clamp_min(1e-38)).max()))
big = torch.zeros(300, 3, 128, dtype=torch.bfloat16, device='cuda')
hip.split_bf16(x[10:50, :32], (0, 1, 2), out=big[10:50, :, 64:96])
assert torch.equal(big[10:50, :, 64:96], s[10:50, :, :32]) and float(big[:, :, :64].abs().sum()) == 0
print('split kernel ok')
# 2. model-level: gradients with and without the split, mid-size C3-like layer stack
def run(flag, batch, t_frames, cell='lstm', layers=2):
    os.environ['CTCASR_SPLIT_GEMM'] = flag
    cfg = ModelConfig(used_model='ds2', conv_filters=(32, 32, 96), num_units_dense=2048,
                      num_layers_rnn=layers, num_units_rnn=1024, rnn_cell=cell, cudnn=True,
                      dense_dropout_rate=0.0, conv_dropout_rate=0.0)
    m = CTCModel(cfg, 'cuda', seed=5)
    rng = np.random.default_rng(1)
    feats = torch.tensor(rng.normal(size=(batch, t_frames, 80)).astype(np.float32), device='cuda')
    flen = torch.full((batch,), t_frames, dtype=torch.int32)
    labels = [list(rng.integers(1, 28, size=20)) for _ in range(batch)]
    loss = m.forward_backward(feats, flen, labels)
    torch.cuda.synchronize()
    return float(loss), m.last_logits.clone(), m.arena.grad.clone(), m
for cell, batch in (('lstm', 32), ('lstm', 16), ('gru', 32)):
    l0, lg0, g0, m0 = run('0', batch, 400, cell)
    l1, lg1, g1, m1 = run('1', batch, 400, cell)
    print(cell, batch, 'loss', l0, l1, 'logits max delta', float((lg0 - lg1).abs().max()),
          'grad max delta', float((g0 - g1).abs().max()), 'grad max', float(g0.abs().max()),
          'rel rms', float((g0 - g1).pow(2).sum().sqrt() / g0.pow(2).sum().sqrt()))
    for name, a, b in m0.arena.layer_slices:
        d = (g0[a:b] - g1[a:b]).pow(2).sum().sqrt() / g0[a:b].pow(2).sum().sqrt().clamp_min(1e-30)
        print('   ', name, float(d))
