import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from ctc_asr_amd import hip
from test_gpu_dgrad16 import publish, rel_errors
H = 1024; DEV = 'cuda'
print('lib', os.environ.get('CTCASR_LIB'))
g = torch.Generator(device=DEV).manual_seed(1)
# 1. pack vs torch
n = 256
w = torch.randn(8 * H, n, device=DEV, generator=g)
packed = hip.dgrad16_pack_weights(w, H, 2048.0)
ws_ = (w * 2048.0).view(2, 4, 64, 2, 4, 2, n // 16, 16)      # dir gate P m q u1 nt col
h1 = ws_.half(); h2 = (ws_ - h1.float()).half()
pc = torch.stack([h1, h2], 0)       # piece dir gate P m q u1 nt col
# -> [dir P m][nt][piece][q][col][u1 gate]
refp = pc.permute(1, 3, 4, 7, 0, 5, 8, 6, 2).contiguous().view(torch.uint8).view(-1)
print('pack equal', torch.equal(refp, packed), float((refp != packed).float().mean()))
# 2. single-entry probes
for (T, B) in ((1, 16), (2, 32), (3, 1)):
    for trial in range(3):
        dxw = torch.zeros(T, B, 2, 4 * H, device=DEV)
        t, b, d, col = [int(torch.randint(0, hi, (1,))) for hi in (T, B, 2, 4 * H)]
        dxw[t, b, d, col] = 3.0
        ws = publish(hip, dxw)
        got = hip.dgrad16_blockscaled(ws, T, B, H, packed, 2048.0, n)
        ref = dxw.view(T * B, 8 * H).double() @ w.double()
        nz = got.abs().amax(dim=1).nonzero().flatten().tolist()
        print('T', T, 'B', B, 'entry', (t, b, d, col), 'row', t * B + b, 'nonzero rows', nz[:8], len(nz),
              'err', rel_errors(got, ref), 'max', float(got.abs().max()), float(ref.abs().max()))
# 3. dense small
T, B = 4, 32
dxw = torch.randn(T, B, 2, 4 * H, device=DEV, generator=g)
ws = publish(hip, dxw)
got = hip.dgrad16_blockscaled(ws, T, B, H, packed, 2048.0, n)
ref = dxw.view(T * B, 8 * H).double() @ w.double()
print('dense', rel_errors(got, ref), float(got.abs().max()), float(ref.abs().max()))
for dirs in ((0, 1), (1, 2)):
    got = hip.dgrad16_blockscaled(ws, T, B, H, packed, 2048.0, n, dirs=dirs)
    r = dxw[:, :, dirs[0]].reshape(T * B, 4 * H).double() @ w[dirs[0] * 4 * H:(dirs[0] + 1) * 4 * H].double()
    print('dir', dirs, rel_errors(got, r), float(got.abs().max()))
