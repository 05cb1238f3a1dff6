"""Which operand layout should the fp16 weight-gradient GEMMs of a recurrent layer use?

Today (split_gemm.wgrad16): the pieces of dxw and of the layer's input / output are row-major
[rows, 3, cols] - rows = the summation axis - so the library runs its TN kernel (both operands
K-slow, `Cijk_Ailk_Bjlk_..._MT256x256x32`).  Alternative: K-contiguous operands ([cols, 3 * rows],
written so by transposing split / copy kernels), the library's NT kernel - the one the forward
projections use - and W_ih's and W_hh's products of a direction in ONE call ([4096 x 3072]).

Measured alone and on the side stream beside the half-chip fp16-pipe backward recurrence (C3:
T' = 500, B = 32, H = 1024; a launch = 250 steps), the shapes of one (launch, direction):
    python tools/wgrad_layout_probe.py [batch]"""
import faulthandler
import os
import sys

faulthandler.dump_traceback_later(240, exit=True)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ctc_asr_amd import hip

hip.load(os.environ.get('CTCASR_LIB'))
F32, F16 = torch.float32, torch.float16
T, B, H = 500, int(sys.argv[1]) if len(sys.argv) > 1 else 32, 1024
G4 = 4 * H
gen = torch.Generator(device='cuda').manual_seed(0)
xw = torch.randn(T, B, 2, G4, device='cuda', generator=gen) * 0.5
w_hh = torch.randn(2, G4, H, device='cuda', generator=gen) / 32
dy = torch.randn(T, B, 2 * H, device='cuda', generator=gen)
wt = hip.transpose_batched(w_hh)
flags = hip.RNN_DEFAULT | hip.RNN_F16 | hip.RNN_XCD_SPLIT
y, reserve, ws = hip.rnn_fwd('lstm', xw, w_hh, flags=flags)
dxw = hip.rnn_bwd('lstm', dy, y, wt, reserve, workspace=ws, flags=flags)
torch.cuda.synchronize()

rows = T * B // 2                       # rows of one launch
K = 3 * rows


def rnd(*shape):
    return (torch.randn(*shape, device='cuda', generator=gen) * 0.25).to(F16)


# today's layout
d_tn = rnd(K, G4)                       # [3 rows, 4096]   (one direction's columns)
x_tn = rnd(K, 2048)
h_tn = rnd(K, 2048)                     # both directions' columns, one half used
# K-contiguous layout
d_nt = rnd(G4, K)
z_nt = rnd(4096, K)                     # [y0 shifted | x | y1 shifted] rows
# data gradient: dxw rows [rows, 3 x 4096 (one direction)] x W_ih^T pieces [2048, 3 x 4096]
dd = rnd(rows, 3 * G4)
wd = rnd(2048, 3 * G4)
dd_full = rnd(2 * rows, 3 * 2 * G4)
wd_full = rnd(2048, 3 * 2 * G4)

cases = {
    'TN W_ih [4096 x 2048], K = {}'.format(K): lambda: torch.mm(d_tn.t(), x_tn, out_dtype=F32),
    'TN W_hh [4096 x 1024]': lambda: torch.mm(d_tn.t(), h_tn[:, :1024], out_dtype=F32),
    'NT W_ih [4096 x 2048]': lambda: torch.mm(d_nt, z_nt[1024:3072].t(), out_dtype=F32),
    'NT W_hh [4096 x 1024]': lambda: torch.mm(d_nt, z_nt[:1024].t(), out_dtype=F32),
    'NT merged [4096 x 3072]': lambda: torch.mm(d_nt, z_nt[:3072].t(), out_dtype=F32),
    'NT merged, swapped [3072 x 4096]': lambda: torch.mm(z_nt[:3072], d_nt.t(), out_dtype=F32),
    'dx half [rows/2 x 2048], K = 3 x 4096': lambda: torch.mm(dd, wd.t(), out_dtype=F32),
    'dx full [rows x 2048], K = 3 x 8192': lambda: torch.mm(dd_full, wd_full.t(), out_dtype=F32),
}


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


alone = {name: timed(fn) for name, fn in cases.items()}
side = torch.cuda.Stream()
ticket = 0
print('batch {}: one launch = {} rows'.format(B, rows))
print('{:45s} {:>9s} {:>12s} {:>16s}'.format('case', 'alone ms', 'beside ms', 'recurrence us/step'))
for name, fn in cases.items():
    # queue enough calls to last the whole launch (250 steps ~ 2.2 ms), time the ones that finish
    # before the recurrence does
    n = max(2, int(2.0 / alone[name]))
    best = []
    for trial in range(3):
        ticket += 1
        torch.cuda.synchronize()
        ready = torch.cuda.Event()
        ready.record()
        r0, r1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        r0.record()
        hip.rnn_bwd('lstm', dy, y, wt, reserve, dxw=dxw, workspace=ws, ticket=ticket,
                    steps=(T // 2, T), flags=flags)
        r1.record()
        with torch.cuda.stream(side):
            side.wait_event(ready)
            hip.rnn_resident_gate('lstm', ws, T, B, H, ticket, 300)
            marks = [torch.cuda.Event(enable_timing=True)]
            marks[0].record(side)
            for _ in range(n):
                fn()
                marks.append(torch.cuda.Event(enable_timing=True))
                marks[-1].record(side)
        torch.cuda.synchronize()
        rec_ms = r0.elapsed_time(r1)
        inside = [marks[i].elapsed_time(marks[i + 1]) for i in range(n)
                  if marks[0].elapsed_time(marks[i + 1]) < rec_ms - 0.05]
        if inside:
            best.append((sum(inside) / len(inside), rec_ms * 1e3 / (T // 2)))
    if best:
        g, r = min(best)
        print('{:45s} {:9.3f} {:12.3f} {:16.2f}'.format(name, alone[name], g, r), flush=True)
    else:
        print('{:45s} {:9.3f} {:>12s}'.format(name, alone[name], 'none inside'), flush=True)
hip.rnn_poll_error('lstm', ws, T, B, H)
print('done')
