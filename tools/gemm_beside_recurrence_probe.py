"""Does a library GEMM beside a resident persistent recurrence launch make progress, or does it
wait for the CUs the recurrence holds?  And does a queue of them deadlock with back-to-back
recurrence launches?"""
import faulthandler, os, sys
faulthandler.dump_traceback_later(60, exit=True)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ctc_asr_amd import hip
hip.load(os.environ.get("CTCASR_LIB"))
F32 = torch.float32
T, B, H = 500, int(sys.argv[1]) if len(sys.argv) > 1 else 16, 1024
kind = sys.argv[2] if len(sys.argv) > 2 else 'bf16'
launches = int(sys.argv[3]) if len(sys.argv) > 3 else 1
g = torch.Generator(device='cuda').manual_seed(0)
xw = torch.randn(T, B, 2, 4 * H, device='cuda', generator=g) * 0.5
w_hh = torch.randn(2, 4 * H, H, device='cuda', generator=g) / 32
dy = torch.randn(T, B, 2 * H, device='cuda', generator=g)
wt = hip.transpose_batched(w_hh)
y, reserve, ws = hip.rnn_fwd('lstm', xw, w_hh)
dxw = hip.rnn_bwd('lstm', dy, y, wt, reserve, workspace=ws)
torch.cuda.synchronize()
R = T * B
rows = R // 2
if kind == 'bf16':
    a = torch.randn(R * 6, 8192, device='cuda', dtype=torch.bfloat16)
    b = torch.randn(R * 6, 2048, device='cuda', dtype=torch.bfloat16)
    out = torch.zeros(4096, 2048, device='cuda')
    gemm = lambda: torch.addmm(out, a[:6 * rows, :4096].t(), b[:6 * rows], out_dtype=F32, out=out)
else:
    a = torch.randn(R, 8192, device='cuda')
    b = torch.randn(R, 2048, device='cuda')
    out = torch.zeros(4096, 2048, device='cuda')
    gemm = lambda: out.addmm_(a[:rows, :4096].t(), b[:rows])
for _ in range(3):
    gemm()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record(); gemm(); e.record(); torch.cuda.synchronize()
print('gemm alone {:.3f} ms'.format(s.elapsed_time(e)))
side = torch.cuda.Stream()
main = torch.cuda.current_stream()
for trial in range(3):
    torch.cuda.synchronize()
    ready = torch.cuda.Event(); ready.record()
    r0, r1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    r0.record()
    bounds = [T * (launches - c) // launches for c in range(launches + 1)]
    for c in range(launches):
        hip.rnn_bwd('lstm', dy, y, wt, reserve, dxw=dxw, workspace=ws, ticket=trial * 8 + c + 1,
                    steps=(bounds[c + 1], bounds[c]))
    r1.record()
    with torch.cuda.stream(side):
        side.wait_event(ready)
        hip.rnn_resident_gate('lstm', ws, T, B, H, trial * 8 + 1, 300)
        g0 = torch.cuda.Event(enable_timing=True); g0.record(side)
        ends = []
        for _ in range(6 if launches > 1 else 1):
            gemm()
            ends.append(torch.cuda.Event(enable_timing=True)); ends[-1].record(side)
    torch.cuda.synchronize()
    print('trial {}: recurrence {:.3f} ms; gemm(s) beside it finished at {} ms after the gate'.format(
        trial, r0.elapsed_time(r1), [round(g0.elapsed_time(x), 3) for x in ends]), flush=True)
hip.rnn_poll_error('lstm', ws, T, B, H)
print('done')
