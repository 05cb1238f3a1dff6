import sys, os, torch
sys.path.insert(0, os.getcwd())
from ctc_asr_amd import hip
hip.load()
def timed(fn, reps=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for B in (32, 16):
    T = 500
    x = torch.randn(B, T, 40, 32, device='cuda'); w = torch.randn(32, 32, 11, 21, device='cuda') * 0.05
    packed = hip.conv_s12_pack_weights(w)
    y = hip.conv_s12_fwd(x, packed, 32, relu_cutoff=20.0, time_major=True)
    dy = torch.randn_like(y); db = torch.zeros(32, device='cuda')
    def unfused():
        dz = hip.bias_act_bwd(y, dy, 20.0, 0.0, db)
        hip.conv_s12_wrw(dz, x, time_major=True); hip.conv_s12_bwd_data(dz, packed, time_major=True)
    def fused():
        hip.conv_s12_wrw(dy, x, time_major=True, act=y, relu_cutoff=20.0, dbias=db)
        hip.conv_s12_bwd_data(dy, packed, time_major=True, act=y, relu_cutoff=20.0)
    print('B', B, 'layer 2: unfused {:.0f} us (mask {:.0f} + wrw {:.0f} + bwd_data {:.0f}) | fused {:.0f} us (wrw {:.0f} + bwd_data {:.0f})'.format(
        timed(unfused), timed(lambda: hip.bias_act_bwd(y, dy, 20.0, 0.0, db)), timed(lambda: hip.conv_s12_wrw(dy, x, time_major=True)),
        timed(lambda: hip.conv_s12_bwd_data(dy, packed, time_major=True)), timed(fused),
        timed(lambda: hip.conv_s12_wrw(dy, x, time_major=True, act=y, relu_cutoff=20.0, dbias=db)),
        timed(lambda: hip.conv_s12_bwd_data(dy, packed, time_major=True, act=y, relu_cutoff=20.0))))
    feats = torch.randn(B, 999, 80, device='cuda'); w0 = torch.randn(32, 1, 11, 41, device='cuda') * 0.1
    y0 = hip.conv0_fwd(feats, w0, relu_cutoff=20.0); dy0 = torch.randn_like(y0)
    print('B', B, 'conv0: mask {:.0f} + wrw {:.0f} | fused wrw {:.0f}'.format(
        timed(lambda: hip.bias_act_bwd(y0, dy0, 20.0, 0.0, db)), timed(lambda: hip.conv0_wrw(dy0, feats)),
        timed(lambda: hip.conv0_wrw(dy0, feats, act=y0, relu_cutoff=20.0, dbias=db))))
