python -m pytest tests/test_gpu_kernels.py -x -q -k "conv_s12" 2>&1 | tail -6
python - <<'PY'
import torch, numpy as np, sys
sys.path.insert(0,'.')
from ctc_asr_amd import hip, split_gemm
hip.load()
for (B,T,F,C) in ((32,999//2+1,40,32),(16,500,40,32),(32,500,20,96)):
    x=torch.rand(B,T,F,32,device='cuda')*20; w=torch.randn(C,32,11,21,device='cuda')*0.05; b=torch.zeros(C,device='cuda')
    p=hip.conv_s12_pack_weights(w); p16=hip.conv_s12_pack_weights16(w)
    for name,fn in (('fp32',lambda: hip.conv_s12_fwd(x,p,C,b,relu_cutoff=20.0)),('fp16x3',lambda: hip.conv_s12_fwd16(x,2.0**11,p16,C,b,relu_cutoff=20.0)),('pack16',lambda: hip.conv_s12_pack_weights16(w,p16))):
        fn(); torch.cuda.synchronize()
        s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10): fn()
        e.record(); torch.cuda.synchronize()
        ms=s.elapsed_time(e)/10
        fl=2.0*B*T*(F//2)*C*32*231
        print((B,T,F,C),name,'%.3f ms'%ms, '%.0f TFLOP/s'%(fl/ms/1e9) if name!='pack16' else '')
PY
