python -m pytest tests/test_gpu_kernels.py -x -q -k "conv_s12" 2>&1 | tail -8
python - <<'PY'
import torch, numpy as np, sys
sys.path.insert(0,'.')
from ctc_asr_amd import hip
hip.load()
for (B,T,F,C) in ((32,500,40,32),(32,500,20,96)):
    dz=torch.randn(B,T,F//2,C,device='cuda'); w=torch.randn(C,32,11,21,device='cuda')*0.05
    p=hip.conv_s12_pack_weights(w); p16=hip.conv_s12_pack_weights16(w)
    for name,fn in (('fp32',lambda: hip.conv_s12_bwd_data(dz,p)),('fp16x3',lambda: hip.conv_s12_bwd_data16(dz,p16))):
        fn(); torch.cuda.synchronize()
        s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10): fn()
        e.record(); torch.cuda.synchronize()
        ms=s.elapsed_time(e)/10
        fl=2.0*B*T*(F//2)*C*32*231
        print((B,T,F,C),'bwd_data',name,'%.3f ms'%ms, '%.0f TFLOP/s'%(fl/ms/1e9))
PY
python -m pytest tests/test_gpu_model.py -x -q -k "logits_loss_and_gradients or benchmark_shape" 2>&1 | tail -4
python bench.py --workload c3 --steps 20 --warmup 5 --no-cpu-baseline --no-other-workloads --no-parity-probe > gpurun_out/r04_conv16.json 2>/dev/null; python tools/show_bench.py gpurun_out/r04_conv16.json
CTCASR_CONV_F16=0 python bench.py --workload c3 --steps 20 --warmup 5 --no-cpu-baseline --no-other-workloads --no-parity-probe > gpurun_out/r04_conv32.json 2>/dev/null; python tools/show_bench.py gpurun_out/r04_conv32.json
