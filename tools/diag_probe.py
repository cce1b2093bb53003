import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from quantum_systems_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
def t(fn, reps=3):
    fn(); torch.cuda.synchronize(); ts=[]
    for _ in range(reps):
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return min(ts)
for blocks in (256, 512):
    for iters, per in ((8000, 8), (4001, 16)):
        sink = torch.zeros(1 + 2*blocks, dtype=torch.int64, device=dev)
        ms = t(lambda: lib.qs_probe_mfma_f64(sink.data_ptr(), blocks, iters, st))
        print(f"probe{per} blocks={blocks}: {blocks*4*iters*per*2048/ms/1e9:.1f} TFLOP/s")
