"""Tile-shape sweep of the GEMM behind the four-index transform (tuning aid).
    python tools/tune_gemm.py [l ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from quantum_systems_amd import kernels as K

dev = torch.device("cuda:0")


def bench(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e-3)
    return min(ts)


def sweep(l, dt, cfgs, key):
    u = torch.rand(l, l, l, l, dtype=torch.float64, device=dev).to(dt)
    C, _ = torch.linalg.qr(torch.randn(l, l, dtype=dt, device=dev))
    Ct = C.conj().T.contiguous()
    out = torch.empty_like(u)
    kf = 4 if dt.is_complex else 1
    for cfg in cfgs:
        K.tuning_set(key, cfg)
        try:
            t = bench(lambda: K.transform_two_body(u, C, Ct, out=out))
            print(f"l={l} {str(dt)[6:]} cfg={cfg}: {t*1e3:8.2f} ms  {kf*8*l**5/t/1e12:6.2f} TFLOP/s", flush=True)
        except Exception as e:  # noqa: BLE001
            print(f"l={l} cfg={cfg}: {e}")
    K.tuning_set(key, 0)
    del u, out
    K.workspace.release()
    torch.cuda.empty_cache()


if __name__ == "__main__":
    pipe = int(os.environ.get("QS_PIPE", "1"))
    K.tuning_set("gemm_pipe", pipe)
    print("gemm_pipe =", pipe)
    ls = [int(x) for x in sys.argv[1:]] or [256, 192, 128, 55]
    for l in ls:
        sweep(l, torch.float64, [0, 1, 5, 8, 9, 10, 11, 12, 13], "gemm_f64_cfg")
    for l in ls:
        if l <= 160:
            sweep(l, torch.complex128, [0, 1, 2, 3, 4, 6, 7, 8, 9], "gemm_c128_cfg")
