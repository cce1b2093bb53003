"""Small-basis transform: the two fused 4-wide passes (qs_sandwich4.hip) against the previous path
(fused (d, c) on the 16-wide instruction + two streaming products), same process, alternating runs:
bit-equality of the results and time per transform.   python tools/sandwich_check.py [l ...]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from quantum_systems_amd import kernels as K  # noqa: E402


def timed(fn, reps=50):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3   # us


ls = [int(a) for a in sys.argv[1:]] or list(range(21, 65))
g = torch.Generator(device="cuda").manual_seed(3)
for l in ls:
    u = torch.rand((l,) * 4, dtype=torch.float64, device="cuda", generator=g) - 0.5
    C, _ = torch.linalg.qr(torch.randn(l, l, dtype=torch.float64, device="cuda", generator=g))
    C = C.contiguous(); Ct = C.t().contiguous()
    out = {}
    # 0 the 16-wide path, 1 automatic; forced wherever the kernel exists: 4 both passes, 5 (d, c) only, 6 (b, a) only
    MODES = (0, 1, 4, 5, 6) if len(ls) > 8 else (0, 1, 2, 3, 4, 10, 11, 13)
    if os.environ.get("QS_MODES"):
        MODES = tuple(int(x) for x in os.environ["QS_MODES"].split(","))
    for mode in MODES:
        if mode >= 10:
            K.tuning_set("sandwich", 1); K.tuning_set("sandwich_mode", mode - 10)
        else:
            K.tuning_reset(); K.tuning_set("sandwich", mode)
        if os.environ.get("QS_V2"):
            K.tuning_set("sandwich_v2", int(os.environ["QS_V2"]))
        if os.environ.get("QS_T2"):
            K.tuning_set("sandwich_t2", int(os.environ["QS_T2"]))
        res = torch.empty_like(u)
        K.transform_two_body(u, C, Ct, out=res)
        disp = K.last_dispatch()
        t = min(timed(lambda: K.transform_two_body(u, C, Ct, out=res)) for _ in range(3))
        out[mode] = (res.clone(), t, disp)
    K.tuning_reset()
    ref = out[0][0]
    flops = 8 * l**5
    line = f"l={l:3d}"
    for mode in MODES:
        same = torch.equal(out[mode][0], ref)
        line += f" | mode {mode}: {out[mode][1]:7.1f} us {flops / out[mode][1] / 1e6:5.1f} TF {'bit-equal' if same else 'DIFFERS ' + format((out[mode][0] - ref).abs().max().item(), '.1e')}"
    print(line, flush=True)
    best = min(MODES, key=lambda m: out[m][1])
    print("       best mode", best, "|", out[1][2], flush=True)
