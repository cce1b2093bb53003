"""Run the small-basis transform N times (for rocprofv3 --kernel-trace: per-pass kernel durations come from the trace).
    python tools/s4_kernel_time.py [l] [reps]        -- prints the wall time per transform as well
    python tools/s4_kernel_time.py --parse <kernel_trace.csv>   -- mean / median / min duration of the two passes"""
import sys

if len(sys.argv) > 2 and sys.argv[1] == "--parse":
    import csv
    import statistics
    rows = [r for r in csv.DictReader(open(sys.argv[2])) if "sandwich4" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
    d = d[len(d) // 4 * 2:]        # drop the warm-up half (an even number: the passes alternate)
    for name, part in (("(d, c)", d[0::2]), ("(b, a)", d[1::2])):
        print(f"{name}: n={len(part)} mean {statistics.mean(part) / 1e3:.2f} us  median {statistics.median(part) / 1e3:.2f}  "
              f"min {min(part) / 1e3:.2f}")
    gaps = [int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) for a, b in zip(rows[:-1], rows[1:])]
    gaps = gaps[len(gaps) // 2:]
    print(f"gap between consecutive launches: median {statistics.median(gaps) / 1e3:.2f} us")
    sys.exit(0)

import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quantum_systems_amd import kernels as K  # noqa: E402

l = int(sys.argv[1]) if len(sys.argv) > 1 else 55
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
g = torch.Generator(device="cuda").manual_seed(3)
u = torch.rand((l,) * 4, dtype=torch.float64, device="cuda", generator=g) - 0.5
C, _ = torch.linalg.qr(torch.randn(l, l, dtype=torch.float64, device="cuda", generator=g))
C = C.contiguous(); Ct = C.t().contiguous()
out = torch.empty_like(u)
if os.environ.get("QS_V2ENV"):
    K.tuning_set("sandwich_v2", int(os.environ["QS_V2ENV"]))
for _ in range(10):
    K.transform_two_body(u, C, Ct, out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    K.transform_two_body(u, C, Ct, out=out)
e1.record(); torch.cuda.synchronize()
print(f"l={l}: {e0.elapsed_time(e1) / reps * 1e3:.1f} us per transform (wall, {reps} in a row) | {K.last_dispatch()}")
