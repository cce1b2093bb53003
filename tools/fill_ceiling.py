"""Write-only ceiling: torch fill of a 40 GB complex128 buffer (what the spin expansion could reach at best)."""
import torch, time
dev = torch.device("cuda:0")
buf = torch.empty(int(40e9 // 16), dtype=torch.complex128, device=dev)
for name, fn in (("zero_", lambda: buf.zero_()), ("fill_(1+2j)", lambda: buf.fill_(1 + 2j))):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): fn()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) * 1e-3 / 5
    print(f"{name}: {buf.numel()*16/t/1e12:.2f} TB/s written")
