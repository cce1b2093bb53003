"""Regression guard for the dispatch of the four-index transform (VERDICT r03 "next" 7).

Every basis size of a list (default 4 ... 260, both dtypes) is timed on the AUTOMATIC route and on each forced alternative
(tuning keys); the run FAILS (exit code 1) when
  * the automatic choice is more than 10 % slower than the best alternative at some size, or
  * a size is more than 25 % below BOTH of its neighbours in TFLOP/s (a cliff the dispatch should not have).
Run it after every kernel or dispatch change, in ONE gpurun call (the boxes of the pool differ):
    python tools/dispatch_guard.py > profiles/rNN_dispatch_guard.txt
QS_GUARD_L=a,b,...  sizes;  QS_GUARD_DTYPES=f64,c128;  QS_GUARD_STEP=n (every n-th size above 96).
Timing: HIP events around back-to-back calls through the Python wrapper, best of three (below ~30 orbitals this is the host's
call rate -- the same for every route, so the comparison stands)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

from quantum_systems_amd import kernels as K

ALTERNATIVES = [      # name, tuning keys, sizes it can change anything for
    ("strip_off", {"gemm_strip": 0}, lambda l, cx: l > 32),
    ("strip_forced", {"gemm_strip": 2, "quad4s": 0, "pair4c": 0, "sandwich": 0, "slab_pair": 0, "gemm_stream": 0},
     lambda l, cx: l > 32),
    ("streamed_off", {"quad4s": 0, "pair4c": 0}, lambda l, cx: l <= 96),
    ("streamed_forced", {"quad4s": 2, "pair4c": 2}, lambda l, cx: 5 <= l <= (64 if cx else 96)),
    ("fused_off", {"sandwich": 0, "small4": 0, "quad4s": 0, "pair4c": 0, "slab_pair": 0}, lambda l, cx: l <= 96),
    ("general_only", {"gemm_fast": 0, "gemm_strip": 0, "quad4s": 0, "pair4c": 0, "sandwich": 0, "small4": 0}, lambda l, cx: l % 16 == 0),
]


def timed(u, C, Ct, out, l):
    for _ in range(2):
        K.transform_two_body(u, C, Ct, out=out)
    torch.cuda.synchronize()
    reps = max(2, min(20, int(2e5 / (l / 64) ** 5 / 160 / (4 if u.is_complex() else 1))))
    best = 1e30
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            K.transform_two_body(u, C, Ct, out=out)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best, K.last_dispatch()


def main():
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(5)
    step = int(os.environ.get("QS_GUARD_STEP", "1"))
    if os.environ.get("QS_GUARD_L"):
        sizes = [int(v) for v in os.environ["QS_GUARD_L"].split(",")]
    else:
        sizes = list(range(4, 97)) + list(range(97, 261, step))
    dtypes = [d == "c128" for d in os.environ.get("QS_GUARD_DTYPES", "f64,c128").split(",")]
    violations = []
    print("# l dtype | auto us TFLOP/s | best alternative (us, x auto) | automatic route")
    for cx in dtypes:
        curve = []
        for l in sizes:
            if cx and l > 260:
                continue
            a = torch.randn((l,) * 4, dtype=torch.float64, device=dev, generator=g)
            u = torch.complex(a, torch.randn((l,) * 4, dtype=torch.float64, device=dev, generator=g)) if cx else a
            del a
            c = torch.randn((l, l), dtype=torch.float64, device=dev, generator=g)
            C = torch.complex(c, torch.randn((l, l), dtype=torch.float64, device=dev, generator=g)) if cx else c
            C = torch.linalg.qr(C)[0].contiguous()
            Ct = C.conj().T.contiguous()
            out = torch.empty_like(u)
            K.tuning_reset()
            t_auto, route = timed(u, C, Ct, out, l)
            ref = out.clone() if l <= 64 else None
            best_name, best_t = "-", 1e30
            for name, keys, applies in ALTERNATIVES:
                if not applies(l, cx):
                    continue
                try:
                    for k, v in keys.items():
                        K.tuning_set(k, v)
                    K.transform_two_body(u, C, Ct, out=out)
                    if K.last_dispatch() == route:
                        continue                      # the same kernels: nothing to compare
                    t, _ = timed(u, C, Ct, out, l)
                    if ref is not None and not torch.equal(out, ref):
                        violations.append(f"l={l} {'c128' if cx else 'f64'}: route {name} is not bit-identical to the automatic one")
                finally:
                    K.tuning_reset()
                if t < best_t:
                    best_name, best_t = name, t
            tf = (32 if cx else 8) * l**5 / t_auto / 1e6
            curve.append((l, tf))
            flag = ""
            if best_t < 1e29 and t_auto > 1.10 * best_t:
                flag = "  <-- AUTOMATIC > 10 % SLOWER"
                violations.append(f"l={l} {'c128' if cx else 'f64'}: automatic {t_auto:.1f} us, {best_name} {best_t:.1f} us")
            alt = f"{best_name} {best_t:.1f} {t_auto / best_t:.2f}" if best_t < 1e29 else "-"
            print(f"{l:4d} {'c128' if cx else 'f64 '} | {t_auto:10.1f} {tf:7.2f} | {alt:32s} | {route[:100]}{flag}", flush=True)
            del u, out
            K.workspace.release()
            torch.cuda.empty_cache()
        for i in range(1, len(curve) - 1):
            (l0, a), (l1, b), (l2, c_) = curve[i - 1], curve[i], curve[i + 1]
            if l1 > 32 and l2 - l0 <= 2 * max(step, 1) + 1 and b < 0.75 * a and b < 0.75 * c_:
                violations.append(f"l={l1} {'c128' if cx else 'f64'}: {b:.1f} TFLOP/s, more than 25 % below both neighbours ({a:.1f}, {c_:.1f})")
    print("# violations:", len(violations))
    for v in violations:
        print("#  ", v)
    return 1 if violations else 0


if __name__ == "__main__":
    sys.exit(main())
