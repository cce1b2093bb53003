"""Experiment: the replicated-u layout with the HBM-bound contraction over a (m = l/G rows against all of u) of
chunk k+1 issued on a second stream while the MFMA-bound d, c contractions of chunk k run.  One GPU, rank 0's share."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from quantum_systems_amd import kernels as K, sharded
l = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda:0")
u = torch.rand(l, l, l, l, dtype=torch.float64, device=dev)
C, _ = torch.linalg.qr(torch.randn(l, l, dtype=torch.float64, device=dev)); C = C.contiguous()
Ct = C.T.contiguous(); CT = C.T.contiguous()
def t(fn, reps=5):
    fn(); torch.cuda.synchronize(); ts=[]
    for _ in range(reps):
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1)*1e-3)
    return min(ts)
side = torch.cuda.Stream()
for world in (2, 4, 8):
    lo, hi = sharded.SlabPartition(l, world).bounds(0)
    pc = hi - lo
    rows = Ct[lo:hi].contiguous()
    base = t(lambda: sharded.transform_two_body_replicated(u, C, Ct, 0, world))
    ref = sharded.transform_two_body_replicated(u, C, Ct, 0, world)
    for nch in (2, 4, 8):
        bc = l // nch
        w = torch.empty((pc, l, l, l), dtype=torch.float64, device=dev)       # W[p, b, c, d]
        t1 = torch.empty((pc, l, l, l), dtype=torch.float64, device=dev)
        t2 = torch.empty((pc, l, l, l), dtype=torch.float64, device=dev)
        out = torch.empty((pc, l, l, l), dtype=torch.float64, device=dev)
        L2, L3 = l * l, l ** 3
        evs = [torch.cuda.Event() for _ in range(nch)]
        def run():
            main = torch.cuda.current_stream()
            side.wait_stream(main)
            with torch.cuda.stream(side):
                for k in range(nch):
                    # a:  W[p, b in chunk, (c,d)] = rows[p, a] u[a, b in chunk, (c,d)]   (strided rows of u)
                    K.gemm_raw(torch.float64, rows, u, w, pc, bc * L2, l, l, L3, L3, a_off=0, b_off=k * bc * L2, c_off=k * bc * L2)
                    evs[k].record(side)
            for k in range(nch):
                main.wait_event(evs[k])
                # d on the chunk: T1[p][(b in chunk, c), s] = W[p][(b in chunk, c), d] C[d, s], batch over p
                K.gemm_raw(torch.float64, w, C, t1, bc * l, l, l, l, l, l, batch=pc, sa=L3, sb=0, sc=L3,
                           a_off=k * bc * L2, c_off=k * bc * L2)
            # c over the whole slab, then b (nothing left on the side stream)
            K.gemm_raw(torch.float64, CT, t1, t2, l, l, l, l, l, l, batch=pc * l, sa=0, sb=L2, sc=L2)
            K.gemm_raw(torch.float64, Ct, t2, out, l, L2, l, l, L2, L2, batch=pc, sa=0, sb=L3, sc=L3)
            return out
        got = run(); torch.cuda.synchronize()
        ok = torch.equal(got, ref)
        tt = t(run)
        print(f"world={world}: plain {base*1e3:7.2f} ms   overlapped a-contraction, {nch} chunks: {tt*1e3:7.2f} ms   ({'bit-equal' if ok else 'DIFFERS'})", flush=True)
        del w, t1, t2, out
