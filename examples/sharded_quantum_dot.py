"""The same physics as quantum_dot_demo.py with the two-body tensor SHARDED over the GPUs of a node: one process per
GPU, the sharded array module behind the reference's `np=` seam.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 examples/sharded_quantum_dot.py [shells]

(on a one-GPU box: QS_EXAMPLE_ONE_DEVICE=1 puts every rank on cuda:0 and carries the collectives over gloo.)
Every rank generates only ITS rows of the Coulomb tensor, the basis change exchanges one all-to-all, the spin
doubling is slab-local, and the Fock matrix / reference energy move l*l numbers / one number over the node -- no rank
ever holds the whole tensor.  Same API calls as with one GPU.
"""

import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch
import torch.distributed as dist

import quantum_systems_amd as qs
from quantum_systems_amd import kernels, sharded
from quantum_systems_amd._lib import check, load
from quantum_systems_amd.two_dim_ho import get_double_well_one_body_elements


def main():
    shells = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    l = shells * (shells + 1) // 2
    one_device = os.environ.get("QS_EXAMPLE_ONE_DEVICE") == "1"
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    dev = 0 if one_device else int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(dev)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29655")
    dist.init_process_group("gloo" if one_device else "nccl", rank=rank, world_size=world)
    mod = qs.ShardedDeviceModule(rank, world, device=f"cuda:{dev}")

    # this rank's rows of the Coulomb tensor straight from the generator kernel (qs_tdho_coulomb_elements takes a
    # row range): the whole tensor never exists anywhere
    p_lo, p_hi = sharded.SlabPartition(l, world).bounds(rank)
    rows = torch.empty((p_hi - p_lo, l, l, l), dtype=torch.float64, device=f"cuda:{dev}")
    if p_hi > p_lo:
        check(load().qs_tdho_coulomb_elements(rows.data_ptr(), l, p_lo, p_hi, torch.cuda.current_stream().cuda_stream),
              "qs_tdho_coulomb_elements")
    basis = qs.BasisSet(l, 2, np=mod)
    shell_of = np.array([s for s in range(1, shells + 1) for _ in range(s)], dtype=np.float64)[:l]   # omega = 1: shell energies
    basis.h = mod.asarray(np.diag(shell_of).astype(np.complex128))
    basis.s = mod.asarray(np.eye(l, dtype=np.complex128))
    basis.u = mod.from_local(rows.to(torch.complex128), l, axis=0)
    system = qs.SpatialOrbitalSystem(2, basis)

    h_dw = get_double_well_one_body_elements(l, 1.0, 1.0, 2.0, dtype=np.complex128, axis=0)
    _, C = np.linalg.eigh(h_dw)
    system.change_basis(mod.asarray(C))                       # streamed transform; u is now sharded over its second index
    gos = system.construct_general_orbital_system()           # the doubling wants leading-index rows: one all-to-all of the
                                                              # (16x smaller) spatial tensor, then slab-local
    e = complex(gos.compute_reference_energy())
    f = torch.as_tensor(gos.construct_fock_matrix(gos.h, gos.u))
    u = gos.u
    if rank == 0:
        held = f"u[{u.lo}:{u.hi}]" if u.axis == 0 else f"u[:, {u.lo}:{u.hi}]"
        print(f"{world} rank(s), {l} orbitals -> {gos.l} spin orbitals; this rank holds {held} "
              f"= {tuple(u.local.shape)} of {tuple(u.shape)} ({u.rows.numel() * 16 / 1e6:.1f} MB)")
        print(f"reference energy {e.real:.8f}, Fock matrix {tuple(f.shape)}, lowest diagonal element "
              f"{float(f.diagonal().real.min()):.6f}, last kernels: {kernels.last_dispatch()}")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
