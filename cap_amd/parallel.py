"""Multi-GPU sharding of the hot path for jobs that run one process per GPU under torch.distributed.
(A single process can also drive every GPU itself - `lib.init(devices=[0, 1, ...])`: batches are dealt and long SRSs
sharded inside the library, no communicator involved; see cap_amd/csrc/context.hpp and tests/test_gpu_multidev.py.)

* Whole proofs are independent -> replicas, no data-path collective (`shard_proofs`).
* One MSM shards by contiguous point range: rank g owns bases[g*N/G, (g+1)*N/G) resident on its GPU
  and receives the matching scalar slice; every rank runs the full Pippenger locally and the single
  exchange step is an all-gather of one 96-byte Jacobian point per rank (RCCL has no elliptic-curve
  reduction op, so no all-reduce), followed by G-1 group additions (`sharded_msm`).

On GPUs the exchange step lives inside libcapgpu.so (`capgpu_comm_init` + `capgpu_msm_g1_sharded*`,
cap_amd/csrc/comm.hip): RCCL all-gather on the library stream from device memory, sum on the device.
`init_library_comm` bootstraps that communicator from an existing torch.distributed job (rank 0's
ncclUniqueId is broadcast to the others).  `sharded_msm` below is the same algorithm with the engine and
the combine passed in - the form the CPU (gloo) tests drive with the oracle standing in for the HIP engine.
"""
from __future__ import annotations

from typing import Callable, Sequence

import numpy as np


def shard_range(n: int, rank: int, world: int) -> tuple[int, int]:
    """contiguous [lo, hi) of rank; sizes differ by at most one."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_proofs(count: int, rank: int, world: int) -> list[int]:
    """proof i -> rank i mod world (the reference's par_iter over notes, params_builder.rs:194-226)."""
    return [i for i in range(count) if i % world == rank]


def broadcast_bytes(payload: bytes | None, nbytes: int, src: int = 0, device=None) -> bytes:
    """rank `src` passes `payload` (nbytes long), the others None; everyone gets the bytes back."""
    import torch
    import torch.distributed as dist

    if dist.get_rank() == src:
        assert payload is not None and len(payload) == nbytes
        t = torch.frombuffer(bytearray(payload), dtype=torch.uint8).clone()
    else:
        t = torch.zeros(nbytes, dtype=torch.uint8)
    if device is not None:
        t = t.to(device)
    dist.broadcast(t, src=src)
    return bytes(t.cpu().numpy().tobytes())


def init_library_comm(lib, device=None) -> tuple[int, int]:
    """Create libcapgpu's RCCL communicator over the ranks of the running torch.distributed job.
    `lib` = cap_amd.lib (already initialised on this rank's GPU).  Returns (rank, world)."""
    import torch.distributed as dist

    rank, world = dist.get_rank(), dist.get_world_size()
    uid = broadcast_bytes(lib.comm_unique_id() if rank == 0 else None, 128, 0, device)
    lib.comm_init(rank, world, uid)
    return rank, world


def all_gather_points(local_point: np.ndarray, device=None) -> np.ndarray:
    """all-gather one Jacobian point (12 x u64) per rank -> (world, 12).  uint64 travels as int64."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size()
    t = torch.from_numpy(np.ascontiguousarray(local_point, dtype=np.uint64).view(np.int64).copy())
    if device is not None:
        t = t.to(device)
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    return np.stack([o.cpu().numpy().view(np.uint64) for o in out])


def sharded_msm(local_msm: Callable[[], np.ndarray], combine: Callable[[Sequence[np.ndarray]], np.ndarray],
                device=None) -> np.ndarray:
    """local_msm() -> this rank's partial sum (Jacobian, 12 words); combine(points) -> their group sum.
    Every rank returns the full result."""
    parts = all_gather_points(local_msm(), device)
    return combine(list(parts))
