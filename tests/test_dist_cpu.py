"""world_size-2 gloo test of the N>1 path: an MSM sharded by point range, one all-gather of a
96-byte partial per rank, G-1 group additions - with the CPU oracle standing in for the HIP engine.
(`-m "not gpu"`)"""
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from cap_amd import parallel as par
    from oracle import capref as cr

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    bases = cr.g1_fixed_base_batch(cr.random_field(11, 1, n, False))
    scalars = cr.random_field(12, 1, n, False)
    lo, hi = par.shard_range(n, rank, world)

    def local():
        return cr.msm_g1(bases[lo:hi], scalars[lo:hi])

    def combine(points):
        acc = points[0]
        for p in points[1:]:
            acc = cr.g1_add(acc, p)
        return acc

    # the bootstrap of the library's RCCL communicator: rank 0's unique id reaches every rank unchanged
    class FakeLib:
        seen = None

        def comm_unique_id(self):
            return bytes((7 * i + 3) % 256 for i in range(128))

        def comm_init(self, r, w, uid):
            FakeLib.seen = (r, w, uid)

    assert par.init_library_comm(FakeLib()) == (rank, world)
    assert FakeLib.seen == (rank, world, bytes((7 * i + 3) % 256 for i in range(128)))
    total = par.sharded_msm(local, combine)
    full = cr.msm_g1(bases, scalars)
    ok = np.array_equal(cr.g1_to_affine(total), cr.g1_to_affine(full))
    mine = par.shard_proofs(5, rank, world)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, bool(ok), mine))


def test_sharded_msm_world2_gloo():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world = 2
    procs = [ctx.Process(target=_worker, args=(r, world, port, 777, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [True, True]
    assert sorted(res[0][2] + res[1][2]) == [0, 1, 2, 3, 4]
