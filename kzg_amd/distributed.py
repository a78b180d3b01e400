"""Multi-GPU MSM: the SRS is sharded contiguously, one process per GPU (torch.distributed; backend
"nccl" is RCCL on ROCm).  Each rank reduces its shard to ONE partial commitment per polynomial on its
own GPU; the partials of a batch of B polynomials (144-byte Jacobian points: no inversion per partial) are
exchanged with a single all_gather over xGMI (world x B x 144 bytes); every rank then adds, per polynomial, the `world` partial points locally
(EC addition is not an RCCL reduction op, so "all-reduce" = all-gather + local sum).

The collective is latency-bound and independent of the polynomial size -- which is why buckets are
reduced locally first (exchanging raw buckets would move ~6 MiB per rank per commitment).

`ShardedCommitter` takes the two local operations as callables so the sharding / collective logic can
be exercised on CPU with the gloo backend (tests/test_distributed_gloo.py injects the oracle there);
`ShardedCommitter.for_engine` wires in the HIP engine, which is the only product configuration.
"""
import ctypes

from . import _lib as L


def shard_range(n, rank, world):
    """Contiguous shard [lo, hi) of n terms for `rank` (first n % world ranks get one extra)."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


class ShardedCommitter:
    def __init__(self, dist, rank, world, local_msm, local_sum, device="cpu", always_gather=False, point_bytes=96):
        """local_msm(scalar_shards, batch) -> torch.uint8[batch*point_bytes]: this rank's partial points on
        `device`;  local_sum(torch.uint8[batch*world*point_bytes] laid out [batch][world][point_bytes], batch)
        -> list of `batch` 96-byte affine results."""
        self.dist, self.rank, self.world = dist, rank, world
        self.local_msm, self.local_sum, self.device = local_msm, local_sum, device
        self.pb = point_bytes
        self.always_gather = always_gather  # run the collective even at world size 1 (testing the RCCL path)

    def commit_batch(self, scalar_shards, batch):
        import torch
        mine = self.local_msm(scalar_shards, batch)                      # [batch][96]
        if self.world == 1 and not self.always_gather:
            return self.local_sum(mine, batch)
        gathered = torch.empty(self.world * batch * self.pb, dtype=torch.uint8, device=self.device)
        self.dist.all_gather_into_tensor(gathered, mine)                 # [world][batch][pb]
        grouped = gathered.view(self.world, batch, self.pb).transpose(0, 1).contiguous().view(-1)  # [batch][world][pb]
        return self.local_sum(grouped, batch)

    def commit(self, scalar_shard):
        return self.commit_batch(scalar_shard, 1)[0]

    @staticmethod
    def for_engine(engine, srs_shard, dist, rank, world, max_batch=16, always_gather=False):
        """Product wiring: partial MSMs and final sums both run in libkzg_mi355x.so on this rank's GPU.
        Shards are kzg_amd.DeviceBuffer objects holding batch * n_shard scalars resident on this GPU."""
        import torch
        dev = torch.device("cuda", engine.device)
        # partials travel as 144-byte Jacobian points: no field inversion per partial, one per final result
        PB = 144
        part = torch.empty(max_batch * PB, dtype=torch.uint8, device=dev)
        out = ctypes.create_string_buffer(96 * max_batch)

        def local_msm(shard, batch):
            n = shard.n // batch
            rc = engine.lib.kzg_msm_g1_batch(engine.ctx, srs_shard.handle, 0, shard.ptr, n, batch, shard.sfmt,
                                             L.IN_DEVICE | L.OUT_DEVICE, ctypes.c_void_p(part.data_ptr()), L.G1_JACOBIAN_MONT)
            if rc:
                raise RuntimeError(engine.last_error())
            return part[: batch * PB]

        def local_sum(grouped, batch):
            torch.cuda.current_stream(dev).synchronize()
            count = grouped.numel() // (PB * batch)
            rc = engine.lib.kzg_g1_sum_batch(engine.ctx, ctypes.c_void_p(grouped.data_ptr()), count, batch,
                                             L.G1_JACOBIAN_MONT, L.IN_DEVICE, out, L.G1_AFFINE_MONT)
            if rc:
                raise RuntimeError(engine.last_error())
            return [out.raw[96 * b: 96 * (b + 1)] for b in range(batch)]

        return ShardedCommitter(dist, rank, world, local_msm, local_sum, device=dev, always_gather=always_gather,
                                point_bytes=PB)
