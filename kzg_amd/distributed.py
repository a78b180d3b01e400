"""Multi-GPU MSM: the SRS is sharded contiguously, one process per GPU (torch.distributed; backend
"nccl" is RCCL on ROCm), each rank reduces its shard to ONE partial commitment on its own GPU and the
96-byte partials are exchanged with a single all_gather over xGMI; every rank then adds the N partial
points locally (EC addition is not an RCCL reduction op, so "all-reduce" = all-gather + local sum).

The collective moves N x 96 bytes -- latency-bound, independent of the polynomial size -- which is why
buckets are reduced locally first (exchanging raw buckets would move tens of MiB per commitment).

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
    def __init__(self, dist, rank, world, local_msm, local_sum, device="cpu"):
        """local_msm(scalar_shard) -> torch.uint8[96] partial point (affine Montgomery) on `device`;
        local_sum(torch.uint8[world*96]) -> 96-byte result."""
        self.dist, self.rank, self.world = dist, rank, world
        self.local_msm, self.local_sum, self.device = local_msm, local_sum, device

    def commit(self, scalar_shard):
        import torch
        mine = self.local_msm(scalar_shard)
        if self.world == 1:
            return self.local_sum(mine)
        gathered = torch.empty(self.world * 96, dtype=torch.uint8, device=self.device)
        self.dist.all_gather_into_tensor(gathered, mine)
        return self.local_sum(gathered)

    @staticmethod
    def for_engine(engine, srs_shard, dist, rank, world):
        """Product wiring: partial MSM and final sum both run in libkzg_mi355x.so on this rank's GPU."""
        import torch
        dev = torch.device("cuda", engine.device)
        part = torch.empty(96, dtype=torch.uint8, device=dev)
        out = ctypes.create_string_buffer(96)

        def local_msm(shard):  # shard: kzg_amd.DeviceBuffer resident on this GPU
            rc = engine.lib.kzg_msm_g1(engine.ctx, srs_shard.handle, 0, shard.ptr, shard.n, shard.sfmt,
                                       L.IN_DEVICE | L.OUT_DEVICE, ctypes.c_void_p(part.data_ptr()), L.G1_AFFINE_MONT)
            if rc:
                raise RuntimeError(engine.last_error())
            return part

        def local_sum(gathered):
            torch.cuda.current_stream(dev).synchronize()
            rc = engine.lib.kzg_g1_sum(engine.ctx, ctypes.c_void_p(gathered.data_ptr()), gathered.numel() // 96,
                                       L.G1_AFFINE_MONT, L.IN_DEVICE, out, L.G1_AFFINE_MONT)
            if rc:
                raise RuntimeError(engine.last_error())
            return out.raw

        return ShardedCommitter(dist, rank, world, local_msm, local_sum, device=dev)
