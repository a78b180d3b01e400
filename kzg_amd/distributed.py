"""Multi-GPU commit: host-side helpers around the C ABI's device group (kzg_mctx, kzg_amd/csrc/mgpu.hip).

The product path is entirely inside libkzg_mi355x.so: the SRS is sharded contiguously (kzg_shard_range), every rank
reduces its slice of each polynomial to ONE 144-byte Jacobian partial on its GPU, one ncclAllGather (RCCL, loaded by
the library) moves world x batch x 144 bytes over xGMI, and every rank adds the `world` partials of each polynomial
locally (EC addition is not an RCCL reduction op, so "all-reduce" = all-gather + local sum).  What lives here:

  * shard_range            -- the library's partition rule (host-only C helper, usable without a GPU);
  * group_from_torch       -- one process per GPU under torch.distributed: rank 0 draws the RCCL unique id, the process
                              group carries the 128 bytes to the other ranks, every rank joins with kzg_mctx_create_rank;
(The CPU model of the exchange used by the world-size-2 gloo test lives with the tests: tests/protocol_model.py.)
"""
import ctypes

from . import _lib as L


def shard_range(n, rank, world):
    """Contiguous shard [lo, hi) of n terms for `rank` (kzg_shard_range: the first n % world ranks get one extra)."""
    lib = L.load()
    lo, hi = ctypes.c_size_t(), ctypes.c_size_t()
    rc = lib.kzg_shard_range(n, rank, world, ctypes.byref(lo), ctypes.byref(hi))
    if rc:
        raise ValueError(f"kzg_shard_range({n}, {rank}, {world}) -> {rc}")
    return lo.value, hi.value


def broadcast_unique_id(dist, rank, make_id):
    """Rank 0 calls make_id() (128 bytes); every rank returns those bytes (carried by the torch process group)."""
    box = [make_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    return box[0]


def group_from_torch(dist, device, rank, world):
    """DeviceGroup for this process's GPU inside an initialised torch.distributed job (one process per GPU)."""
    from .api import DeviceGroup
    uid = broadcast_unique_id(dist, rank, DeviceGroup.unique_id) if world > 1 else DeviceGroup.unique_id()
    return DeviceGroup.for_rank(device, rank, world, uid)
