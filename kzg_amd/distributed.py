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
import os

from . import _lib as L

# What a ONE-NODE host exports before the first RCCL call of the process (values the host already exported are kept).  The
# device group never leaves the node (8 GPUs over xGMI), but RCCL does not know that: it bootstraps every communicator -- a
# world-1 one included -- over TCP on the first non-loopback interface it finds, starts its RAS thread on that interface, and
# probes InfiniBand and network plugins.  On a box whose interface swallows packets that cost round 4's driver run five minutes
# per communicator; on loopback the same formation takes well under a second (profiles/r05_rccl_formation_ab.txt).
SINGLE_NODE_RCCL_ENV = {
    "NCCL_SOCKET_IFNAME": "lo",     # bootstrap + socket transport on loopback
    "NCCL_RAS_ENABLE": "0",         # no RAS listener threads / sockets (2.24+)
    "NCCL_IB_DISABLE": "1",         # no InfiniBand probing: nothing crosses the node
    "NCCL_NET_PLUGIN": "none",      # no external network plugin search
}


def single_node_rccl_env():
    """setdefault()s SINGLE_NODE_RCCL_ENV into this process' environment (KZG_RCCL_SINGLE_NODE_ENV=0 leaves it alone).  Called by
    DeviceGroup before anything loads RCCL; a Rust host does the same with std::env::set_var.  Returns
    the variables it set."""
    if os.environ.get("KZG_RCCL_SINGLE_NODE_ENV", "1") == "0":
        return {}
    done = {}
    for k, v in SINGLE_NODE_RCCL_ENV.items():
        if k not in os.environ:
            os.environ[k] = v       # os.environ assignment calls putenv: the C library sees it
            done[k] = v
    return done


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
