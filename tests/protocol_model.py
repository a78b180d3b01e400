"""TEST SCAFFOLD (not a product path): the exchange of kzg_amd/csrc/mgpu.hip restated over any torch.distributed backend with
the per-rank GPU operations injected, so that the N > 1 protocol -- partition rule, record layout, status agreement -- can be
exercised on CPU with gloo (tests/test_distributed_gloo.py injects the oracle).

Record a rank contributes (mgpu.hip, record_bytes): `batch` partial points followed by ONE status slot of the same size whose
first 4 bytes are the rank's local kzg_status (little-endian int32).  Every rank always enters the all-gather, with its failure
code if its local phase failed; afterwards every rank reads every status and all return the first failing rank's code.  The
gathered buffer is [world][batch + 1][point_bytes]; polynomial b's partial of rank w sits at slot w * (batch + 1) + b."""
import struct


class RankFailed(Exception):
    def __init__(self, rank, status):
        super().__init__(f"rank {rank} failed in its local phase (status {status})")
        self.rank, self.status = rank, status


class GroupDead(Exception):
    """the exchange did not complete within the deadline (a peer is dead or stalled): mgpu.hip aborts the communicators and every
    later call on the group fails at once (option gather_timeout_ms)"""


class ProtocolModel:
    """local_msm(scalar_shards, batch) -> bytes[batch * point_bytes] (this rank's partial points; may raise);
    local_sum(slots, world, batch, stride) -> list of `batch` results, slots = the gathered bytes, partial of rank w for
    polynomial b at slot w * stride + b."""

    def __init__(self, dist, rank, world, local_msm, local_sum, point_bytes=96, gather_timeout_s=None, grow=None):
        self.dist, self.rank, self.world = dist, rank, world
        self.local_msm, self.local_sum, self.pb = local_msm, local_sum, point_bytes
        self.gather_timeout_s, self.dead = gather_timeout_s, False
        # exchange buffers (mctx_buffers): `cap` is this rank's own capacity, `agreed` the batch every rank is KNOWN to hold buffers
        # for.  grow(batch) stands for the rank's allocations and may raise; `collectives` logs what this rank entered.
        self.cap = self.agreed = 64
        self.grow = grow or (lambda batch: None)
        self.collectives = []

    def _agree(self, code):
        """mctx_agree: a status-only all-gather over buffers that exist since the group was formed; every rank returns the first
        failing rank's code"""
        import torch
        mine = torch.tensor([code, self.rank], dtype=torch.int32)
        allv = torch.empty(2 * self.world, dtype=torch.int32)
        self.collectives.append(("agree", 8))
        self.dist.all_gather_into_tensor(allv, mine)
        for w in range(self.world):
            if int(allv[2 * w]) != 0:
                raise RankFailed(w, int(allv[2 * w]))

    def _buffers(self, batch):
        """The decision to agree follows what the ranks last AGREED on -- the same on every rank -- not a rank's own capacity: after a
        growth that failed on one rank only, the others hold larger buffers than that rank, and deciding by capacity would send the
        failed rank into the agreement while the others go straight to the data exchange (the bug tests/test_gpu_mgpu_world.py found)."""
        if batch <= self.agreed:
            return
        code = 0
        if self.cap < batch:
            try:
                self.grow(batch)
                self.cap = batch
            except Exception as e:  # noqa: BLE001
                code = getattr(e, "status", -3)
        self._agree(code)
        self.agreed = batch

    def commit_batch(self, scalar_shards, batch):
        import datetime
        import torch
        if self.dead:
            raise GroupDead("this device group is dead")
        self._buffers(batch)
        status = 0
        try:
            mine = bytes(self.local_msm(scalar_shards, batch))
            assert len(mine) == batch * self.pb
        except Exception as e:  # noqa: BLE001  the rank still enters the exchange, with its failure code
            status = getattr(e, "status", -4)
            mine = bytes(batch * self.pb)
        rec = mine + struct.pack("<ii", status, self.rank) + bytes(self.pb - 8)
        gathered = torch.empty(self.world * len(rec), dtype=torch.uint8)
        self.collectives.append(("gather", len(rec)))
        if self.gather_timeout_s is None:
            self.dist.all_gather_into_tensor(gathered, torch.frombuffer(bytearray(rec), dtype=torch.uint8))
        else:   # mctx_wait: poll with a deadline; on expiry abort and retire the group
            work = self.dist.all_gather_into_tensor(gathered, torch.frombuffer(bytearray(rec), dtype=torch.uint8), async_op=True)
            try:
                done = work.wait(timeout=datetime.timedelta(seconds=self.gather_timeout_s))
            except RuntimeError:
                done = False
            if done is False:
                self.dead = True
                raise GroupDead("the all-gather did not complete within %.1f s" % self.gather_timeout_s)
        raw = gathered.numpy().tobytes()
        stride = batch + 1
        for w in range(self.world):
            st = struct.unpack_from("<i", raw, (w * stride + batch) * self.pb)[0]
            if st != 0:
                raise RankFailed(w, st)
        return self.local_sum(raw, self.world, batch, stride)
