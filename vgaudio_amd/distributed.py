"""Multi-GPU plumbing: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI
on the GPU box, "gloo" in the CPU tests).

Channels (ADPCM/ADX) and streams (HCA) are independent units (GcAdpcmFormat.cs:65-68,
CriAdxFormat.cs:67-81, one CriHcaEncoder per file CriHcaFormat.cs:44), so the data path has NO
collective: rank r owns a contiguous block of channels.  The only exchange is the FINAL GATHER
(SURVEY.md 8e): the reference leaves every channel's bitstream and coefficients in one place
(GcAdpcmFormat.cs:65-74), so rank 0 receives every rank's ADPCM rows and coefficient rows
(BitstreamGather: grouped point-to-point send/recv in channel chunks -- seven peers arrive on
seven distinct xGMI links, there is no ring) plus the max-over-ranks of the step time.
"""
import os

import torch
import torch.distributed as dist


def env_world():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def shard_channels(total_channels, world, rank):
    """Contiguous block [first, first+count) of channels for `rank`; blocks differ by at most one."""
    base, extra = divmod(total_channels, world)
    count = base + (1 if rank < extra else 0)
    first = rank * base + min(rank, extra)
    return first, count


def init(backend, device=None):
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    if dist.is_initialized():
        return
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=device)
    else:
        dist.init_process_group(backend)


def gather_channel_metadata(local, counts):
    """all_gather of per-channel rows ([count_r, k] per rank, counts may differ by one) ->
    [sum(counts), k] on every rank, in channel order."""
    world = dist.get_world_size()
    if world == 1:
        return local
    k = local.shape[1]
    pad = max(counts)
    padded = torch.zeros((pad, k), dtype=local.dtype, device=local.device)
    padded[:local.shape[0]] = local
    # RCCL/gloo have no int16 datatype: move the rows as bytes
    raw = padded.view(torch.uint8)
    parts = [torch.empty_like(raw) for _ in range(world)]
    dist.all_gather(parts, raw)
    return torch.cat([parts[r].view(local.dtype)[:counts[r]] for r in range(world)], dim=0)


def max_over_ranks(value, device):
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


class BitstreamGather:
    """Final gather of a sharded encode: every rank's output rows ([count_r, pitch] uint8 bitstream rows and
    [count_r, 16] int16 coefficient rows) to rank 0, in channel order.

    Grouped point-to-point transfers (batch_isend_irecv = one ncclGroup per chunk on RCCL): rank 0 posts one
    receive per peer and chunk, every peer one send per chunk, so the seven peers of an 8-GPU node stream over
    seven distinct xGMI links at once.  `chunk_channels` bounds a transfer (a few hundred MB), so that a caller
    encoding in channel chunks can hand each finished chunk over while the next one computes; with
    async_op=True the transfer runs on the process group's own stream next to the following step's kernels
    (the caller keeps the source tensors untouched until wait()).  RCCL/gloo have no int16 type: rows move as bytes.
    """

    def __init__(self, counts, pitch, device, chunk_channels=512):
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.counts = [int(counts)] * self.world if isinstance(counts, int) else [int(c) for c in counts]
        assert len(self.counts) == self.world
        self.firsts = [sum(self.counts[:r]) for r in range(self.world)]
        self.pitch = int(pitch)
        self.chunk = max(1, int(chunk_channels))
        self.all_adpcm = self.all_coefs = None
        if self.rank == 0:
            total = sum(self.counts)
            self.all_adpcm = torch.empty((total, self.pitch), dtype=torch.uint8, device=device)
            self.all_coefs = torch.empty((total, 16), dtype=torch.int16, device=device)

    def gather(self, adpcm, coefs, async_op=False):
        """adpcm [count, pitch] uint8, coefs [count, 16] int16 (this rank's rows).  Returns a list of work handles
        (empty when the transfer has completed)."""
        n = self.counts[self.rank]
        assert adpcm.shape[0] >= n and adpcm.shape[1] == self.pitch and coefs.shape[0] >= n
        if self.rank == 0:
            self.all_adpcm[:n].copy_(adpcm[:n], non_blocking=True)
            self.all_coefs[:n].copy_(coefs[:n].reshape(n, 16), non_blocking=True)
        if self.world == 1:
            return []
        coef_bytes = coefs[:n].reshape(n, 16).contiguous().view(torch.uint8)
        works = []
        longest = max(self.counts)
        for c0 in range(0, longest, self.chunk):
            ops = []
            if self.rank == 0:
                for r in range(1, self.world):
                    c1 = min(c0 + self.chunk, self.counts[r])
                    if c1 > c0:
                        f = self.firsts[r]
                        ops.append(dist.P2POp(dist.irecv, self.all_adpcm[f + c0:f + c1], r))
                        ops.append(dist.P2POp(dist.irecv, self.all_coefs[f + c0:f + c1].view(torch.uint8), r))
            else:
                c1 = min(c0 + self.chunk, n)
                if c1 > c0:
                    ops.append(dist.P2POp(dist.isend, adpcm[c0:c1], 0))
                    ops.append(dist.P2POp(dist.isend, coef_bytes[c0:c1], 0))
            if ops:
                works.extend(dist.batch_isend_irecv(ops))
        if async_op:
            return works
        for w in works:
            w.wait()
        return []

    def verify(self, adpcm, coefs, nbytes):
        """rank 0: its own rows arrived unchanged and every peer's rows are populated (frame headers name a
        predictor 0..7 and a scale 0..12, GcAdpcmEncoder.cs:83,118-170).  Returns a short description."""
        if self.rank != 0:
            return None
        n = self.counts[0]
        ok = bool(torch.equal(self.all_adpcm[:n, :nbytes], adpcm[:n, :nbytes]) and
                  torch.equal(self.all_coefs[:n], coefs[:n].reshape(n, 16)))
        full = nbytes - nbytes % 8
        heads = self.all_adpcm[:, :full].reshape(self.all_adpcm.shape[0], -1, 8)[:, :, 0]
        ok = ok and int((heads >> 4).max()) <= 7 and int((heads & 15).max()) <= 12
        return "own rows identical, all %d channels' frame headers valid" % self.all_adpcm.shape[0] if ok else "MISMATCH"
