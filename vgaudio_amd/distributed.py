"""Multi-GPU plumbing: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI
on the GPU box, "gloo" in the CPU tests).

Channels (ADPCM/ADX) and streams (HCA) are independent units (GcAdpcmFormat.cs:65-68,
CriAdxFormat.cs:67-81, one CriHcaEncoder per file CriHcaFormat.cs:44), so the data path has NO
collective: rank r owns a contiguous block of channels.  The only exchange is the per-channel
metadata the caller needs in one place (16 coefficients = 32 B per channel), gathered to every
rank with one all_gather, and the max-over-ranks of the step time for reporting.
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
