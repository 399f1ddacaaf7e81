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


RENDEZVOUS_PORT_TAKEN = 98      # exit code of a rank 0 whose rendezvous port was taken (EADDRINUSE): launch_local_ranks tries again


def init(backend, device=None, timeout_s=None):
    """timeout_s: how long a collective may wait for a rank that never arrives (default: torch's ten minutes for RCCL)"""
    import datetime
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    if dist.is_initialized():
        return
    kw = {} if timeout_s is None else {"timeout": datetime.timedelta(seconds=timeout_s)}
    try:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device, **kw)
        else:
            dist.init_process_group(backend, **kw)
    except Exception as e:  # noqa: BLE001 -- only to give ONE failure an exit code of its own, everything else re-raises
        # rank 0 hosts the rendezvous store: a port somebody else took between launch_local_ranks' look-up and this bind is
        # the one failure a relaunch on another port cures -- the launcher retries on this exit code and on nothing else
        text = str(e).lower()
        if env_world()[0] == 0 and ("address already in use" in text or "eaddrinuse" in text):
            import sys
            print("rendezvous port %s is taken: %s" % (os.environ.get("MASTER_PORT"), e), file=sys.stderr, flush=True)
            sys.stderr.flush()
            os._exit(RENDEZVOUS_PORT_TAKEN)
        raise


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


def launch_local_ranks(argv, nproc, master_port=None, extra_env=None, timeout=None):
    """One process per GPU without torchrun: runs `python argv...` nproc times with RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR / MASTER_PORT set (the environment torch.distributed.run would give them) and waits for all of them.
    Rank 0 inherits stdout (it prints the result line); the other ranks' stdout goes to stderr.

    Exit code: 0 only if every rank ended with 0.  Otherwise the first non-zero code seen (a rank that fails takes the
    others down: they would wait in a collective for ever), 124 if `timeout` ran out, 125 if ranks were still running
    `grace` seconds after rank 0 had finished cleanly (they are killed; rank 0's line is out by then).

    The rendezvous port: when the caller names none, a free loop-back port is looked up and released again before the
    ranks bind it -- another process can take it in between; rank 0 then leaves with RENDEZVOUS_PORT_TAKEN (init() above)
    and the launch is tried again on another port (twice).  Any other failure -- an import error, a failed assertion, a
    digest that differs -- is final at the first attempt.  HSA_ENABLE_IPC_MODE_LEGACY=0 is set for the ranks when the caller's environment
    does not set it: this image's driver only supports dmabuf IPC, and RCCL fails with hipIpcGetMemHandle errors without
    it (the variable is read by the ROCm runtime of the rank processes only; the launcher's own process is not touched)."""
    import socket
    import subprocess
    import sys
    import time

    def free_port():
        with socket.socket() as sock:                      # a free port on the loop-back interface
            sock.bind(("127.0.0.1", 0))
            return sock.getsockname()[1]

    def run_once(port):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(nproc),
                   LOCAL_WORLD_SIZE=str(nproc), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        env.update(extra_env or {})
        procs = []
        for r in range(nproc):
            procs.append(subprocess.Popen([sys.executable] + list(argv), env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                                          stdout=None if r == 0 else sys.stderr))
        started = time.monotonic()
        deadline = None if timeout is None else started + timeout
        grace = 20.0
        rank0_done_at = None
        rc = 0
        early_rank0_failure = False
        live = list(procs)
        while live:
            for p in list(live):
                code = p.poll()
                if code is None:
                    continue
                live.remove(p)
                if p is procs[0] and code == 0:
                    rank0_done_at = time.monotonic()
                if code != 0 and rc == 0:
                    rc = code
                    early_rank0_failure = p is procs[0] and code == RENDEZVOUS_PORT_TAKEN
            straggling = rank0_done_at is not None and time.monotonic() > rank0_done_at + grace
            if rc != 0 or straggling or (deadline is not None and time.monotonic() > deadline):
                for p in live:
                    p.kill()
                for p in live:
                    p.wait()
                return (rc if rc else (125 if straggling else 124)), early_rank0_failure
            time.sleep(0.05)
        return rc, early_rank0_failure

    if master_port is not None:
        return run_once(master_port)[0]
    rc = 0
    for _attempt in range(3):
        rc, early = run_once(free_port())
        if rc == 0 or not early:
            break
    return rc


def describe_job(device):
    """What the first run on real multi-GPU hardware should be able to prove from its result line alone: the backend and its
    version, the world size the BACKEND reports (not the environment's), and for every rank its host, process, device index,
    device name and PCI bus id (gathered with the job's own collective: the list has one entry per rank only if the
    collective reached all of them; two ranks on one bus id mean the job shares a GPU)."""
    import socket
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    backend = dist.get_backend() if dist.is_initialized() else "none"
    dev = torch.device(device)
    mine = {"rank": rank, "host": socket.gethostname(), "pid": os.getpid(), "device": str(dev)}
    if dev.type == "cuda":
        props = torch.cuda.get_device_properties(dev)
        bus = None
        if all(hasattr(props, k) for k in ("pci_domain_id", "pci_bus_id", "pci_device_id")):
            bus = "%04x:%02x:%02x" % (props.pci_domain_id, props.pci_bus_id, props.pci_device_id)
        mine.update({"name": props.name, "pci_bus_id": bus, "compute_units": props.multi_processor_count,
                     "memory_GiB": round(props.total_memory / 2 ** 30, 1)})
    # 1022 bytes of text + a two-byte length: a long host name next to a long device name still parses on the other side
    SLOT = 1024
    text = repr(mine).encode()[:SLOT - 2]
    buf = torch.zeros(SLOT, dtype=torch.uint8)
    buf[:len(text)] = torch.frombuffer(bytearray(text), dtype=torch.uint8)
    buf[SLOT - 2] = len(text) & 0xFF
    buf[SLOT - 1] = len(text) >> 8
    on = dev if backend == "nccl" else torch.device("cpu")
    parts = [torch.zeros(SLOT, dtype=torch.uint8, device=on) for _ in range(world)]
    if world > 1:
        dist.all_gather(parts, buf.to(on))
    else:
        parts = [buf]
    import ast
    ranks = []
    for t in parts:
        t = t.cpu()
        n = int(t[SLOT - 2]) | (int(t[SLOT - 1]) << 8)
        try:
            ranks.append(ast.literal_eval(bytes(t[:n].tolist()).decode()))
        except (ValueError, SyntaxError):
            ranks.append({"unreadable": True})
    version = None
    if backend == "nccl":
        try:
            version = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:  # noqa: BLE001 -- a missing version string must not cost the result line
            version = "unknown"
    buses = [(r.get("host"), r.get("pci_bus_id")) for r in ranks if r.get("pci_bus_id")]     # the same bus id on two hosts is two GPUs
    return {"backend": backend, "rccl_version": version, "world_size_reported_by_backend": world,
            "world_size_env": int(os.environ.get("WORLD_SIZE", "1")), "ranks": ranks,
            "distinct_devices": len(set((r.get("host"), r.get("pci_bus_id") or r.get("device")) for r in ranks)),
            "shared_devices": len(buses) != len(set(buses))}


_GOLD = 0x9E3779B97F4A7C15
_MIX = 0xC2B2AE3D27D4EB4F


def _i64(v):
    v &= (1 << 64) - 1
    return v - (1 << 64) if v >= (1 << 63) else v


def row_digests(rows, nbytes, coefs, chunk=512):
    """Per-row part of rows_digest: [count] int64 = sum over words of x[j] * b(j) (mod 2^64) of every row (its first
    `nbytes` bytes as little-endian 64-bit words + the tail bytes one by one, then its 16 coefficients).  Independent of
    the row's position in the batch, so a committed list of them names the channel that differs."""
    count = rows.shape[0]
    dev = rows.device
    n8 = nbytes // 8
    tail = nbytes - 8 * n8
    assert rows.dtype == torch.uint8 and rows.stride(1) == 1 and rows.stride(0) % 8 == 0 and rows.data_ptr() % 8 == 0
    jw = torch.arange(n8 + tail + 4, dtype=torch.int64, device=dev)
    b = (jw * _i64(_MIX) + 1) | 1
    out = torch.empty(count, dtype=torch.int64, device=dev)
    for c0 in range(0, count, chunk):
        c1 = min(c0 + chunk, count)
        inner = torch.zeros(c1 - c0, dtype=torch.int64, device=dev)
        if n8:
            inner += (rows[c0:c1, :8 * n8].view(torch.int64) * b[:n8]).sum(dim=1)
        if tail:
            inner += (rows[c0:c1, 8 * n8:nbytes].to(torch.int64) * b[n8:n8 + tail]).sum(dim=1)
        inner += (coefs[c0:c1].reshape(c1 - c0, 16).contiguous().view(torch.int64) * b[n8 + tail:n8 + tail + 4]).sum(dim=1)
        out[c0:c1] = inner
    return out


def combine_row_digests(inner, first_channel):
    """rows_digest from row_digests: sum over rows of a(g) * inner[g] (mod 2^64), g = GLOBAL channel index."""
    g = torch.arange(first_channel, first_channel + inner.shape[0], dtype=torch.int64, device=inner.device)
    a = (g * _i64(_GOLD) + _i64(_MIX)) | 1
    return int((inner * a).sum().item()) & ((1 << 64) - 1)


def rows_digest(rows, nbytes, coefs, first_channel, chunk=512):
    """A 64-bit position-weighted checksum of a block of output rows, computed where the rows live (CPU or GPU):
    sum over rows of a(g) * sum over words of x[g][j] * b(j)  (mod 2^64), g = GLOBAL channel index, the row's first
    `nbytes` bytes read as little-endian 64-bit words (+ the tail bytes one by one) followed by its 16 coefficients.
    A row that arrives in another row's place, truncated, stale or shifted changes the sum.  Returns a Python int."""
    return combine_row_digests(row_digests(rows, nbytes, coefs, chunk), first_channel)


def max_over_ranks(value, device):
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else device)
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
        self.device = device
        # gloo moves host memory only: device rows then travel through host copies (the CPU tests, and trying the N > 1
        # path on a box with fewer GPUs than ranks); RCCL sends straight from HBM to HBM
        self.via_host = self.world > 1 and dist.get_backend() == "gloo" and torch.device(device).type != "cpu"
        self.peer_digests = None               # rank 0: [world] int64, what every rank says its rows sum to (rows_digest)
        if self.rank == 0:
            self.peer_digests = torch.zeros(self.world, dtype=torch.int64, device=device)
            total = sum(self.counts)
            self.all_adpcm = torch.empty((total, self.pitch), dtype=torch.uint8, device=device)
            self.all_coefs = torch.empty((total, 16), dtype=torch.int16, device=device)

    def gather(self, adpcm, coefs, async_op=False, nbytes=None):
        """adpcm [count, pitch] uint8, coefs [count, 16] int16 (this rank's rows).  Returns a list of work handles
        (empty when the transfer has completed).  With `nbytes` (the bytes of a row that carry data) every rank also
        sends the 64-bit digest of what it is sending (rows_digest over its global channel indices), which verify()
        holds against a digest recomputed over what arrived."""
        n = self.counts[self.rank]
        digest = None
        if nbytes is not None:
            digest = torch.tensor([_i64(rows_digest(adpcm[:n], nbytes, coefs[:n], self.firsts[self.rank]))],
                                  dtype=torch.int64, device=adpcm.device)
            self._keep = digest                            # alive until the transfer is done
            if self.rank == 0:
                self.peer_digests[0:1].copy_(digest)
        assert adpcm.shape[0] >= n and adpcm.shape[1] == self.pitch and coefs.shape[0] >= n
        if self.rank == 0:
            self.all_adpcm[:n].copy_(adpcm[:n], non_blocking=True)
            self.all_coefs[:n].copy_(coefs[:n].reshape(n, 16), non_blocking=True)
        if self.world == 1:
            return []
        coef_bytes = coefs[:n].reshape(n, 16).contiguous().view(torch.uint8)
        works = []
        if self.via_host:
            return self._gather_via_host(adpcm, coef_bytes, digest, async_op)
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
        if digest is not None:
            if self.rank == 0:
                ops = [dist.P2POp(dist.irecv, self.peer_digests[r:r + 1].view(torch.uint8), r) for r in range(1, self.world)]
            else:
                ops = [dist.P2POp(dist.isend, digest.view(torch.uint8), 0)]
            works.extend(dist.batch_isend_irecv(ops))
        if async_op:
            return works
        for w in works:
            w.wait()
        return []

    def _gather_via_host(self, adpcm, coef_bytes, digest, async_op):
        class Landing:                                  # a receive into host memory, copied to its place when waited for
            def __init__(self, work, host, dest):
                self.work, self.host, self.dest = work, host, dest

            def wait(self):
                self.work.wait()
                if self.dest is not None:
                    self.dest.copy_(self.host)

        n = self.counts[self.rank]
        works = []
        for c0 in range(0, max(self.counts), self.chunk):
            if self.rank == 0:
                for r in range(1, self.world):
                    c1 = min(c0 + self.chunk, self.counts[r])
                    if c1 > c0:
                        f = self.firsts[r]
                        for dest in (self.all_adpcm[f + c0:f + c1], self.all_coefs[f + c0:f + c1].view(torch.uint8)):
                            host = torch.empty(dest.shape, dtype=torch.uint8)
                            works.append(Landing(dist.irecv(host, r), host, dest))
            else:
                c1 = min(c0 + self.chunk, n)
                if c1 > c0:
                    for src in (adpcm[c0:c1], coef_bytes[c0:c1]):
                        host = src.cpu()
                        works.append(Landing(dist.isend(host, 0), host, None))
        if digest is not None:
            if self.rank == 0:
                for r in range(1, self.world):
                    host = torch.empty(8, dtype=torch.uint8)
                    works.append(Landing(dist.irecv(host, r), host, self.peer_digests[r:r + 1].view(torch.uint8)))
            else:
                host = digest.view(torch.uint8).cpu()
                works.append(Landing(dist.isend(host, 0), host, None))
        if async_op:
            return works
        for w in works:
            w.wait()
        return []

    def verify(self, adpcm, coefs, nbytes):
        """rank 0, after a gather(..., nbytes=nbytes): its own rows arrived unchanged, and for EVERY rank the digest
        recomputed over the rows that arrived (in the places they arrived at) equals the digest that rank sent.
        Returns a short description, "MISMATCH ..." naming the ranks that differ otherwise."""
        if self.rank != 0:
            return None
        n = self.counts[0]
        own = bool(torch.equal(self.all_adpcm[:n, :nbytes], adpcm[:n, :nbytes]) and
                   torch.equal(self.all_coefs[:n], coefs[:n].reshape(n, 16)))
        sent = [int(v) & ((1 << 64) - 1) for v in self.peer_digests.tolist()]
        bad = []
        for r in range(self.world):
            f, c = self.firsts[r], self.counts[r]
            got = rows_digest(self.all_adpcm[f:f + c], nbytes, self.all_coefs[f:f + c], f)
            if got != sent[r]:
                bad.append(r)
        if own and not bad:
            return "own rows identical; per-rank 64-bit digests of all %d channels match what each rank sent" % self.all_adpcm.shape[0]
        return "MISMATCH (own rows %s, ranks with a wrong digest: %s)" % ("ok" if own else "differ", bad)
