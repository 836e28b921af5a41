"""Multi-GPU: frames are independent after sync, so the batch is cut
into contiguous blocks of frames, one block per rank, each rank demodulates its own
block, and ONE all-gather of the bit-packed output (4 092 B/frame at config 2 --
never int64-per-bit, never symbols) makes the decoded payload available everywhere.

One process per GPU; `torch.distributed` backend "nccl" is RCCL on ROCm (xGMI inside
a node); "gloo" is used by the CPU tests.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Join the process group described by RANK/WORLD_SIZE/MASTER_* (torchrun)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend is None:
            backend = os.environ.get("GF3_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available():
            local = local % torch.cuda.device_count()     # rehearsals with more ranks than GPUs (gloo only)
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    elif torch.cuda.is_available():
        local = local % torch.cuda.device_count()
    return rank, world, local


def shard_range(F_total: int, rank: int, world: int):
    """Frames [lo, hi) owned by `rank`: contiguous blocks, remainder spread over the first ranks."""
    base, rem = divmod(F_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def cyclic_frame_index(rank: int, world: int, F_local: int, chunks: int):
    """Global frame numbers owned by `rank` under the block-cyclic sharding the overlapped gather
    uses: the batch is cut into `chunks` pieces of world*Fc frames, rank r owning block r of each
    piece, so a plain all_gather_into_tensor of chunk c lands in global frame order
    [c*world*Fc, (c+1)*world*Fc).  Returns int64 [F_local]."""
    assert F_local % chunks == 0
    Fc = F_local // chunks
    c = torch.arange(chunks).repeat_interleave(Fc)
    i = torch.arange(Fc).repeat(chunks)
    return c * (world * Fc) + rank * Fc + i


class OverlappedGather:
    """One all-gather per chunk on a side stream, issued as soon as the chunk's bits exist, so
    the xGMI transfer of chunk c runs under the kernels of chunk c+1.  `out` is [world*F_local, row]
    in the global order of cyclic_frame_index()."""

    def __init__(self, out: torch.Tensor, F_local: int, chunks: int):
        self.out, self.F, self.chunks = out, F_local, chunks
        self.Fc = F_local // chunks
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.active = dist.is_initialized()           # a one-rank group still runs the collective (rehearsal)
        self.side = torch.cuda.Stream() if out.is_cuda else None
        self.works = []

    def chunk_done(self, c: int, local_bits_chunk: torch.Tensor):
        if not self.active:
            return
        dst = self.out[c * self.world * self.Fc: (c + 1) * self.world * self.Fc]
        if self.side is None:
            dist.all_gather_into_tensor(dst, local_bits_chunk)
            return
        ev = torch.cuda.Event()
        ev.record()
        with torch.cuda.stream(self.side):
            self.side.wait_event(ev)
            self.works.append(dist.all_gather_into_tensor(dst, local_bits_chunk, async_op=True))

    def finish(self):
        for w in self.works:
            w.wait()
        self.works.clear()
        if self.side is not None:
            torch.cuda.current_stream().wait_stream(self.side)


def all_gather_bits(local_bits: torch.Tensor, F_total: int | None = None, out: torch.Tensor | None = None):
    """local_bits: [F_local, bytes_per_frame] uint8 -> [F_total, bytes_per_frame] on every rank,
    frames in global order.  Equal shards use a single all_gather_into_tensor."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local_bits
    world = dist.get_world_size()
    row = local_bits.shape[1]
    if F_total is None or F_total % world == 0:
        if out is None:
            out = torch.empty((local_bits.shape[0] * world, row), dtype=torch.uint8, device=local_bits.device)
        dist.all_gather_into_tensor(out, local_bits.contiguous())
        return out
    # ragged split: pad every shard to the largest one, still a single collective, then trim
    sizes = [hi - lo for lo, hi in (shard_range(F_total, r, world) for r in range(world))]
    mx = max(sizes)
    padded = torch.zeros((mx, row), dtype=torch.uint8, device=local_bits.device)
    padded[: local_bits.shape[0]] = local_bits
    buf = torch.empty((world * mx, row), dtype=torch.uint8, device=local_bits.device)
    dist.all_gather_into_tensor(buf, padded)
    return torch.cat([buf[r * mx: r * mx + sizes[r]] for r in range(world)], dim=0)


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def max_over_ranks(x: float, device) -> float:
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return x
    t = torch.tensor([x], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_floats(x: float, device):
    """[x of rank 0, x of rank 1, ...] on every rank (one small all-gather; [x] without a process group)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [float(x)]
    t = torch.tensor([x], dtype=torch.float64, device=device)
    out = torch.empty(dist.get_world_size(), dtype=torch.float64, device=device)
    dist.all_gather_into_tensor(out, t)
    return [float(v) for v in out.cpu()]


def spawn_ranks(cmd, n, port=None, env=None, relay=None, poll_s=0.2, grace_s=10.0):
    """Start `cmd` (an argv list) n times as FRESH processes, one per rank, with the torchrun environment
    (RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR=127.0.0.1, MASTER_PORT) and wait for all of them.

    What `python bench.py --gpus N` does when it was not started under torchrun.  The caller must not have touched
    the GPU: the children are plain subprocess.Popen children (never os.exec*), each initialises its own device.
    Rank 0's stdout is relayed line by line through `relay` (default: print), the other ranks' stdout is dropped
    (they print nothing by contract), stderr is inherited.  If a rank exits non-zero the others are given
    `grace_s` seconds and then terminated by their exact PIDs.  Whatever ends the wait -- normal completion, an
    exception, KeyboardInterrupt, a SIGTERM turned into SystemExit by the caller's handler -- no child is left
    behind: every one still running is terminated, then killed, by its exact PID.
    Returns (rc, rank-0 stdout lines): rc is 0 only if every rank returned 0, else the exit code of the rank that
    failed FIRST of its own accord (not the -15 of a straggler this function terminated afterwards).
    port: rendezvous port; None picks a free one (the probe socket stays bound until the children exist, which
    narrows, but cannot close, the window in which another job could take the port: pass a port to be sure)."""
    import socket
    import subprocess
    import sys
    import threading
    import time
    probe = None
    if port is None:
        probe = socket.socket()
        probe.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        probe.bind(("127.0.0.1", 0))
        port = probe.getsockname()[1]
    relay = relay or (lambda line: (sys.stdout.write(line + "\n"), sys.stdout.flush()))
    procs, lines = [], []
    first_failure = None                                      # (rank, code) of the first rank that failed by itself
    pump_thread = None
    try:
        for r in range(n):
            e = dict(os.environ if env is None else env)
            e.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                     MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            procs.append(subprocess.Popen(list(cmd), env=e, text=True,
                                          stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
        if probe is not None:
            probe.close()                                     # (the ranks bind it seconds later, after their imports)
            probe = None

        def pump():
            for line in procs[0].stdout:
                line = line.rstrip("\n")
                lines.append(line)
                relay(line)
        pump_thread = threading.Thread(target=pump, daemon=True)
        pump_thread.start()
        failed_at = None
        while any(p.poll() is None for p in procs):
            if first_failure is None:
                for r, p in enumerate(procs):
                    if p.poll() not in (None, 0):
                        first_failure, failed_at = (r, p.returncode), time.monotonic()
                        break
            if failed_at is not None and time.monotonic() - failed_at > grace_s:
                break                                         # the finally block ends the stragglers
            time.sleep(poll_s)
    finally:
        if probe is not None:
            probe.close()
        for p in procs:                                       # exact PIDs of our own children only
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=5)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
        if pump_thread is not None:
            pump_thread.join(timeout=5)
    if first_failure is None:                                 # (a failure between the last poll and the loop's exit)
        first_failure = next(((r, p.returncode) for r, p in enumerate(procs) if p.returncode != 0), None)
    return (first_failure[1] if first_failure else 0), lines
