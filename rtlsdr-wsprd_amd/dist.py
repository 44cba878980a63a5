"""Multi-GPU fan-out / fan-in for the WSPR decode path.

Segments are independent (wspr_decode keeps no state between calls when
usehashtable = 0, wsprd.c:478-479), so the path shards by segment with NO data-path
collective: one process per GPU decodes its own contiguous block of segments.
torch.distributed (backend "nccl" = RCCL over xGMI on MI355X, "gloo" on CPU for
tests) is used only to
  * broadcast the decoder options from rank 0 (40-byte struct), and
  * gather the fixed-size spot records {n, decoder_results[K]} on rank 0.
Payloads are KB..MB per rank, i.e. latency-bound on any xGMI link.
"""
import ctypes as C

import numpy as np
import torch
import torch.distributed as dist


def shard_range(nseg_total, rank, world):
    """Contiguous block of segments owned by `rank` (SURVEY §8e partitioning)."""
    base, rem = divmod(nseg_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _dev():
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")


def broadcast_options(opt, src=0):
    """Broadcast a decoder_options ctypes struct from `src`; returns the received struct."""
    raw = np.frombuffer(bytes(opt), dtype=np.uint8).copy()
    t = torch.from_numpy(raw).to(_dev())
    dist.broadcast(t, src=src)
    out = type(opt).from_buffer_copy(t.cpu().numpy().tobytes())
    return out


def pack_spots(results_array, counts, nseg, max_results, record_size):
    """ctypes decoder_results[nseg*K] + int[nseg] -> uint8 tensor [nseg, 4 + K*record_size]."""
    rec = np.frombuffer(results_array, dtype=np.uint8).reshape(nseg, max_results * record_size)
    cnt = np.frombuffer(counts, dtype=np.int32).reshape(nseg, 1).view(np.uint8).reshape(nseg, 4)
    return torch.from_numpy(np.concatenate([cnt, rec], axis=1))


def gather_spots(packed, dst=0):
    """All ranks hold the same number of segments (weak scaling) -> one fixed-size gather.
    Returns on dst a uint8 tensor [world, nseg, rec] (CPU), elsewhere None."""
    world, rank = dist.get_world_size(), dist.get_rank()
    t = packed.to(_dev())
    if dist.get_backend() == "nccl":
        # RCCL gather
        bufs = [torch.empty_like(t) for _ in range(world)] if rank == dst else None
        dist.gather(t, bufs, dst=dst)
    else:
        bufs = [torch.empty_like(t) for _ in range(world)] if rank == dst else None
        dist.gather(t, bufs, dst=dst)
    if rank != dst:
        return None
    return torch.stack(bufs).cpu()


def unpack_counts(gathered):
    """[world, nseg, rec] uint8 -> int32 [world, nseg] spot counts."""
    return gathered[:, :, :4].contiguous().numpy().view(np.int32).reshape(gathered.shape[0], gathered.shape[1])


def unpack_messages(gathered, max_results, record_size, msg_offset=28, msg_len=23):
    """Decoded message strings per (rank, segment) from a gathered tensor."""
    g = gathered.numpy()
    counts = unpack_counts(gathered)
    out = []
    for r in range(g.shape[0]):
        segs = []
        for s in range(g.shape[1]):
            msgs = []
            for k in range(int(counts[r, s])):
                o = 4 + k * record_size + msg_offset
                raw = bytes(g[r, s, o:o + msg_len])
                msgs.append(raw.split(b"\0", 1)[0].decode())
            segs.append(msgs)
        out.append(segs)
    return out
