"""Rooted gather of variable-length .nblic streams across ranks (one process per GPU).

Images are independent, so the hot path shards with no data-path collective; the only
exchange is this one gather of the finished streams to rank 0.  With the
``nccl`` backend (RCCL on ROCm) the payload moves GPU->GPU over xGMI; with ``gloo`` the same
code runs on CPU tensors, which is how tests/test_distributed.py covers it.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np
import torch
import torch.distributed as dist


def pack(streams: Sequence[bytes | np.ndarray]):
    """Concatenate streams -> (uint8 payload, int64 lengths)."""
    arrs = [np.frombuffer(s, np.uint8) if isinstance(s, (bytes, bytearray)) else np.asarray(s, np.uint8) for s in streams]
    lens = np.array([a.size for a in arrs], np.int64)
    payload = np.concatenate(arrs) if arrs else np.zeros(0, np.uint8)
    return payload, lens


def gather_streams(streams: Sequence[bytes | np.ndarray], device: torch.device, group=None, dst: int = 0) -> Optional[List[List[bytes]]]:
    """Every rank contributes its list of streams; rank ``dst`` receives ``[rank][image] -> bytes``
    (others get None).  Two collectives: an all_gather of the per-image lengths and one padded
    gather of the payload."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    payload, lens = pack(streams)
    n_img = torch.tensor([len(lens)], dtype=torch.int64, device=device)
    counts = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(counts, n_img, group=group)
    max_img = int(max(int(c.item()) for c in counts))
    lens_t = torch.zeros(max_img, dtype=torch.int64, device=device)
    lens_t[: len(lens)] = torch.from_numpy(lens).to(device)
    all_lens = [torch.zeros(max_img, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(all_lens, lens_t, group=group)
    totals = [int(l.sum().item()) for l in all_lens]
    slot = max(max(totals), 1)
    buf = torch.zeros(slot, dtype=torch.uint8, device=device)
    if payload.size:
        buf[: payload.size] = torch.from_numpy(payload.copy()).to(device)
    recv = [torch.zeros(slot, dtype=torch.uint8, device=device) for _ in range(world)] if rank == dst else None
    dist.gather(buf, recv, dst=dst, group=group)
    if rank != dst:
        return None
    out: List[List[bytes]] = []
    for r in range(world):
        host = recv[r].cpu().numpy()
        ls = all_lens[r].cpu().numpy()[: int(counts[r].item())]
        offs = np.concatenate([[0], np.cumsum(ls)])
        out.append([host[int(offs[k]): int(offs[k + 1])].tobytes() for k in range(len(ls))])
    return out
