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


def gather_packed(payload: torch.Tensor, lens: torch.Tensor, group=None, dst: int = 0):
    """Rooted gather of one packed payload per rank.

    payload : 1-D uint8 tensor (on the communication device) holding this rank's streams back to
              back; lens : 1-D int64 tensor of their lengths on the same device.
    Rank ``dst`` gets ``(payloads, lengths)``: per rank a uint8 tensor trimmed to its byte count
    and its int64 length vector, both still on the device; other ranks get None.  Three
    collectives in all: a tiny all_gather of (image count, payload bytes) that sizes the padding, one
    all_gather of the padded length vectors, and one padded gather of the payload -- the only one that
    carries data, each non-root GPU sending straight to the root over its own xGMI link.
    """
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    device = payload.device
    meta = torch.tensor([lens.numel(), int(payload.numel())], dtype=torch.int64, device=device)
    metas = [torch.zeros(2, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(metas, meta, group=group)
    metas = [m.cpu() for m in metas]
    max_img = max(int(m[0]) for m in metas)
    slot = max(max(int(m[1]) for m in metas), 1)
    lens_pad = torch.zeros(max(max_img, 1), dtype=torch.int64, device=device)
    lens_pad[: lens.numel()] = lens
    all_lens = [torch.zeros_like(lens_pad) for _ in range(world)]
    dist.all_gather(all_lens, lens_pad, group=group)
    if payload.numel() == slot:
        buf = payload
    else:
        buf = torch.zeros(slot, dtype=torch.uint8, device=device)
        buf[: payload.numel()] = payload
    recv = [torch.empty(slot, dtype=torch.uint8, device=device) for _ in range(world)] if rank == dst else None
    dist.gather(buf, recv, dst=dst, group=group)
    if rank != dst:
        return None
    return ([recv[r][: int(metas[r][1])] for r in range(world)], [all_lens[r][: int(metas[r][0])] for r in range(world)])


def gather_streams(streams: Sequence[bytes | np.ndarray], device: torch.device, group=None, dst: int = 0) -> Optional[List[List[bytes]]]:
    """Every rank contributes its list of streams; rank ``dst`` receives ``[rank][image] -> bytes``
    (others get None).  Convenience wrapper over :func:`gather_packed` that stages through host
    memory on both ends."""
    payload, lens = pack(streams)
    got = gather_packed(torch.from_numpy(payload.copy()).to(device), torch.from_numpy(lens).to(device), group, dst)
    if got is None:
        return None
    out: List[List[bytes]] = []
    for data, ls in zip(*got):
        host, ls = data.cpu().numpy(), ls.cpu().numpy()
        offs = np.concatenate([[0], np.cumsum(ls)])
        out.append([host[int(offs[k]): int(offs[k + 1])].tobytes() for k in range(len(ls))])
    return out
