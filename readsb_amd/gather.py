"""Aggregator step of the multi-GPU job: every rank demodulates its own stream (or its own
time-chunk of one capture), then the decoded message counts and the fixed-size 64-byte message
records are gathered on rank 0 — the role TCP beast forwarding between readsb daemons plays in
the reference (`--net-connector`, README.md:40-51; net_io.c:1655).  No collective is needed
inside the data path: buffers are independent.

Backend-agnostic (`nccl` = RCCL over xGMI on the MI355X node, `gloo` in the CPU tests)."""
import numpy as np
import torch
import torch.distributed as dist


def gather_messages(msgs: np.ndarray, device: torch.device, dst: int = 0):
    """msgs: structured array of 64-byte mgpu_msg records of THIS rank.
    Returns (counts list for every rank, list of per-rank record arrays on dst / None elsewhere)."""
    world, rank = dist.get_world_size(), dist.get_rank()
    rec = msgs.dtype.itemsize
    cnt = torch.tensor([len(msgs)], dtype=torch.int64, device=device)
    counts = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(counts, cnt)                       # 8 bytes per rank
    counts = [int(c.item()) for c in counts]
    cap = max(counts) if counts else 0
    buf = torch.zeros(max(cap, 1) * rec, dtype=torch.uint8, device=device)
    if len(msgs):
        flat = torch.from_numpy(np.ascontiguousarray(msgs).view(np.uint8).reshape(-1))
        buf[: flat.numel()] = flat.to(device)
    gathered = [torch.zeros_like(buf) for _ in range(world)] if rank == dst else None
    dist.gather(buf, gathered, dst=dst)                # padded fixed-size records
    if rank != dst:
        return counts, None
    out = []
    for r in range(world):
        raw = gathered[r][: counts[r] * rec].cpu().numpy()
        out.append(raw.view(msgs.dtype).copy())
    return counts, out


def merge_by_timestamp(per_rank):
    """Rank-0 side: one list ordered by the 12 MHz timestamp (stable for equal stamps)."""
    allm = np.concatenate(per_rank) if per_rank else np.zeros(0)
    order = np.argsort(allm["timestamp"], kind="stable")
    return allm[order]
