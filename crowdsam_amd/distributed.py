"""Image sharding + the one collective of the path: the final detection gather.

The reference shards images over GPUs with one ``tools/test.py`` subprocess per GPU and gathers through
``temp_result_{rank}.json`` files (tools/batch_eval.py:8-29,80-95).  Here: one process per GPU under
``torch.distributed`` (backend "nccl" == RCCL over xGMI on ROCm; "gloo" in the CPU tests), the same
contiguous index ranges, and ONE variable-length all-gather of fixed-width detection rows
``(image_index, x0, y0, x1, y1, score)`` at the end of the run (all_gather of counts, then a padded
all_gather).  Volume is tens of MB at most: latency-bound, xGMI link bandwidth is irrelevant.
There is no collective inside the per-image path.
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_range(n_items, rank, world):
    """Contiguous [start, end) of tools/batch_eval.py:80-89: floor(n/world) each, last rank takes the rest."""
    per = n_items // world
    start = rank * per
    end = n_items if rank == world - 1 else start + per
    return start, end


def gather_rows(rows, device=None):
    """All-gather variable-length float32 rows [n_i, C] from every rank -> [sum n_i, C] on every rank,
    ordered by rank (== image order for contiguous shards)."""
    rows = np.ascontiguousarray(rows, dtype=np.float32)
    if rows.ndim != 2:
        raise ValueError("rows must be [n, C]")
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return rows
    world = dist.get_world_size()
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    C = rows.shape[1]
    n = torch.tensor([rows.shape[0]], dtype=torch.int64, device=device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n)
    counts = [int(c.item()) for c in counts]
    cap = max(max(counts), 1)
    buf = torch.zeros((cap, C), dtype=torch.float32, device=device)
    if rows.shape[0]:
        buf[: rows.shape[0]] = torch.from_numpy(rows).to(device)
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    return np.concatenate([o[:c].cpu().numpy() for o, c in zip(out, counts)], axis=0)


def detections_to_rows(image_index, boxes, scores):
    boxes = np.asarray(boxes, np.float32).reshape(-1, 4)
    scores = np.asarray(scores, np.float32).reshape(-1, 1)
    idx = np.full((len(boxes), 1), image_index, np.float32)
    return np.concatenate([idx, boxes, scores], axis=1)
