"""Multi-GPU host logic: one process per GPU, torch.distributed (RCCL on the GPU box, gloo in the CPU
tests).

Two ways the hot path shards:
  * throughput - pairs are independent units: round-robin over the ranks, no data-path collective;
  * one hard pair - synthesised views of an image are independent up to matching: each rank detects and
    describes its views, then ONE exchange step all-gathers {descriptors, centres, (view, index)} so that
    every rank holds the full train set.  Per-rank counts differ, so counts are exchanged first and the
    payload is padded to the maximum (a single fixed-size all-gather; on the xGMI mesh a one-shot
    all-gather of a few MB beats a ring).
"""
import numpy as np


def shard_pairs(n_pairs, rank, world):
    """Global pair indices owned by `rank` (round robin: pair p -> rank p % world)."""
    return list(range(rank, n_pairs, world))


def largest_first_views(view_areas, world):
    """Greedy longest-processing-time assignment of views to ranks (tilted views are much smaller than
    the frontal one).  Returns a list of view-index lists, one per rank; deterministic."""
    order = sorted(range(len(view_areas)), key=lambda i: (-view_areas[i], i))
    load = [0.0] * world
    out = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda q: (load[q], q))
        out[r].append(i)
        load[r] += view_areas[i]
    return out


def allgather_ragged(local, dist, device=None):
    """All-gather of per-rank arrays with different leading sizes.  `local`: numpy array [n_i, ...] (same
    trailing shape and dtype on every rank).  Returns (concatenation in rank order, counts)."""
    import torch
    world = dist.get_world_size()
    dev = device if device is not None else "cpu"
    n = torch.tensor([local.shape[0]], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n)
    counts = [int(c.item()) for c in counts]
    cap = max(max(counts), 1)
    pad = np.zeros((cap,) + local.shape[1:], local.dtype)
    pad[:local.shape[0]] = local
    flat = torch.from_numpy(pad.view(np.uint8).reshape(cap, -1)).to(dev)
    bufs = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(bufs, flat)
    parts = []
    for r in range(world):
        a = bufs[r].cpu().numpy().reshape(-1).view(local.dtype).reshape((cap,) + local.shape[1:])
        parts.append(a[:counts[r]])
    return np.concatenate(parts, axis=0), counts


def gather_pair_results(local_ids, local_vals, n_pairs, dist, device=None):
    """Every rank contributes {pair id: value}; returns the length-n_pairs vector on every rank."""
    rec = np.zeros(len(local_ids), np.dtype([("id", "i8"), ("v", "f8")]))
    rec["id"] = local_ids
    rec["v"] = local_vals
    allrec, _ = allgather_ragged(rec, dist, device)
    out = np.full(n_pairs, np.nan)
    out[allrec["id"]] = allrec["v"]
    return out
