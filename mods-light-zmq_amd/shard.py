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


# ---- one hard pair: views sharded over the ranks ---------------------------------------------------------
REGION_BYTES = 208      # sizeof(mods_region)


def exchange_blocks(local_blocks, n_jobs, dist, device=None, rec_bytes=REGION_BYTES):
    """The one exchange step of the view-sharded path.  `local_blocks`: {job index: uint8 torch tensor holding
    k * rec_bytes bytes} for the jobs this rank computed (every job is computed by exactly one rank).
    Returns the list of all n_jobs blocks, in job order, on every rank: first the counts (one small
    all-reduce), then ONE all-gather of the per-rank payloads padded to the largest.  With the RCCL
    backend the payload stays in HBM; gloo (CPU tests, or several ranks on one GPU) stages through the host."""
    import torch
    world = dist.get_world_size()
    on_host = dist.get_backend() == "gloo"
    dev = "cpu" if on_host else device
    counts = torch.zeros(n_jobs, dtype=torch.int64, device=dev)
    for j, b in local_blocks.items():
        counts[j] = b.numel() // rec_bytes
    owner = torch.full((n_jobs,), -1, dtype=torch.int64, device=dev)
    for j in local_blocks:
        owner[j] = dist.get_rank()
    dist.all_reduce(counts, op=dist.ReduceOp.SUM)
    dist.all_reduce(owner, op=dist.ReduceOp.MAX)
    counts_h = counts.cpu().tolist()
    owner_h = owner.cpu().tolist()
    if min(owner_h) < 0:
        raise RuntimeError("view job %d was computed by no rank" % owner_h.index(min(owner_h)))
    per_rank = [sum(counts_h[j] for j in range(n_jobs) if owner_h[j] == r) for r in range(world)]
    cap = max(max(per_rank), 1) * rec_bytes
    mine = [j for j in range(n_jobs) if owner_h[j] == dist.get_rank()]
    payload = torch.zeros(cap, dtype=torch.uint8, device=dev)
    off = 0
    for j in mine:
        b = local_blocks[j]
        payload[off:off + b.numel()] = b.to(dev) if on_host else b
        off += b.numel()
    bufs = [torch.empty_like(payload) for _ in range(world)]
    dist.all_gather(bufs, payload)
    out = [None] * n_jobs
    offs = [0] * world
    for j in range(n_jobs):
        r = owner_h[j]
        nb = counts_h[j] * rec_bytes
        blk = bufs[r][offs[r]:offs[r] + nb]
        out[j] = blk.to(device) if (on_host and device is not None) else blk
        offs[r] += nb
    return out, counts_h


def match_ladder_distributed(pkg, ctx, img_ptr, w, h, steps, dist, device, params=None, min_matches=15, seed_time=None):
    """mods.cpp:202-383 with the views of every step sharded over the ranks (largest first), one exchange
    step per ladder step, the FGINN search split by query rows, and duplicate filtering + LO-RANSAC on
    rank 0.  Every rank returns the same dict.  img_ptr: [2][h][w] fp32 in this rank's HBM."""
    import ctypes as C
    import torch
    params = params or pkg.PairParams.default()
    rank, world = dist.get_rank(), dist.get_world_size()
    rep1, rep2 = pkg.ImgRep(ctx), pkg.ImgRep(ctx)
    history = []
    plane_bytes = w * h * 4
    res = None
    n_views = 0
    for si, step in enumerate(steps):
        views = pkg.view_schedule(step, history)
        jobs = [(im, v) for im in (0, 1) for v in views]
        areas = []
        for _, (zoom, tilt, phi) in jobs:
            g = pkg.view_geometry(w, h, tilt, phi, zoom, step.initSigma)
            areas.append(float(g.w_new * g.h_new))
        mine = largest_first_views(areas, world)[rank]
        local = {}
        # a step whose descriptor list names a Half* descriptor runs the orientation of every view in doHalfSIFT mode
        # (imagerepresentation.cpp:725-731; csrc/imgrep.hip does the same for the one-GPU ladder).  HalfRootSIFT LISTS
        # (fginn_ratio_half > 0) are the C++ form's business (mods_match_ladder_groups_multi), not this harness's.
        if step.fginn_ratio_half > 0:
            raise ValueError("match_ladder_distributed: HalfRootSIFT lists are not supported by the Python harness")
        desc = type(params.desc).from_buffer_copy(params.desc)
        desc.ori_halfMode = 1 if step.half_orientation else 0
        for j in mine:
            im, (zoom, tilt, phi) = jobs[j]
            _, _, nr = ctx.detect_describe_view_dev(img_ptr + im * plane_bytes, w, h, tilt, phi, zoom, step.initSigma, step.doBlur,
                                                    params.det, desc)
            blk = torch.empty(nr * REGION_BYTES, dtype=torch.uint8, device=device)
            if nr:
                ctx.regions_copy_dev(0, blk.data_ptr(), nr)
            local[j] = blk
        blocks, counts = exchange_blocks(local, len(jobs), dist, device)
        n_views += len(jobs)
        for j, (im, _) in enumerate(jobs):
            if counts[j]:
                (rep2 if im else rep1).append_dev(blocks[j].data_ptr(), counts[j])
        ctx.sync()
        del blocks
        n = len(rep1)
        q0, q1 = n * rank // world, n * (rank + 1) // world
        tent, u6, laf = pkg.match_reps(ctx, rep1, rep2, q0, q1, step.fginn_ratio, params.contradDist, params.nn)
        tent_all, _ = allgather_ragged(tent, dist, None if dist.get_backend() == "gloo" else device)
        u6_all, _ = allgather_ragged(u6, dist, None if dist.get_backend() == "gloo" else device)
        laf_all, _ = allgather_ragged(laf, dist, None if dist.get_backend() == "gloo" else device)
        out = np.zeros(16, np.float64)
        if rank == 0:
            tent_v, u6_v, _, nu, Hm, stats = pkg.verify_tentatives(tent_all, u6_all, laf_all, params, device=ctx.device, seed_time=seed_time)
            ninl = len(tent_v)
            res = dict(steps_done=si + 1, n_views=n_views, n_described=[len(rep1), len(rep2)], n_tentatives=len(tent_all),
                       n_unique=nu, n_inliers=ninl, stats=stats, H=np.asarray(Hm).reshape(-1),
                       matches=u6_v[:, [0, 1, 3, 4]] if ninl else np.zeros((0, 4)))
            out[0] = ninl
        t = torch.from_numpy(out)
        if dist.get_backend() != "gloo":
            t = t.to(device)
        dist.broadcast(t, src=0)
        ninl = int(t.cpu()[0].item())
        if ninl >= min_matches:
            break
    # every rank returns rank 0's summary
    obj = [res]
    dist.broadcast_object_list(obj, src=0)
    rep1.close(); rep2.close()
    return obj[0]
