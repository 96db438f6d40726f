#!/usr/bin/env python3
"""Headline benchmark: image pairs / second, end to end (detect + describe + match + RANSAC) on
synthetic 1920x1080 pairs, HessianAffine + RootSIFT, one identity view per image
(BASELINE.json configs[1]).

  python bench.py --gpus N --steps K --warmup W

A step = one batch of --pairs-per-step (default 32) image pairs through the whole hot path (detect, describe,
match, duplicate filter, LO-RANSAC) with all images already resident in HBM; the reported value is pairs / second.  N > 1 is launched by torch.distributed.run, one rank per GPU; pairs are independent
units, so every rank works on its own pairs (weak scaling, no data-path collective) and the reported
value is all pairs / max-over-ranks time.

Rank 0 prints one JSON line.  Besides the contract fields it carries
  roofline     - the Gaussian-blur kernel of the Hessian pyramid (largest share of the HBM-bound
                 pyramid time): algorithmic bytes (8 B/px per blur launch) / mean launch time
                 measured with HIP events on the context's stream during the timed steps
  cpu_baseline - the CPU oracle (oracle/) timed on this host on one pair of the same workload
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

W, H = 1920, 1080
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md


def cpu_baseline(img1, img2, seed):
    """Oracle chain on one pair, single thread.  Returns (pairs_per_s, seconds, inliers)."""
    import pipeline_oracle as po
    import refdeg
    t0 = time.time()
    if refdeg.available():
        r = po.match_pair(img1, img2, seed_time=seed)
        ninl = r["n_inliers"]
    else:   # GPU box without oracle/_ref: everything up to the tentatives (RANSAC is <1% of the CPU time)
        import orc
        ra, _ = orc.detect_describe(img1)
        rb, _ = orc.detect_describe(img2)
        tc = orc.match_fginn(ra, rb, 0.8)
        ninl = len(orc.duplicate_filter(tc, ra, rb, 2.0, 1))
    dt = time.time() - t0
    return 1.0 / dt, dt, ninl


MFMA_I8_PEAK_TOPS = 5000.0   # dense int8 = 2 x the bf16 rate (MI355X_MICROARCH.md MFMA table; measured ceiling >= 3944)


def _hard_pair(synth, np, w, h, seed, tilt=6.0):
    """image 2 = image 1 under a strong tilt (anisotropy `tilt` at 35 degrees) + rotation."""
    a = synth.texture(w, h, seed)
    ang = np.deg2rad(35.0)
    R = np.array([[np.cos(ang), -np.sin(ang)], [np.sin(ang), np.cos(ang)]])
    A = R @ np.diag([1.0, 1.0 / tilt]) @ R.T @ np.array([[np.cos(0.3), -np.sin(0.3)], [np.sin(0.3), np.cos(0.3)]])
    c = np.array([w / 2.0, h / 2.0])
    Hm = np.eye(3)
    Hm[:2, :2] = A
    Hm[:2, 2] = c - A @ c
    return a, synth.warp(a, Hm, seed=seed)


def other_configs(args):
    """Measurement of the non-headline configurations (not what the driver runs; same JSON shape)."""
    import numpy as np
    import torch
    import __graft_entry__ as ge
    import synth
    pkg = ge.load_package()
    if pkg.lib().mods_device_count() <= 0:
        raise SystemExit("bench.py needs an MI355X: libmodsgpu has no CPU path")
    torch.cuda.set_device(0)
    pkg.ransac_pin_seed(12345)
    out = {"metric": "image_pairs_per_sec_end_to_end", "unit": "pairs/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic"}
    if args.config == "c4":
        w = h = 1024
        pairs = [synth.pair(w, h, seed=4000 + i) for i in range(args.pairs)]
        dev = [torch.from_numpy(np.stack([a, b])).cuda() for a, b, _ in pairs]
        params = pkg.PairParams.default()
        pipe = pkg.Pipeline(0, w, h, params, args.gpu_workers, args.verify_workers, args.pairs_per_batch)

        def run(n):
            res, pending = [], 0
            for i in range(n):
                if pending >= pipe.capacity - 1:
                    res.append(pipe.next()[0]); pending -= 1
                pipe.submit(dev[i % len(dev)].data_ptr(), i); pending += 1
            while pending:
                res.append(pipe.next()[0]); pending -= 1
            return res
        pps = max(1, args.pairs_per_step)
        run(args.warmup * pps)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        res = run(args.steps * pps)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        out.update(value=round(len(res) / dt, 3), ms_per_step=round(dt / args.steps * 1e3, 4),
                   config={"workload": "1024x1024 pairs, HessianAffine+RootSIFT, LO-RANSAC H (BASELINE configs[3], per-GPU rate)",
                           "pairs_per_step": pps,
                           "overlap": "%d gpu workers + %d verify workers" % (args.gpu_workers, args.verify_workers),
                           "keypoints_per_image": list(res[-1].n_described), "mean_inliers": round(sum(r.n_inliers for r in res) / len(res), 1)})
        pipe.close()
    elif args.config == "c5":
        w = h = 4096
        a = synth.texture(w, h, 5000, blobs=30000)
        Hm = synth.random_homography(np.random.default_rng(5000 + 104729), w, h)
        b = synth.warp(a, Hm, seed=5000)
        t = torch.from_numpy(np.stack([a, b])).cuda()
        ctx = pkg.Context(0, w, h, 2)
        params = pkg.PairParams.default()
        params.ransac.useF = 1
        for _ in range(args.warmup):
            pkg.match_pair_dev(ctx, t.data_ptr(), w, h, params)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        res = [pkg.match_pair_dev(ctx, t.data_ptr(), w, h, params)[0] for _ in range(args.steps)]
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        ctx.timing_enable(["match"]); ctx.timing_reset()
        last = pkg.match_pair_dev(ctx, t.data_ptr(), w, h, params)[0]
        mms, mn, _ = ctx.timing_read("match")
        # one pass over the N x M x 128 contraction (pass 1; pass 2 now runs on the few undecided queries only and is not counted)
        flops = 2.0 * last.n_described[0] * last.n_described[1] * 128
        ach = flops / (mms * 1e-3) / 1e12 if mms else 0.0
        out.update(value=round(args.steps / dt, 4), ms_per_step=round(dt / args.steps * 1e3, 3),
                   config={"workload": "4096x4096 pair, exact FGINN match + DEGENSAC F (BASELINE configs[4])",
                           "keypoints_per_image": list(last.n_described), "tentatives": last.n_tentatives, "inliers": last.n_inliers,
                           "ransac_samples": last.ransac_samples, "ransac_lo": last.ransac_lo,
                           "stage_ms": {"detect_describe": round(last.ms_detect_describe, 2), "match": round(last.ms_match, 2),
                                        "duplicates": round(last.ms_duplicates, 2), "ransac": round(last.ms_ransac, 2)}},
                   roofline={"kernel": "match stage (pack + match_nn1_kernel + mid + match_fginn_kernel + emit, i8 MFMA; ops of pass 1 only)", "bound": "mfma", "achieved": round(ach, 2),
                             "peak": MFMA_I8_PEAK_TOPS, "unit": "TOP/s", "frac": round(ach / MFMA_I8_PEAK_TOPS, 4), "traffic": None,
                             "stage_ms": round(mms, 3)})
        ctx.close()
    else:   # c3
        w, h = 1920, 1080
        a, b = _hard_pair(synth, np, w, h, 3000)
        t = torch.from_numpy(np.stack([a, b])).cuda()
        d = pkg.view_ctx_dims(w, h)
        ctx = pkg.Context(0, d[0], d[1], 1)
        rep1, rep2 = pkg.ImgRep(ctx, 1 << 20), pkg.ImgRep(ctx, 1 << 20)
        steps = pkg.iters_mods_steps()
        for _ in range(args.warmup):
            pkg.match_ladder_dev(ctx, t.data_ptr(), w, h, steps, rep1, rep2)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        res = [pkg.match_ladder_dev(ctx, t.data_ptr(), w, h, steps, rep1, rep2)[0] for _ in range(args.steps)]
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        r = res[-1]
        out.update(value=round(args.steps / dt, 4), ms_per_step=round(dt / args.steps * 1e3, 3),
                   config={"workload": "iters_MODS.ini HessianAffine steps on one hard 1920x1080 pair (tilt 6), 1 GPU (BASELINE configs[2])",
                           "steps_done": r.steps_done, "views": r.n_views, "regions": list(r.n_described), "tentatives": r.n_tentatives,
                           "inliers": r.n_inliers,
                           "stage_ms": {"synth_detect_describe": round(r.ms_detect_describe, 2), "match": round(r.ms_match, 2),
                                        "duplicates": round(r.ms_duplicates, 2), "ransac": round(r.ms_ransac, 2)}})
        rep1.close(); rep2.close(); ctx.close()
    print(json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--pairs", type=int, default=4, help="distinct synthetic pairs cycled through the steps")
    ap.add_argument("--pairs-per-step", type=int, default=32, help="image pairs in the batch that one step processes")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gpu-workers", type=int, default=3, help="pipeline threads running detect/describe/match (one context each)")
    ap.add_argument("--verify-workers", type=int, default=6, help="pipeline threads running duplicate filter + LO-RANSAC")
    ap.add_argument("--pairs-per-batch", type=int, default=8, help="pairs a GPU worker pushes through detect/describe as one batch of launches")
    ap.add_argument("--serial", action="store_true", help="no cross-pair overlap: one mods_match_pair_dev call per step")
    ap.add_argument("--config", default="c2", choices=["c2", "c3", "c4", "c5"],
                    help="BASELINE.json configs[]: c2 = the headline 1080p pair (default, what the driver runs); c3 = view-synthesis "
                         "ladder on a hard 1080p pair; c4 = 1-MP pairs (throughput); c5 = 4096x4096 pair with DEGENSAC F verification")
    args = ap.parse_args()
    if args.config != "c2":
        return other_configs(args)

    import numpy as np
    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge
    import synth

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # MODS_BENCH_SHARE_GPU=1 (development aid for 1-GPU boxes): every rank uses GPU 0 and the ranks meet over gloo, so that the
    # multi-rank control flow can be exercised without a multi-GPU node
    share = os.environ.get("MODS_BENCH_SHARE_GPU") == "1"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    device = local_rank if (world > 1 and not share) else 0
    torch.cuda.set_device(device)

    pkg = ge.load_package()
    if pkg.lib().mods_device_count() <= 0:
        raise SystemExit("bench.py needs an MI355X: libmodsgpu has no CPU path")

    # synthetic inputs: seed = 1000*config + pair index (config 2 = the 1080p pair), distinct per rank
    pairs_host = [synth.pair(W, H, seed=2000 + rank * 100 + i) for i in range(args.pairs)]
    pairs_dev = [torch.from_numpy(np.stack([a, b])).cuda(device) for a, b, _ in pairs_host]
    torch.cuda.synchronize()

    ctx = pkg.Context(device, W, H, 2)
    pkg.lib().mods_ransac_set_device(device)
    params = pkg.PairParams.default()
    pkg.ransac_pin_seed(12345)
    pipe = None if args.serial else pkg.Pipeline(device, W, H, params, args.gpu_workers, args.verify_workers, args.pairs_per_batch)

    def step(i):
        res, _ = pkg.match_pair_dev(ctx, pairs_dev[i % len(pairs_dev)].data_ptr(), W, H, params)
        return res

    def run(n_steps):
        """n_steps pairs through the hot path; returns the per-pair results in step order."""
        if pipe is None:
            return [step(i) for i in range(n_steps)]
        out, pending = [], 0
        for i in range(n_steps):
            if pending >= pipe.capacity - 1:
                out.append(pipe.next()[0]); pending -= 1
            pipe.submit(pairs_dev[i % len(pairs_dev)].data_ptr(), i); pending += 1
        while pending:
            out.append(pipe.next()[0]); pending -= 1
        return out

    pps = max(1, args.pairs_per_step)
    run(args.warmup * pps)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    results = run(args.steps * pps)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    n_pairs = len(results)
    inl = sum(r.n_inliers for r in results)
    stage_ms = [sum(getattr(r, f) for r in results) for f in ("ms_detect_describe", "ms_match", "ms_duplicates", "ms_ransac")]
    # roofline leg: the pyramid blur launches of the same workload in the same batching, bracketed by HIP events on
    # the worker's stream (a separate short pass with ONE gpu worker: event recording does not perturb the timed region
    # and a second stream does not stretch the kernels that are being timed)
    if pipe is None:
        ctx.timing_enable(["blur", "blur_small"]); ctx.timing_reset()
        for i in range(8):
            step(i)
        blur_ms, blur_n, blur_bytes = ctx.timing_read("blur")
        small_ms, small_n, small_bytes = ctx.timing_read("blur_small")
        ctx.timing_enable([])
    else:
        # what a GPU worker launches for one batch: the images of pairs_per_batch pairs in one detect/describe pass
        nb = max(1, args.pairs_per_batch)
        batch_t = torch.cat([pairs_dev[i % len(pairs_dev)] for i in range(nb)], dim=0).contiguous()
        bctx = pkg.Context(device, W, H, 2 * nb)
        torch.cuda.synchronize()
        for _ in range(2):
            bctx.detect_describe_dev(batch_t.data_ptr(), 2 * nb, W, H, params.det, params.desc)
        bctx.timing_enable(["blur", "blur_small"]); bctx.timing_reset()
        for _ in range(6):
            bctx.detect_describe_dev(batch_t.data_ptr(), 2 * nb, W, H, params.det, params.desc)
        blur_ms, blur_n, blur_bytes = bctx.timing_read("blur")
        small_ms, small_n, small_bytes = bctx.timing_read("blur_small")
        bctx.close()
        del batch_t
    last = step(0)

    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cpu" if share else "cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    if rank == 0:
        value = world * n_pairs / dt
        achieved = (blur_bytes / blur_n) / (blur_ms / blur_n * 1e-3) / 1e9 if blur_n else 0.0
        all_n, all_ms, all_bytes = blur_n + small_n, blur_ms + small_ms, blur_bytes + small_bytes
        achieved_all = all_bytes / (all_ms * 1e-3) / 1e9 if all_n else 0.0
        out = {
            "metric": "image_pairs_per_sec_end_to_end", "value": round(value, 3), "unit": "pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "single 1920x1080 pair, HessianAffine+RootSIFT, 1 synth iteration (BASELINE configs[1])",
                       "pairs_per_step": pps, "image": "1920x1080",
                       "overlap": "serial" if pipe is None else "%d gpu workers x %d pairs per batch + %d verify workers" % (args.gpu_workers, args.pairs_per_batch, args.verify_workers), "matcher": "linear (exact), FGINN 0.8",
                       "verification": "LO-RANSAC homography, Sampson, th 4 px", "parallelism": "pairs sharded, %d rank(s)" % world,
                       "keypoints_per_image": list(last.n_described), "tentatives": last.n_tentatives,
                       "inliers_last_pair": last.n_inliers, "mean_inliers": round(inl / n_pairs, 1),
                       "ransac_samples_last_pair": last.ransac_samples, "ransac_lo_last_pair": last.ransac_lo,
                       "stage_ms_per_pair": {"detect_describe": round(stage_ms[0] / n_pairs, 3), "match": round(stage_ms[1] / n_pairs, 3),
                                             "duplicates": round(stage_ms[2] / n_pairs, 3), "ransac": round(stage_ms[3] / n_pairs, 3)}},
            # the dominant kernel of the pyramid: the 32-row-tile instantiation of the blur (octaves 0-1 at this batching: 95 % of
            # the pyramid's bytes, 3/4 of its time); the launches of the smaller planes use the 16-row instantiation and are
            # launch-size bound: "all_blur_launches" is the figure over both
            "roofline": {"kernel": "gauss_blur_fast_kernel<R,32,2>", "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                         # HBM bytes per launch from the PMC passes committed in profiles/r01_pmc_blur_traffic.csv
                         # (FETCH_SIZE doubled per the gfx950 note + WRITE_SIZE, mean over the blur launches of the default batching,
                         # 16 images per launch; other batchings were not measured)
                         "traffic": 177224280 if (pipe is not None and args.pairs_per_batch == 8 and args.config == "c2") else None,
                         "launches": blur_n, "mean_launch_us": round(blur_ms / max(blur_n, 1) * 1e3, 3),
                         "algorithmic_bytes_per_launch": round(blur_bytes / max(blur_n, 1), 1),
                         "all_blur_launches": {"achieved": round(achieved_all, 2), "frac": round(achieved_all / HBM_PEAK_GBS, 4),
                                               "launches": all_n, "mean_launch_us": round(all_ms / max(all_n, 1) * 1e3, 3),
                                               "algorithmic_bytes_per_launch": round(all_bytes / max(all_n, 1), 1)}},
        }
        if not args.no_cpu_baseline and world == 1:
            a, b, _ = pairs_host[0]
            v, secs, ninl = cpu_baseline(a, b, 12345)
            out["cpu_baseline"] = {"value": round(v, 5), "unit": "pairs/s", "cores": 1, "kind": "port",
                                   "sample": "1 of the benchmark's 1920x1080 pairs through the CPU oracle (oracle/), %.1f s, %d inliers" % (secs, ninl)}
        print(json.dumps(out))
    if pipe is not None:
        pipe.close()
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
