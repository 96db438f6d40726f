#!/usr/bin/env python3
"""Headline benchmark: image pairs / second, end to end (detect + describe + match + RANSAC) on
synthetic 1920x1080 pairs, HessianAffine + RootSIFT, one identity view per image
(BASELINE.json configs[1]).

  python bench.py --gpus N --steps K --warmup W

A step = one batch of --pairs-per-step (default 256) image pairs through the whole hot path, SURVEY.md 8d's boundary:
two decoded 8-bit grey images in (pinned) host memory -> upload -> detect, describe, match, duplicate filter, LO-RANSAC ->
inlier set + H on the host (mods.cpp:184-383).  --input hbm keeps the images resident in HBM instead (fp32).  The reported
value is pairs / second.  N > 1 is launched by torch.distributed.run, one rank per GPU; pairs are independent units, so every
rank works on its own pairs (weak scaling, no data-path collective) and the reported value is all pairs / max-over-ranks time.

Rank 0 prints one JSON line.  Besides the contract fields it carries
  roofline       - the Gaussian-blur kernel of the Hessian pyramid (largest share of the HBM-bound pyramid time): algorithmic
                   bytes (8 B/px per blur launch) / mean launch time from HIP events recorded on the workers'
                   streams DURING the timed steps (the kernels of the other workers run beside it); "isolated" inside it is
                   the same launch sequence on one stream with nothing else on the GPU (a separate short leg)
  roofline_match - the brute-force descriptor search at the size of BASELINE configs[4]: 2*N*M*128 integer operations / the
                   time of the match stage (HIP events), against the dense int8 MFMA peak
  cpu_baseline   - the CPU oracle (oracle/) timed on this host on one pair of the same workload: OpenMP inside the stages at two
                   team sizes (the faster one is the value), and the reference's own task structure (2 images side by side,
                   mods.cpp:234-251)
  per_keypoint   - SURVEY 8d's informational figures: keypoints / s and bilinear taps / s of the Baumberg iteration, the orientation
                   patches and the measurement-region extraction, regions / s of SIFT (isolated leg, HIP-event scopes per stage)
  harder_verification - the same pipeline on pairs where only 40 % of image 2 follows the homography (tens of RANSAC samples per
                   pair instead of 3): a short driver-timed leg behind the timed steps
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


def pmc_blur_traffic():
    """HBM bytes per launch of the dominant blur instantiation from the newest committed PMC passes (profiles/rNN_pmc_blur_traffic.csv:
    FETCH_SIZE doubled per the gfx950 note + WRITE_SIZE, separate rocprofv3 --pmc runs, at the default batching: 32 images per launch)."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_blur_traffic.csv")), reverse=True):
        try:
            for line in open(path):
                if line.startswith("# launches of gauss_blur_fast_kernel<R; 32; 2> only:"):
                    return int(line.split(":")[1].split()[0])
        except OSError:
            pass
    return None


def cpu_baseline(img1, img2, seed):
    """The CPU oracle on one pair of the workload, three ways.  Returns the cpu_baseline object of the JSON line."""
    import orc
    import pipeline_oracle as po
    import refdeg
    # threads of the OpenMP leg: the cores this process may actually use (affinity mask, cgroup quota), at most 64 (one socket:
    # the loops are short, more threads only add fork/join cost).  A box whose quota is invisible here (or that is busy) makes a
    # 64-thread run SLOWER than one core, so smaller teams are timed as well and the fastest one is reported with its size.
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            ncpu = min(ncpu, max(1, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    ncpu = max(1, min(ncpu, 64))

    def verify(ra, rb, tc):
        un = orc.duplicate_filter(tc, ra, rb, 2.0, 1)
        if refdeg.available():       # the reference's own degensac (oracle/_ref), single thread as in the reference
            return po.loransac_h(po.u6_of(ra, rb, un), po.laf_of(ra, rb, un), seed_time=seed)[2]
        return len(un)               # GPU box without oracle/_ref: RANSAC is < 1 % of the CPU time

    def chain(inner_threads, image_tasks):
        orc.lib().orc_set_threads(inner_threads)
        t0 = time.time()
        (ra, _), (rb, _) = po.pmap(orc.detect_describe, (img1, img2), threads=image_tasks)
        tc = orc.match_fginn(ra, rb, 0.8)
        ninl = verify(ra, rb, tc)
        orc.lib().orc_set_threads(1)
        return time.time() - t0, ninl

    # OpenMP over rows / keypoints / queries inside every stage.  Two team sizes are timed (the loops are short: on a box whose cores
    # are shared or whose quota is invisible a large team is SLOWER than a small one) and the FASTER one is the reported value, with
    # its size as `cores`; then the reference's own task structure (the two images side by side, everything else serial).
    teams = sorted({min(ncpu, 16), min(ncpu, 8)}, reverse=True)
    timed = []
    ninl = 0
    for n_try in teams:
        t_try, ninl = chain(n_try, 1)
        timed.append((t_try, n_try))
    t_best, n_best = min(timed)
    t_ref, _ = chain(1, 2)                # the reference's structure: the two images as two tasks, the rest serial
    return {"value": round(1.0 / t_best, 5), "unit": "pairs/s", "cores": n_best, "kind": "port",
            "sample": "1 of the benchmark's 1920x1080 pairs through the CPU oracle (oracle/; RANSAC = the reference's degensac "
                      "when oracle/_ref is present), %d inliers: %s with OpenMP over rows / keypoints / queries (the fastest team is "
                      "the value), %.1f s in the reference's task structure (2 images side by side, mods.cpp:234-251); %d usable cores"
                      % (ninl, ", ".join("%.1f s on %d cores" % (t, n) for t, n in timed), t_ref, ncpu),
            "teams": [{"value": round(1.0 / t, 5), "cores": n} for t, n in timed],
            "cores_usable": ncpu,
            "reference_task_structure": {"value": round(1.0 / t_ref, 5), "cores": 2}}


def thread_cpu_ns():
    """CPU nanoseconds of every thread of this process, keyed name/tid (/proc/self/task/*/schedstat)."""
    out = {}
    try:
        for tid in os.listdir("/proc/self/task"):
            try:
                with open("/proc/self/task/%s/comm" % tid) as f:
                    name = f.read().strip()
                with open("/proc/self/task/%s/schedstat" % tid) as f:
                    ns = int(f.read().split()[0])
            except (OSError, ValueError, IndexError):
                continue
            out[name + "/" + tid] = ns
    except OSError:
        pass
    return out


def host_share(torch, device, local_rank, local_world):
    """Cores of this rank: usable cores (affinity, cgroup quota) / ranks of the node, from the GPU's NUMA node when known.
    Pins the process (threads created later inherit the mask).  Returns the "host" object of the JSON line."""
    try:
        usable = sorted(os.sched_getaffinity(0))
    except AttributeError:
        usable = list(range(os.cpu_count() or 1))
    quota = None
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = max(1, int(float(q) / float(period)))
    except (OSError, ValueError):
        pass
    n_usable = min(len(usable), quota) if quota else len(usable)
    per_rank = max(1, n_usable // max(1, local_world))
    pinned, numa = False, None
    if local_world > 1:
        cand = usable
        try:   # cores next to the GPU: /sys/bus/pci/devices/<domain:bus:device.0>/local_cpulist
            pr = torch.cuda.get_device_properties(device)
            bdf = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
            numa = int(open("/sys/bus/pci/devices/%s/numa_node" % bdf).read())
            local = set()
            for part in open("/sys/bus/pci/devices/%s/local_cpulist" % bdf).read().strip().split(","):
                lo, _, hi = part.partition("-")
                local.update(range(int(lo), int(hi or lo) + 1))
            near = [c for c in usable if c in local]
            if len(near) >= per_rank:
                cand = near
        except (OSError, ValueError, AttributeError, RuntimeError):
            pass
        if quota and quota < len(usable):
            # the limit is a CPU-time quota, not a set of cores: a slice of per_rank named cores would queue a rank's dozen threads
            # on two CPUs while 200 others idle; stay on the cores next to the GPU (or wherever the mask allows)
            mine = cand
        else:
            # the ranks that share `cand` take consecutive slices of it
            k = local_rank % max(1, len(cand) // per_rank)
            mine = cand[k * per_rank:(k + 1) * per_rank] or cand[:per_rank]
        try:
            os.sched_setaffinity(0, mine)
            pinned = True
        except (OSError, AttributeError):
            pass
    return {"cores_visible": os.cpu_count(), "cores_usable": n_usable, "cgroup_quota": quota, "cores_per_rank": per_rank,
            "pinned": pinned, "gpu_numa_node": numa}


MFMA_I8_PEAK_TOPS = 5000.0   # dense int8 = 2 x the bf16 rate (MI355X_MICROARCH.md: "I8 ... ~2x bf16 rate"; no spec line of its own)
MFMA_I8_MEASURED_TOPS = 3944.0   # the guide's only measured int8 figure (16x16x64 micro-benchmark ceiling, ">= 3944 TOPS")


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
        pps = max(1, args.pairs_per_step or 48)
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
        if args.scene == "two_planes":     # a scene with parallax: no single homography explains it, DEGENSAC's ordinary branch
            a, b, _, _, _ = synth.pair_two_planes(w, h, seed=5000, blobs=30000)
        else:                              # one plane: every good 7-point sample is H-degenerate (plane-and-parallax search)
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
        serial_rate, serial_ms = args.steps / dt, dt / args.steps * 1e3
        # the pair pipeline with the F verification: DEGENSAC of pair i (host: LO fits, the degenerate branch) runs under detect /
        # describe / match of the pairs behind it, as it does for H in the headline configuration
        gw, vw = min(args.gpu_workers, 2), min(args.verify_workers, 6)
        pipe = pkg.Pipeline(0, w, h, params, gw, vw, 1)

        def run(n):
            res, pending = [], 0
            for i in range(n):
                if pending >= pipe.capacity - 1:
                    res.append(pipe.next()[0]); pending -= 1
                pipe.submit(t.data_ptr(), i); pending += 1
            while pending:
                res.append(pipe.next()[0]); pending -= 1
            return res
        pps = max(1, min(args.pairs_per_step or 8, 8))
        run(max(1, args.warmup) * pps)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        pres = run(args.steps * pps)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        pipe.close()
        same = all(r.n_inliers == res[-1].n_inliers and r.n_tentatives == res[-1].n_tentatives for r in pres)
        args_steps_c5 = args.steps
        args.steps = len(pres)          # the rate below is pairs of the pipelined pass per second
        ctx.timing_enable(["match"]); ctx.timing_reset()
        for _ in range(4):
            last = pkg.match_pair_dev(ctx, t.data_ptr(), w, h, params)[0]
        mms, mn, _ = ctx.timing_read("match")
        mms /= 4.0
        # one pass over the N x M x 128 contraction (pass 1; pass 2 now runs on the few undecided queries only and is not counted)
        flops = 2.0 * last.n_described[0] * last.n_described[1] * 128
        ach = flops / (mms * 1e-3) / 1e12 if mms else 0.0
        out.update(value=round(args.steps / dt, 4), ms_per_step=round(dt / args_steps_c5 * 1e3, 3), steps=args_steps_c5,
                   config={"workload": "4096x4096 pair, exact FGINN match + DEGENSAC F (BASELINE configs[4]); scene: " + args.scene,
                           "pairs_per_step": pps, "overlap": "%d gpu workers + %d verify workers, one pair per batch" % (gw, vw),
                           "one_pair_at_a_time": {"value": round(serial_rate, 3), "ms_per_pair": round(serial_ms, 3)},
                           "pipeline_results_equal_serial": bool(same),
                           "keypoints_per_image": list(last.n_described), "tentatives": last.n_tentatives, "inliers": last.n_inliers,
                           "ransac_samples": last.ransac_samples, "ransac_lo": last.ransac_lo,
                           "stage_ms": {"detect_describe": round(last.ms_detect_describe, 2), "match": round(last.ms_match, 2),
                                        "duplicates": round(last.ms_duplicates, 2), "ransac": round(last.ms_ransac, 2)}},
                   roofline={"kernel": "match stage (pack + match_nn1_kernel + mid + match_fginn_kernel + emit, i8 MFMA; ops of pass 1 only)", "bound": "mfma", "achieved": round(ach, 2),
                             "peak": MFMA_I8_PEAK_TOPS, "unit": "TOP/s", "frac": round(ach / MFMA_I8_PEAK_TOPS, 4), "traffic": None,
                             "stage_ms": round(mms, 3)})
        ctx.close()
    print(json.dumps(out))


def config_c3(args):
    """BASELINE configs[2]: the view-synthesis ladder of iters_MODS.ini on ONE hard 1920x1080 pair.  --gpus 1: one context (the views
    of a step side by side on its view workers).  --gpus N (launched by torch.distributed.run, one rank per GPU): the views of every
    step sharded over the ranks largest first, ONE all-gather of the described regions per step over RCCL, the search split by query
    rows, verification on rank 0 (mods-light-zmq_amd/shard.py); rank 0 then times the same ladder through the C++ form of
    that decomposition (mods_match_ladder_multi: one process, one host thread + one communicator rank per device) while the other
    ranks wait.  --min-matches (default: never reached) keeps the step loop from stopping early, so that every step of the file runs."""
    import numpy as np
    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge
    import synth
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    share = os.environ.get("MODS_BENCH_SHARE_GPU") == "1"
    if world > 1:
        import datetime
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        pg_timeout = datetime.timedelta(seconds=int(os.environ.get("MODS_BENCH_PG_TIMEOUT", "300")))
        if share:
            dist.init_process_group("gloo", timeout=pg_timeout)
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=pg_timeout)
    device = local_rank if (world > 1 and not share) else 0
    torch.cuda.set_device(device)
    pkg = ge.load_package()
    if pkg.lib().mods_device_count() <= 0:
        raise SystemExit("bench.py needs an MI355X: libmodsgpu has no CPU path")
    pkg.ransac_pin_seed(12345)
    out = {"metric": "image_pairs_per_sec_end_to_end", "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic"}
    w, h = W, H
    a, b = _hard_pair(synth, np, w, h, 3000)
    t = torch.from_numpy(np.stack([a, b])).cuda(device)
    d = pkg.view_ctx_dims(w, h)
    ctx = pkg.Context(device, d[0], d[1], 2)      # two image slots: a view of both images in one chain of launches
    steps = pkg.iters_mods_steps()
    mm = args.min_matches
    if world > 1:
        import importlib
        shard = importlib.import_module("mods_light_zmq_amd.shard")
        what = "iters_MODS.ini HessianAffine steps, views sharded over %d ranks (one all-gather of the regions per step)" % world

        def run_once():
            return shard.match_ladder_distributed(pkg, ctx, t.data_ptr(), w, h, steps, dist, device, min_matches=mm, seed_time=12345)
    elif args.ladder == "hessian":     # the two HessianAffine sections only
        rep1, rep2 = pkg.ImgRep(ctx, 1 << 20), pkg.ImgRep(ctx, 1 << 20)

        def run_once():
            return pkg.match_ladder_dev(ctx, t.data_ptr(), w, h, steps, rep1, rep2, min_matches=mm)[0]
        what = "iters_MODS.ini HessianAffine steps"
    else:                              # the file as it is: [MSER0], [MSER1], [HessianAffine2], [HessianAffine3]
        rep1, rep2 = pkg.ImgRep(ctx, 1 << 20), pkg.ImgRep(ctx, 1 << 20)
        L = pkg.LadderStep.make
        det_steps = [[None, None, steps[0], steps[1]],
                     [L((1,), 360.0, scales=(1, 0.25, 0.125), init_sigma=0.8, fginn=0.85, half_orientation=1),
                      L((1, 3, 6), 360.0, scales=(1, 0.25), init_sigma=0.8, fginn=0.8, half_orientation=1), None, None]]
        dets = [pkg.HessAffParams.default(), pkg.HessAffParams.mser()]
        repm1, repm2 = pkg.ImgRep(ctx, 1 << 20), pkg.ImgRep(ctx, 1 << 20)

        def run_once():
            return pkg.match_ladder_dets_dev(ctx, t.data_ptr(), w, h, det_steps, dets, [rep1, repm1], [rep2, repm2], min_matches=mm)[0]
        what = "iters_MODS.ini, all four steps (MSER, MSER, HessianAffine, HessianAffine)"
    for _ in range(args.warmup):
        run_once()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res = [run_once() for _ in range(args.steps)]
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cpu" if share else "cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    r = res[-1]
    get = (lambda k: r[k]) if isinstance(r, dict) else (lambda k: getattr(r, k))
    cfg = {"workload": what + " on one hard 1920x1080 pair (tilt 6), BASELINE configs[2]", "min_matches": mm,
           "view_workers": int(os.environ.get("MODS_LADDER_WORKERS", "4")),
           "steps_done": get("steps_done"), "views": get("n_views"), "regions": list(get("n_described")), "tentatives": get("n_tentatives"),
           "unique": get("n_unique"), "inliers": get("n_inliers"),
           "parallelism": "views sharded, %d rank(s)" % world}
    if not isinstance(r, dict):
        cfg["ransac_samples"] = r.ransac_samples
        cfg["stage_ms"] = {"synth_detect_describe": round(r.ms_detect_describe, 2), "match": round(r.ms_match, 2),
                           "duplicates": round(r.ms_duplicates, 2), "ransac": round(r.ms_ransac, 2)}
    if world > 1 and rank == 0:
        # the sharded result against the one-GPU ladder of the same pair (same pinned seed): identical banks, tentatives and inliers
        rep1, rep2 = pkg.ImgRep(ctx, 1 << 20), pkg.ImgRep(ctx, 1 << 20)
        pkg.ransac_pin_seed(12345)
        one = pkg.match_ladder_dev(ctx, t.data_ptr(), w, h, steps, rep1, rep2, min_matches=mm)[0]
        cfg["equals_one_gpu_ladder"] = bool(list(one.n_described) == list(get("n_described")) and one.n_tentatives == get("n_tentatives")
                                            and one.n_unique == get("n_unique") and one.n_inliers == get("n_inliers"))
        rep1.close(); rep2.close()
    if world > 1:
        dist.barrier()
    if world > 1 and not share:
        # the same decomposition as ONE process (C++ host: a thread and a communicator rank per device); rank 0 runs it, the others wait
        multi = None
        if rank == 0:
            try:
                mg = pkg.Multi(list(range(world)), w, h)
                for _ in range(max(1, args.warmup)):
                    mg.match_ladder(a, b, steps, min_matches=mm)
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    mr, _ = mg.match_ladder(a, b, steps, min_matches=mm)
                dm = (time.perf_counter() - t1) / args.steps
                multi = {"what": "mods_match_ladder_multi: one process, one host thread + one RCCL rank per device (images from host memory)",
                         "ms_per_pair": round(dm * 1e3, 3), "value": round(1.0 / dm, 4), "uses_rccl": bool(mg.uses_rccl),
                         "steps_done": mr.steps_done, "views": mr.n_views, "inliers": mr.n_inliers,
                         "same_inliers_as_ranks": bool(mr.n_inliers == get("n_inliers"))}
                mg.close()
            except Exception as e:      # reported, not fatal: the ranks' figure above stands on its own
                multi = {"error": str(e)[:200]}
        dist.barrier()
        cfg["in_process_multi"] = multi
    out.update(value=round(args.steps / dt, 4), ms_per_step=round(dt / args.steps * 1e3, 3), config=cfg)
    if rank == 0:
        print(json.dumps(out))
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--pairs", type=int, default=6, help="distinct synthetic pairs cycled through the steps")
    ap.add_argument("--pairs-per-step", type=int, default=None,
                    help="image pairs in the batch that one step processes (default 256 for the headline configuration: 20 steps are\n"
                         "~5000 pairs = a timed region of several seconds; 48 for c4, 8 for c5)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gpu-workers", type=int, default=4, help="pipeline threads running detect/describe/match (one context each; round 6, tools/sweep_pipeline.sh: 4 x 16 pairs per batch 994 pairs/s, 6 x 8 928)")
    ap.add_argument("--verify-workers", type=int, default=8, help="pipeline threads running duplicate filter + LO-RANSAC")
    ap.add_argument("--pairs-per-batch", type=int, default=16, help="pairs a GPU worker pushes through detect/describe as one batch of launches (at most 16: one grouped match launch)")
    ap.add_argument("--serial", action="store_true", help="no cross-pair overlap: one mods_match_pair_dev call per step (images in HBM)")
    ap.add_argument("--input", default="host_u8", choices=["host_u8", "host_f32", "hbm"],
                    help="where a pair lives when its step starts: 8-bit grey in pinned host memory (default: the boundary of the "
                         "reference's step loop), fp32 in pinned host memory, or fp32 resident in HBM")
    ap.add_argument("--no-match-leg", action="store_true", help="skip the configs[4]-sized match measurement (roofline_match)")
    ap.add_argument("--no-harder-leg", action="store_true", help="skip the short leg on pairs with 40 % inliers (harder_verification)")
    ap.add_argument("--scene", default="planar", choices=["planar", "two_planes"],
                    help="--config c5: one plane (SURVEY 8d's generator: every sample is H-degenerate) or two planes with parallax")
    ap.add_argument("--ladder", default="full", choices=["full", "hessian"],
                    help="--config c3: the whole iters_MODS.ini (MSER steps 0-1, HessianAffine steps 2-3) or its HessianAffine steps only")
    ap.add_argument("--min-matches", type=int, default=1 << 30,
                    help="--config c3: [Iterations] minMatches of the step loop (default: never reached, every step of the ladder runs; "
                         "15 = the reference's file, where the loop stops after the first step that verifies 15 matches)")
    ap.add_argument("--keep-workers", action="store_true", help="N > 1: do not shrink the worker counts to the rank's share of the host cores")
    ap.add_argument("--inlier-ratio", type=float, default=0.0,
                    help="0 (default): SURVEY 8d's pairs (one homography, ~94 %% of the tentatives are inliers: 3 RANSAC samples); "
                         "0 < R < 1: only the left R of image 2 follows the homography, the rest a second motion, so the "
                         "verification runs hundreds of samples (a robustness figure, not the headline)")
    ap.add_argument("--config", default="c2", choices=["c2", "c3", "c4", "c5"],
                    help="BASELINE.json configs[]: c2 = the headline 1080p pair (default, what the driver runs); c3 = view-synthesis "
                         "ladder on a hard 1080p pair; c4 = 1-MP pairs (throughput); c5 = 4096x4096 pair with DEGENSAC F verification")
    args = ap.parse_args()
    if args.config == "c3":
        return config_c3(args)
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
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    # MODS_BENCH_SHARE_GPU=1 (development aid for 1-GPU boxes): every rank uses GPU 0 and the ranks meet over gloo, so that the
    # multi-rank control flow can be exercised without a multi-GPU node
    share = os.environ.get("MODS_BENCH_SHARE_GPU") == "1"
    if world > 1:
        import datetime
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # a rank that never arrives makes the others fail after this long instead of hanging the node
        pg_timeout = datetime.timedelta(seconds=int(os.environ.get("MODS_BENCH_PG_TIMEOUT", "300")))
        if share:
            dist.init_process_group("gloo", timeout=pg_timeout)
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=pg_timeout)
    device = local_rank if (world > 1 and not share) else 0
    torch.cuda.set_device(device)

    pkg = ge.load_package()
    if pkg.lib().mods_device_count() <= 0:
        raise SystemExit("bench.py needs an MI355X: libmodsgpu has no CPU path")
    if args.serial:
        args.input = "hbm"
    # host side of a rank: the cores this process may use (affinity mask, cgroup quota) are split between the ranks of the node,
    # each rank takes its share from the cores of its GPU's NUMA node when sysfs tells them (the pipeline's worker threads inherit
    # the mask), and the number of verify threads follows the share
    host = host_share(torch, device, local_rank, local_world if world > 1 else 1)
    if world > 1 and not args.keep_workers:
        # the pipeline's threads sleep while they wait (MODS_SYNC), so the GPU workers cost a tenth of a core each and stay; the
        # verify threads do the host's real work (~2 ms of CPU per pair): two per core of the rank's share
        cpr = host["cores_per_rank"]
        if cpr < 2:
            args.gpu_workers = min(args.gpu_workers, 3)
        args.verify_workers = max(3, min(args.verify_workers, 2 * cpr))

    # synthetic inputs: seed = 1000*config + pair index (config 2 = the 1080p pair), distinct per rank.  The generator makes
    # 8-bit valued images ("uint8 then float32"), so the 8-bit and the fp32 form of a pair are the same image.
    if args.inlier_ratio > 0:
        pairs_host = [synth.pair_partial(W, H, seed=2000 + rank * 100 + i, frac=args.inlier_ratio) for i in range(args.pairs)]
    else:
        pairs_host = [synth.pair(W, H, seed=2000 + rank * 100 + i) for i in range(args.pairs)]
    stacks = [np.stack([a, b]) for a, b, _ in pairs_host]
    pairs_dev = [torch.from_numpy(x).cuda(device) for x in stacks]
    pinned = []
    if args.input != "hbm":
        dt_in = np.uint8 if args.input == "host_u8" else np.float32
        for x in stacks:
            buf = pkg.PinnedBuffer(x.shape, dt_in)
            buf.array[...] = x.astype(dt_in)
            pinned.append(buf)
    torch.cuda.synchronize()

    ctx = pkg.Context(device, W, H, 2)
    # the context of the isolated legs (rank 0, behind the timed steps) is created BEFORE the pipeline: the runtime spreads the streams
    # of a process over a few hardware queues in the order they appear, and a context whose two streams - created after the
    # pipeline's twenty - end up on one queue runs its side stream behind its main stream (1.8 instead of 0.95 ms per scale space)
    bctx = None
    if rank == 0 and not args.serial:
        bctx = pkg.Context(device, W, H, 2 * max(1, args.pairs_per_batch))
        bctx.detect_describe_dev(torch.cat([pairs_dev[0]] * max(1, args.pairs_per_batch), dim=0).contiguous().data_ptr(), 2 * max(1, args.pairs_per_batch), W, H)
    pkg.lib().mods_ransac_set_device(device)
    params = pkg.PairParams.default()
    pkg.ransac_pin_seed(12345)
    pipe = None if args.serial else pkg.Pipeline(device, W, H, params, args.gpu_workers, args.verify_workers, args.pairs_per_batch)

    def step(i):
        res, _ = pkg.match_pair_dev(ctx, pairs_dev[i % len(pairs_dev)].data_ptr(), W, H, params)
        return res

    def submit(i):
        k = i % len(pairs_dev)
        if args.input == "hbm":
            pipe.submit(pairs_dev[k].data_ptr(), i)
        else:
            pipe.submit_host(pinned[k].ptr.value, i, u8=(args.input == "host_u8"))

    def run(n_steps):
        """n_steps pairs through the hot path; returns the per-pair results in step order."""
        if pipe is None:
            return [step(i) for i in range(n_steps)]
        out, pending = [], 0
        for i in range(n_steps):
            if pending >= pipe.capacity - 1:
                out.append(pipe.next()[0]); pending -= 1
            submit(i); pending += 1
        while pending:
            out.append(pipe.next()[0]); pending -= 1
        return out

    pps = max(1, args.pairs_per_step or 256)
    run(args.warmup * pps)
    # (no per-launch timers inside the timed region: the workers replay their detect + describe chain as a hipGraph, which stage timers
    # switch off, and an event pair per blur launch is not part of the product path; the in-pipeline figure of the blur kernel comes
    # from a short instrumented leg right behind the timed steps)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    if pipe is not None:
        pipe.cpu_seconds(reset=True)
    cpu0 = time.process_time()
    thr0 = thread_cpu_ns()
    t0 = time.perf_counter()
    replays0 = pipe.graph_replays() if pipe is not None else 0
    results = run(args.steps * pps)
    torch.cuda.synchronize()
    replays = (pipe.graph_replays() - replays0) if pipe is not None else 0
    cpu_s = time.process_time() - cpu0          # every thread of this rank: submit loop, GPU workers, verify workers, RANSAC task pool
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    worker_cpu = pipe.cpu_seconds() if pipe is not None else (0.0, 0.0)
    thr1 = thread_cpu_ns()
    per_tid = {k: thr1[k] - thr0.get(k, 0) for k in thr1}
    by_name = {}
    for k, v in per_tid.items():
        nm = k.split("/")[0]
        tot, cnt, top = by_name.get(nm, (0, 0, 0))
        by_name[nm] = (tot + v, cnt + (1 if v > 0 else 0), max(top, v))
    by_thread = sorted(by_name.items(), key=lambda kv: -kv[1][0])
    # instrumented leg: HIP events around every blur launch, on the streams the launches go to (the workers' streams), while the
    # pipeline runs as in the timed steps (eager launches: the timers switch the graph replay off)
    timed = pipe if pipe is not None else ctx
    timed.timing_enable(["blur", "blur_small"])
    if pipe is None:
        ctx.timing_reset()
    run(min(2, args.steps) * pps)
    torch.cuda.synchronize()
    blur_ms, blur_n, blur_bytes = timed.timing_read("blur")
    small_ms, small_n, small_bytes = timed.timing_read("blur_small")
    timed.timing_enable([])
    n_pairs = len(results)
    inl = sum(r.n_inliers for r in results)
    stage_ms = [sum(getattr(r, f) for r in results) for f in ("ms_detect_describe", "ms_match", "ms_duplicates", "ms_ransac")]

    # isolated leg: what a GPU worker launches for one batch (the images of pairs_per_batch pairs in one detect/describe pass),
    # on ONE stream with nothing else on the GPU; also yields the regions for the match leg
    nb = 1 if pipe is None else max(1, args.pairs_per_batch)
    iso = None
    match_leg = None
    if rank == 0:
        n_img = 2 * min(nb, len(pairs_dev)) if nb > 1 else 2
        reps = (2 * nb + n_img - 1) // n_img
        batch_t = torch.cat([pairs_dev[i % len(pairs_dev)] for i in range(nb)], dim=0).contiguous()
        if bctx is None:
            bctx = pkg.Context(device, W, H, 2 * nb)
        torch.cuda.synchronize()
        for _ in range(2):
            bctx.detect_describe_dev(batch_t.data_ptr(), 2 * nb, W, H, params.det, params.desc)
        pyr_stages = ["blur", "blur_small", "response", "resize", "nms", "pyramid"]
        bctx.pyramid_streams(1)           # per-launch scopes: every launch alone on the GPU
        bctx.timing_enable(pyr_stages); bctx.timing_reset()
        for _ in range(6):
            bctx.detect_describe_dev(batch_t.data_ptr(), 2 * nb, W, H, params.det, params.desc)
        i_ms, i_n, i_bytes = bctx.timing_read("blur")
        is_ms, is_n, is_bytes = bctx.timing_read("blur_small")
        pyr_ms = {st: bctx.timing_read(st)[0] / 6.0 for st in pyr_stages}
        # the whole scale space as ONE scope, in a pass of its own: with the per-stage scopes on, every launch is bracketed by two
        # more event records and the octaves' side stream is timed apart
        pyr_ms["pyramid_with_stage_scopes"] = pyr_ms["pyramid"]
        for n_streams, key in ((1, "pyramid_one_stream"), (2, "pyramid")):
            bctx.pyramid_streams(n_streams)
            bctx.timing_enable(["pyramid"]); bctx.timing_reset()
            for _ in range(6):
                bctx.detect_describe_dev(batch_t.data_ptr(), 2 * nb, W, H, params.det, params.desc)
            pyr_ms[key] = bctx.timing_read("pyramid")[0] / 6.0
        # SURVEY 8d's informational per-keypoint figures: HIP-event scopes around the per-keypoint stages of the same batch
        kp_stages = ["baumberg", "orient", "extract", "sift"]
        bctx.pyramid_streams(2)
        bctx.baumberg_stats_enable(True)
        bctx.timing_enable(kp_stages); bctx.timing_reset()
        kp_reps = 4
        for _ in range(kp_reps):
            nd_b, nr_b = bctx.detect_describe_dev(batch_t.data_ptr(), 2 * nb, W, H, params.det, params.desc)
        kp_ms = {st: bctx.timing_read(st)[0] / kp_reps for st in kp_stages}
        b_kp = b_it = 0
        for i in range(2 * nb):
            a_, b_ = bctx.baumberg_stats(i)
            b_kp += a_; b_it += b_
        b_kp /= kp_reps; b_it /= kp_reps
        bctx.baumberg_stats_enable(False)
        smm = int(params.det.smmWindowSize)
        ps_d, mr_d = int(params.desc.desc_patchSize), float(params.desc.desc_mrSize)
        ex_taps = 0
        for i in range(2 * nb):
            sc = np.asarray(bctx.regions_fetch(i)["s"], dtype=np.float64)
            P = 2 * np.ceil(sc * mr_d).astype(np.int64) + 1           # DescribeRegions, synth-detection.hpp:170-263
            blurred = (P.astype(np.float64) / ps_d) > 0.4
            ex_taps += int((np.where(blurred, (P + 2) ** 2, 0) + ps_d * ps_d).sum())
        per_kp = {"what": "one batch of %d images on one stream, nothing else on the GPU; HIP-event scope per stage" % (2 * nb),
                  "baumberg": {"ref": "affine.cpp:26-158", "keypoints": int(b_kp), "iterations": int(b_it), "taps_per_iteration": smm * smm,
                               "ms": round(kp_ms["baumberg"], 4),
                               "kpts_per_s": round(b_kp / (kp_ms["baumberg"] * 1e-3), 0) if kp_ms["baumberg"] else None,
                               "taps_per_s": round(b_it * smm * smm / (kp_ms["baumberg"] * 1e-3), 0) if kp_ms["baumberg"] else None},
                  "orient": {"ref": "synth-detection.cpp:836-929", "keypoints": int(sum(nd_b)), "taps_per_keypoint": int(params.desc.ori_patchSize) ** 2,
                             "ms": round(kp_ms["orient"], 4),
                             "kpts_per_s": round(sum(nd_b) / (kp_ms["orient"] * 1e-3), 0) if kp_ms["orient"] else None},
                  "extract": {"ref": "synth-detection.hpp:170-263", "regions": int(sum(nr_b)), "taps": ex_taps,
                              "taps_model": "(P + 2)^2 window samples (imageToPatchScale > 0.4) + patchSize^2 resampled pixels per region",
                              "ms": round(kp_ms["extract"], 4),
                              "regions_per_s": round(sum(nr_b) / (kp_ms["extract"] * 1e-3), 0) if kp_ms["extract"] else None,
                              "taps_per_s": round(ex_taps / (kp_ms["extract"] * 1e-3), 0) if kp_ms["extract"] else None},
                  "sift": {"ref": "siftdesc.cpp:346-400", "regions": int(sum(nr_b)), "ms": round(kp_ms["sift"], 4),
                           "regions_per_s": round(sum(nr_b) / (kp_ms["sift"] * 1e-3), 0) if kp_ms["sift"] else None}}
        bctx.timing_enable([])
        iso = (i_ms, i_n, i_bytes, is_ms, is_n, is_bytes)
        del reps
        if not args.no_match_leg:
            # BASELINE configs[4]-sized search (~50 k descriptors per side, SURVEY 8d: N x M x 128): the regions of the first
            # images of the benchmark's pairs against the regions of their second images
            rep_q, rep_t = pkg.ImgRep(bctx, 1 << 17), pkg.ImgRep(bctx, 1 << 17)
            k = 0
            while k < len(pairs_dev):          # distinct images only (a repeated image would add exact duplicates)
                p = k
                bctx.detect_describe_dev(pairs_dev[p].data_ptr(), 2, W, H, params.det, params.desc)
                rep_q.append_ctx(0); rep_t.append_ctx(1)
                k += 1
            nq, nt = len(rep_q), len(rep_t)
            pkg.match_reps(bctx, rep_q, rep_t)                      # warm-up
            reps_m = 5
            bctx.timing_enable(["match"]); bctx.timing_reset()
            for _ in range(reps_m):
                tent, _, _ = pkg.match_reps(bctx, rep_q, rep_t)
            m_ms, m_n, _ = bctx.timing_read("match")
            # the matrix-core kernel alone (match_nn1_kernel), in a pass of its own so that its event pair is not inside the stage's figure
            bctx.timing_enable(["match_nn1"]); bctx.timing_reset()
            for _ in range(reps_m):
                pkg.match_reps(bctx, rep_q, rep_t)
            k_ms, k_n, _ = bctx.timing_read("match_nn1")
            bctx.timing_enable([])
            match_leg = (nq, nt, m_ms / reps_m, len(tent), k_ms / max(k_n, 1), k_n // reps_m)
            rep_q.close(); rep_t.close()
        bctx.close()
        del batch_t
    last = step(0)
    # BASELINE configs[1] is "Single 1920x1080 pair": the latency of ONE pair with nothing else in flight, two ways - one
    # mods_match_pair_dev call (images fp32 in HBM, every stage on the caller's context, the runtime's own wait), and one pair through
    # the pipeline from 8-bit images in pinned host memory (its threads sleep between looks at their streams: +0.05-0.1 ms per stage)
    latency = None
    if rank == 0:
        import statistics
        ts = []
        for i in range(12):
            t1 = time.perf_counter(); step(i); ts.append((time.perf_counter() - t1) * 1e3)
        latency = {"one_call_hbm_f32": {"median": round(statistics.median(ts[2:]), 3), "min": round(min(ts[2:]), 3), "calls": len(ts) - 2}}
        if pipe is not None:
            tp = []
            for i in range(12):
                t1 = time.perf_counter(); submit(i); pipe.next(); tp.append((time.perf_counter() - t1) * 1e3)
            latency["pipeline_one_in_flight_" + args.input] = {"median": round(statistics.median(tp[2:]), 3), "min": round(min(tp[2:]), 3), "calls": len(tp) - 2}

    # a harder verification, driver-timed: only the left 40 % of image 2 follows the homography (the rest a second motion), so LO-RANSAC
    # draws tens of samples per pair instead of 3 - the same pipeline, a short leg (SURVEY 8d's generator stays the headline)
    harder = None
    if rank == 0 and pipe is not None and args.inlier_ratio == 0 and not args.no_harder_leg:
        hp = [synth.pair_partial(W, H, seed=2900 + i, frac=0.4) for i in range(3)]
        hbuf = []
        for a_, b_, _ in hp:
            buf = pkg.PinnedBuffer((2, H, W), np.uint8)
            buf.array[...] = np.stack([a_, b_]).astype(np.uint8)
            hbuf.append(buf)

        def run_h(n):
            out, pending = [], 0
            for i in range(n):
                if pending >= pipe.capacity - 1:
                    out.append(pipe.next()[0]); pending -= 1
                pipe.submit_host(hbuf[i % len(hbuf)].ptr.value, i, u8=True); pending += 1
            while pending:
                out.append(pipe.next()[0]); pending -= 1
            return out
        run_h(64)
        torch.cuda.synchronize(); th0 = time.perf_counter()
        hres = run_h(3 * pps)
        torch.cuda.synchronize(); th = time.perf_counter() - th0
        harder = {"what": "3 x %d pairs whose image 2 follows the homography on its left 40 %% only (synth.pair_partial), same pipeline" % pps,
                  "value": round(len(hres) / th, 3), "unit": "pairs/s",
                  "mean_ransac_samples": round(sum(r.ransac_samples for r in hres) / len(hres), 1),
                  "mean_inliers": round(sum(r.n_inliers for r in hres) / len(hres), 1),
                  "mean_tentatives": round(sum(r.n_tentatives for r in hres) / len(hres), 1)}
        for b_ in hbuf:
            b_.close()

    rank_rates = [n_pairs / dt]
    if world > 1:
        mine = torch.tensor([dt], dtype=torch.float64, device="cpu" if share else "cuda")
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        rank_rates = [round(n_pairs / float(t.item()), 2) for t in every]     # a slow rank (host cores, a busy GPU) shows here
        tmax = mine.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    if rank == 0:
        value = world * n_pairs / dt

        def gbs(by, ms):
            return by / (ms * 1e-3) / 1e9 if ms else 0.0
        achieved = gbs(blur_bytes, blur_ms)
        i_ms, i_n, i_bytes, is_ms, is_n, is_bytes = iso
        traffic = pmc_blur_traffic() if (args.config == "c2" and pipe is not None and args.pairs_per_batch == 16 and args.inlier_ratio == 0) else None
        out = {
            "metric": "image_pairs_per_sec_end_to_end", "value": round(value, 3), "unit": "pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "latency_ms_single_pair": latency,
            "host": host, "pairs_per_s_by_rank": rank_rates,
            # what a pair costs the HOST (rank 0, timed steps): CPU seconds of the whole process per pair (every thread), and of the
            # pipeline's own worker threads inside their stages; process_cpu_s_per_pair x pairs/s = busy cores per rank
            "host_cpu": {"process_cpu_ms_per_pair": round(cpu_s / n_pairs * 1e3, 3), "gpu_workers_cpu_ms_per_pair": round(worker_cpu[0] / n_pairs * 1e3, 3),
                         "verify_workers_cpu_ms_per_pair": round(worker_cpu[1] / n_pairs * 1e3, 3), "busy_cores": round(cpu_s / dt, 2),
                         "by_thread_name_ms_per_pair": {k: {"total": round(v[0] * 1e-6 / n_pairs, 3), "threads": v[1], "busiest": round(v[2] * 1e-6 / n_pairs, 3)}
                                                        for k, v in by_thread[:6] if v[0] > 0}},
            "config": {"workload": "single 1920x1080 pair, HessianAffine+RootSIFT, 1 synth iteration (BASELINE configs[1])"
                                   + (" - VARIANT: second motion over %.0f %% of image 2 (--inlier-ratio)" % (100 * (1 - args.inlier_ratio)) if args.inlier_ratio > 0 else ""),
                       "pairs_per_step": pps, "image": "1920x1080",
                       "input": {"host_u8": "two 8-bit grey images in pinned host memory per pair, uploaded inside the timed region",
                                 "host_f32": "two fp32 grey images in pinned host memory per pair, uploaded inside the timed region",
                                 "hbm": "fp32 images resident in HBM"}[args.input],
                       "output": "inlier set + H on the host",
                       "overlap": "serial" if pipe is None else "%d gpu workers x %d pairs per batch + %d verify workers" % (args.gpu_workers, args.pairs_per_batch, args.verify_workers), "matcher": "linear (exact), FGINN 0.8",
                       "streams_per_gpu_worker": int(os.environ.get("MODS_PIPELINE_STREAMS", "1")), "detect_describe_batches_replayed_as_graph": replays,
                       "verification": "LO-RANSAC homography, Sampson, th 4 px", "parallelism": "pairs sharded, %d rank(s)" % world,
                       "keypoints_per_image": list(last.n_described), "tentatives": last.n_tentatives,
                       "inliers_last_pair": last.n_inliers, "mean_inliers": round(inl / n_pairs, 1),
                       "ransac_samples_last_pair": last.ransac_samples, "ransac_lo_last_pair": last.ransac_lo,
                       "stage_ms_per_pair": {"detect_describe": round(stage_ms[0] / n_pairs, 3), "match": round(stage_ms[1] / n_pairs, 3),
                                             "duplicates": round(stage_ms[2] / n_pairs, 3), "ransac": round(stage_ms[3] / n_pairs, 3)}},
            # the dominant kernel of the pyramid: the 32-row-tile instantiation of the fused blur + Hessian-response kernel (octaves
            # 0-1 at this batching: 95 % of the pyramid's bytes), measured during the timed steps; algorithmic bytes by SURVEY 8d's
            # unfused model: 8 B/px for the blur + 8 B/px for the response of every level it produces.  The launches of the smaller
            # planes use the 16-row instantiation and are launch-size bound: "all_blur_launches" is the figure over both
            "roofline": {"kernel": "gauss_blur_fast_kernel<R,32,2,true> (blur + Hessian response)", "bound": "hbm",
                         # `achieved` / `frac`: the kernel's own figure - its launches for one batch on one stream with nothing else on the
                         # GPU (HIP events, a separate leg right after the timed steps; details under "isolated").  *_in_pipeline: the
                         # same launches while the pipeline runs, where six contexts share the GPU and a launch gets a fraction of the
                         # bandwidth - a contention figure, not the kernel's
                         "achieved": round(gbs(i_bytes, i_ms), 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs(i_bytes, i_ms) / HBM_PEAK_GBS, 4),
                         "achieved_in_pipeline": round(achieved, 2), "frac_in_pipeline": round(achieved / HBM_PEAK_GBS, 4),
                         # a launch is priced by SURVEY 8d's UNFUSED model (blur 8 B/px + response 8 B/px = 16 B/px), as 8d asks; the
                         # fused kernel's own algorithmic bytes are 12 B/px (reads 4, writes 8): *_fused_model
                         "bytes_model": "SURVEY 8d unfused: 16 B/px per level",
                         "achieved_fused_model": round(gbs(i_bytes, i_ms) * 0.75, 2), "frac_fused_model": round(gbs(i_bytes, i_ms) * 0.75 / HBM_PEAK_GBS, 4),
                         # HBM bytes per launch from the PMC passes committed under profiles/ (FETCH_SIZE doubled per the gfx950
                         # note + WRITE_SIZE, mean over the blur launches of the default batching, 32 images per launch)
                         "traffic": traffic,
                         # what the counters say the kernel moves per second against the 8 TB/s peak (traffic / mean launch time):
                         # the kernel's REAL HBM rate, beside the model figure above
                         "hbm_counter_frac": round(traffic / (i_ms / max(i_n, 1) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if (traffic and i_ms) else None,
                         "measured": "HIP events around every launch, on the stream it is launched on",
                         "launches": i_n, "mean_launch_us": round(i_ms / max(i_n, 1) * 1e3, 3),
                         "algorithmic_bytes_per_launch": round(i_bytes / max(i_n, 1), 1),
                         "in_pipeline": {"what": "the same launches on the workers' streams while the pipeline runs (six contexts share the GPU; a short instrumented leg behind the timed steps)",
                                         "launches": blur_n, "mean_launch_us": round(blur_ms / max(blur_n, 1) * 1e3, 3),
                                         "all_blur_launches": {"achieved": round(gbs(blur_bytes + small_bytes, blur_ms + small_ms), 2),
                                                               "frac": round(gbs(blur_bytes + small_bytes, blur_ms + small_ms) / HBM_PEAK_GBS, 4),
                                                               "launches": blur_n + small_n}},
                         "isolated": {"what": "one batch's launches on one stream, nothing else on the GPU (separate leg after the timed steps)",
                                      "achieved": round(gbs(i_bytes, i_ms), 2), "frac": round(gbs(i_bytes, i_ms) / HBM_PEAK_GBS, 4),
                                      "achieved_fused_model": round(gbs(i_bytes, i_ms) * 0.75, 2), "frac_fused_model": round(gbs(i_bytes, i_ms) * 0.75 / HBM_PEAK_GBS, 4),
                                      "launches": i_n, "mean_launch_us": round(i_ms / max(i_n, 1) * 1e3, 3),
                                      "all_blur_launches": {"achieved": round(gbs(i_bytes + is_bytes, i_ms + is_ms), 2),
                                                            "frac": round(gbs(i_bytes + is_bytes, i_ms + is_ms) / HBM_PEAK_GBS, 4),
                                                            "launches": i_n + is_n}}},
        }
        if iso is not None:
            # the scale space as a whole by SURVEY 8d's model (97 B/px over all octaves + 8 B/px initial blur = 284.77 MB per
            # 1080p image): the same isolated leg, HIP-event scopes per stage on one stream.  "kernels" adds up the launches'
            # own scopes (blur of the large and of the small planes, response, decimation, NMS + compaction; the octaves from the third
            # on run on a side stream, so these scopes overlap in time); "one_scope" is a single scope from the first blur launch to
            # the end of the compaction, measured in a pass without the per-stage scopes
            n_images = 2 * nb
            sum_p = 0
            cw, ch = W, H
            while ch > 12 and cw > 12:      # the octave ladder of pyramid.cpp:520-528 (border 5), cvRound halving
                sum_p += cw * ch
                cw, ch = int(round(cw * 0.5)), int(round(ch * 0.5))     # Python rounds half to even, as cvRound does
            pyr_bytes = n_images * (97.0 * sum_p + 8.0 * W * H)
            k_ms = sum(pyr_ms[st] for st in ("blur", "blur_small", "response", "resize", "nms"))
            out["roofline_pyramid"] = {"what": "scale space of a %d-image batch of one context: every blur / response / decimation launch of every octave + NMS + "
                                               "compaction (isolated leg)" % n_images,
                                       "bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS, "bytes_model": "SURVEY 8d: 97 B/px over all octaves + 8 B/px initial blur",
                                       "algorithmic_bytes": pyr_bytes, "stage_ms": {st: round(v, 4) for st, v in pyr_ms.items()},
                                       "kernels": {"ms": round(k_ms, 4), "achieved": round(gbs(pyr_bytes, k_ms), 2), "frac": round(gbs(pyr_bytes, k_ms) / HBM_PEAK_GBS, 4)},
                                       "large_planes_only": {"what": "blur (32-row tiles) + response + decimation + NMS: the four scopes round 2 was judged on",
                                                             "ms": round(k_ms - pyr_ms["blur_small"], 4),
                                                             "frac": round(gbs(pyr_bytes, k_ms - pyr_ms["blur_small"]) / HBM_PEAK_GBS, 4)},
                                       "one_scope_one_stream": {"ms": round(pyr_ms["pyramid_one_stream"], 4), "frac": round(gbs(pyr_bytes, pyr_ms["pyramid_one_stream"]) / HBM_PEAK_GBS, 4)},
                                       "one_scope": {"what": "as shipped: the octaves from the third on (and their NMS) on a side stream", "ms": round(pyr_ms["pyramid"], 4), "achieved": round(gbs(pyr_bytes, pyr_ms["pyramid"]), 2),
                                                     "frac": round(gbs(pyr_bytes, pyr_ms["pyramid"]) / HBM_PEAK_GBS, 4)}}
        if match_leg:
            nq, nt, mms, ntent, kms, klaunches = match_leg
            ops = 2.0 * nq * nt * 128
            ach = ops / (mms * 1e-3) / 1e12
            kach = ops / (kms * klaunches * 1e-3) / 1e12 if kms else 0.0
            out["roofline_match"] = {"kernel": "match stage (pack + match_nn1_kernel + fix + mid + match_fginn_kernel + emit; i8 MFMA)", "bound": "mfma",
                                     "achieved": round(ach, 2), "peak": MFMA_I8_PEAK_TOPS, "unit": "TOP/s",
                                     "frac": round(ach / MFMA_I8_PEAK_TOPS, 4), "traffic": None,
                                     # the same against the guide's only MEASURED int8 figure (micro-benchmark ceiling >= 3944 TOPS)
                                     "peak_measured": MFMA_I8_MEASURED_TOPS, "frac_of_measured_peak": round(ach / MFMA_I8_MEASURED_TOPS, 4),
                                     "kernel_frac_of_measured_peak": round(kach / MFMA_I8_MEASURED_TOPS, 4),
                                     # the matrix-core kernel on its own (HIP events around match_nn1_kernel only): what the MFMA pipe does
                                     "kernel_frac": round(kach / MFMA_I8_PEAK_TOPS, 4), "kernel_achieved": round(kach, 2),
                                     "kernel_us": round(kms * 1e3, 2), "kernel_launches_per_search": klaunches,
                                     "queries": nq, "trains": nt, "ops": ops, "stage_ms": round(mms, 4), "tentatives": ntent,
                                     "measured": "HIP events around the match stage, BASELINE configs[4]-sized lists built from the "
                                                 "benchmark's regions (separate leg after the timed steps)"}
        if iso is not None:
            out["per_keypoint"] = per_kp
        if harder is not None:
            out["harder_verification"] = harder
        if not args.no_cpu_baseline and world == 1:
            a, b, _ = pairs_host[0]
            out["cpu_baseline"] = cpu_baseline(a, b, 12345)
        print(json.dumps(out))
    if pipe is not None:
        pipe.close()
    ctx.close()
    for b in pinned:
        b.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
