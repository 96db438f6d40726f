"""CPU test of the N > 1 host path: two gloo processes shard pairs / views and run the one exchange step
(ragged all-gather of descriptors) exactly as the RCCL path does on the GPU box."""
import os
import socket
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent('''
    import os, sys
    sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
    import numpy as np
    import torch.distributed as dist
    import __graft_entry__ as ge
    import orc, synth
    pkg = ge.load_package()
    import importlib.util
    spec = importlib.util.spec_from_file_location("shard", os.path.join(%r, "mods-light-zmq_amd", "shard.py"))
    shard = importlib.util.module_from_spec(spec); spec.loader.exec_module(shard)
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    # 1. throughput sharding: disjoint cover, results gathered in pair order
    ids = shard.shard_pairs(7, rank, world)
    vals = [10.0 * p + 1 for p in ids]
    full = shard.gather_pair_results(ids, vals, 7, dist)
    assert np.array_equal(full, 10.0 * np.arange(7) + 1), full
    # 2. one pair, views sharded: every rank describes its own "view" with the CPU oracle (no GPU here),
    #    one ragged all-gather gives every rank the whole train set in rank order
    img = synth.texture(160 + 40 * rank, 120, seed=5 + rank)
    regs, _ = orc.detect_describe(img)
    rec = np.zeros(len(regs), np.dtype([("desc", "u1", (128,)), ("xy", "f4", (2,)), ("view", "i4"), ("idx", "i4")]))
    rec["desc"] = regs["desc"]; rec["xy"][:, 0] = regs["x"]; rec["xy"][:, 1] = regs["y"]; rec["view"] = rank; rec["idx"] = np.arange(len(regs))
    allrec, counts = shard.allgather_ragged(rec, dist)
    assert sum(counts) == len(allrec) and counts[rank] == len(regs)
    off = sum(counts[:rank])
    assert np.array_equal(allrec[off:off + len(regs)], rec)
    assert np.array_equal(allrec["view"], np.repeat(np.arange(world), counts))
    # every rank now matches its own queries against the gathered train set: same answer as a single
    # process matching against the concatenation
    chk = np.zeros(1, np.int64); chk[0] = int(allrec["desc"].astype(np.int64).sum())
    import torch
    t = torch.from_numpy(chk.copy()); dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert int(t.item()) == int(chk[0])
    views = shard.largest_first_views([100, 12, 50, 50, 12, 25], world)
    assert sorted(sum(views, [])) == list(range(6)) and abs(sum([100, 12, 50, 50, 12, 25][i] for i in views[0]) - 124.5) <= 12.5
    # 3. the exchange step of the view-sharded ladder: every job computed by one rank, every rank ends up with all
    #    blocks in job order (counts differ, one padded all-gather)
    import torch
    n_jobs = 7
    owners = shard.largest_first_views([9, 1, 4, 4, 2, 7, 3], world)
    def block(j):
        k = (j * 5) %% 4 + (1 if j != 3 else 0)      # job 3 has no regions at all
        g = np.random.default_rng(100 + j).integers(0, 256, k * shard.REGION_BYTES, dtype=np.uint8)
        return torch.from_numpy(g)
    local = {j: block(j) for j in owners[rank]}
    blocks, counts = shard.exchange_blocks(local, n_jobs, dist)
    for j in range(n_jobs):
        assert torch.equal(blocks[j], block(j)), j
        assert counts[j] == block(j).numel() // shard.REGION_BYTES
    dist.barrier(); dist.destroy_process_group()
    print("rank", rank, "ok")
''') % (ROOT, ROOT, ROOT)


def test_two_rank_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\\n%s" % (r, o)
        assert "rank %d ok" % r in o


def test_multi_gpu_view_assignment_matches_the_python_rule(pkg):
    """mods_multi_assign (the C++ multi-GPU ladder's sharding of view jobs, csrc/multi.hip) = shard.largest_first_views:
    every job has one owner, loads are balanced, deterministic."""
    import numpy as np
    import importlib
    shard = importlib.import_module("mods_light_zmq_amd.shard")
    rng = np.random.default_rng(3)
    for n_dev in (1, 2, 3, 8):
        for n_jobs in (1, 5, 22, 62):
            areas = rng.uniform(1e4, 2e6, n_jobs).round()
            areas[::7] = areas[0]                                   # ties
            owner = pkg.multi_assign(areas, n_dev)
            want = shard.largest_first_views(list(areas), n_dev)
            for d, jobs in enumerate(want):
                assert sorted(jobs) == list(np.flatnonzero(owner == d)), (n_dev, n_jobs, d)
            loads = [areas[owner == d].sum() for d in range(n_dev)]
            assert max(loads) - min(loads) <= areas.max() + 1e-9      # LPT: no device is more than one job ahead
