#!/usr/bin/env python3
"""Digest of tools/pmc_summary.py output (several passes concatenated): one line per kernel with derived ratios.
  share of wave time: issuing any instruction / issuing VALU / parked in s_waitcnt or a barrier (WAIT_ANY) / stalled at issue
  (WAIT_INST_ANY, of which LDS); LDS bank-conflict cycles / LDS active cycles; L2 hit rate; VALU, LDS, VMEM instructions per wave.
usage: pmc_digest.py summary.txt"""
import collections
import sys

k = collections.defaultdict(dict)
name = None
for line in open(sys.argv[1]):
    if not line.startswith(" "):
        name = line.split(" dispatches")[0].strip()
        name = name[name.find("mods::") + 6:] if "mods::" in name else name
        continue
    parts = line.split()
    k[name][parts[0]] = float(parts[1])
print("%-44s %9s %8s %6s %6s %6s %6s %6s %7s %6s %8s %7s %7s" % ("kernel", "waves", "kcyc/wv", "issue", "valu", "wait", "stall", "st.lds", "ldsconf", "L2hit", "valu/wv", "lds/wv", "vmem/wv"))
for n, c in sorted(k.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    wc = c.get("SQ_WAVE_CYCLES", 0)
    if not wc:
        continue
    wv = max(c.get("SQ_WAVES", 0), 1)
    f = lambda key: c.get(key, 0) / wc
    conf = c.get("SQ_LDS_BANK_CONFLICT", 0) / c["SQ_LDS_IDX_ACTIVE"] if c.get("SQ_LDS_IDX_ACTIVE") else 0
    hit = c.get("TCC_HIT_sum", 0) / (c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0)) if c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0) else float("nan")
    print("%-44s %9d %8.1f %6.2f %6.2f %6.2f %6.2f %6.2f %7.3f %6.2f %8.0f %7.0f %7.0f" % (
        n[:44], wv, 4 * wc / wv / 1e3, f("SQ_ACTIVE_INST_ANY"), f("SQ_ACTIVE_INST_VALU"), f("SQ_WAIT_ANY"), f("SQ_WAIT_INST_ANY"), f("SQ_WAIT_INST_LDS"),
        conf, hit, c.get("SQ_INSTS_VALU", 0) / wv, c.get("SQ_INSTS_LDS", 0) / wv, c.get("SQ_INSTS_VMEM_RD", 0) / wv))
