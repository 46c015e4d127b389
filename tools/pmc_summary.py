"""Condense rocprofv3 --pmc passes of bench.py into profiles/<tag>_pmc_traffic.json and a markdown table.

usage: python tools/pmc_summary.py <tag> <dir with *_counter_collection.csv> [<dir> ...]

Each directory holds one pass (FETCH_SIZE, WRITE_SIZE or SQ counters), collected separately as
MI355X_MICROARCH.md prescribes.  FETCH_SIZE / WRITE_SIZE are reported in units of 1 KiB; on gfx950 FETCH_SIZE
counts 128-byte requests as 64 bytes, so the read side is doubled (same guide, HBM section).
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name.replace(" ", "")


def main():
    tag, dirs = sys.argv[1], sys.argv[2:]
    vals = defaultdict(lambda: defaultdict(list))        # kernel -> counter -> [per-dispatch values]
    grids = defaultdict(lambda: defaultdict(dict))       # kernel -> dispatch-key -> counters (per grid size)
    for d in dirs:
        for path in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
            with open(path, newline="") as f:
                for row in csv.DictReader(f):
                    k = short(row["Kernel_Name"])
                    if k.startswith("__amd") or "at::" in k or "elementwise" in k:
                        continue
                    v = float(row["Counter_Value"])
                    vals[k][row["Counter_Name"]].append(v)
                    grids[k][int(row["Grid_Size"])].setdefault(row["Counter_Name"], []).append(v)
    out, lines = {}, []
    lines.append("| kernel | launches | HBM read MB (FETCH_SIZE x2) | HBM write MB | total MB / launch | MFMA busy cycles per MFMA | LDS bank-conflict cycles |")
    lines.append("|---|---|---|---|---|---|---|")
    for k in sorted(vals):
        c = vals[k]
        n = max(len(v) for v in c.values())
        mean = lambda name: (sum(c[name]) / len(c[name])) if c.get(name) else None
        rd = mean("FETCH_SIZE")
        wr = mean("WRITE_SIZE")
        rd_b = rd * 1024 * 2 if rd is not None else None
        wr_b = wr * 1024 if wr is not None else None
        busy, mops, mops16 = mean("SQ_VALU_MFMA_BUSY_CYCLES"), mean("SQ_INSTS_VALU_MFMA_MOPS_F32"), mean("SQ_INSTS_VALU_MFMA_MOPS_F16")
        per = None
        if busy and mops:
            # MOPS_* count 512-flop units: one 32x32x2 f32 MFMA = 4096 flops = 8 units... report busy per instruction
            per = busy / (mops / 8.0) if mops else None
        elif busy and mops16:
            per = busy / (mops16 / 64.0)      # one 32x32x16 f16 MFMA = 32768 flops = 64 units
        conf = mean("SQ_LDS_BANK_CONFLICT")
        if rd_b is not None and wr_b is not None:
            out[k] = {"launches": n, "hbm_read_bytes_per_launch": rd_b, "hbm_write_bytes_per_launch": wr_b,
                      "traffic_bytes_per_launch": rd_b + wr_b,
                      "by_grid": {str(g): {"hbm_read_bytes": 2048 * sum(cc["FETCH_SIZE"]) / len(cc["FETCH_SIZE"]),
                                           "hbm_write_bytes": 1024 * sum(cc["WRITE_SIZE"]) / len(cc["WRITE_SIZE"])}
                                  for g, cc in grids[k].items() if "FETCH_SIZE" in cc and "WRITE_SIZE" in cc}}
        f = lambda x, s=1e6: "-" if x is None else "%.1f" % (x / s)
        lines.append("| %s | %d | %s | %s | %s | %s | %s |" % (
            k, n, f(rd_b), f(wr_b), f((rd_b or 0) + (wr_b or 0)) if rd_b is not None else "-",
            "-" if per is None else "%.1f" % per, "-" if conf is None else "%.0f" % conf))
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles")
    with open(os.path.join(root, "%s_pmc_traffic.json" % tag), "w") as f:
        json.dump(out, f, indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
