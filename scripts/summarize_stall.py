"""Reduce the rocprofv3 --pmc CSVs of scripts/pmc_stall_diag.sh to one table per kernel: wave-cycle split (parked / issue-stalled / issuing),
MFMA-busy over kernel time, FIFO-full counters.  usage: summarize_stall.py DIR NAME NPASSES -> DIR/NAME_stall_summary.json (+ printed)."""
import collections
import csv
import glob
import json
import os
import sys

d, name, npass = sys.argv[1], sys.argv[2], int(sys.argv[3])
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for p in range(1, npass + 1):
    for f in glob.glob(os.path.join(d, "**", f"{name}_p{p}*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("nerf::", "")
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, cs in acc.items():
    c = {n: sum(v) / len(v) for n, v in cs.items()}
    row = dict(c)
    wc = c.get("SQ_WAVE_CYCLES")
    if wc:
        for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VMEM",
                  "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_MISC"):
            if n in c:
                row[n + "_over_wave_cycles"] = round(c[n] / wc, 4)
    if c.get("SQ_VALU_MFMA_BUSY_CYCLES") and c.get("GRBM_GUI_ACTIVE"):
        # MI355X: 1024 SIMDs, GRBM_GUI_ACTIVE summed over 8 XCDs (profiles/README.md)
        row["mfma_busy_over_kernel_time"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (c["GRBM_GUI_ACTIVE"] / 8.0), 4)
    out[k] = row
with open(os.path.join(d, f"{name}_stall_summary.json"), "w") as f:
    json.dump(out, f, indent=1, sort_keys=True)
for k in sorted(out):
    if any(s in k for s in ("field", "k_dw", "render_pair")):
        keep = {n: v for n, v in out[k].items() if n.endswith("_over_wave_cycles") or n in ("mfma_busy_over_kernel_time", "SQ_VMEM_WR_TA_DATA_FIFO_FULL",
                                                                                          "SQ_VMEM_TA_ADDR_FIFO_FULL", "SQ_VMEM_TA_CMD_FIFO_FULL", "SQ_LDS_BANK_CONFLICT")}
        print(k, json.dumps(keep, sort_keys=True))
