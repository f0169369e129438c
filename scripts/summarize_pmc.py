"""Reduce the rocprofv3 --pmc CSVs of scripts/profile_pmc.sh to profiles/<tag>_pmc.json (+ pmc_latest.json).
HBM bytes per launch follow MI355X_MICROARCH.md section HBM: FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
counts wide coalesced reads at half their size, so the read side is doubled."""
import csv, glob, json, os, sys, collections
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "pmc")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
mode = sys.argv[2] if len(sys.argv) > 2 else "forward"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(src, f"{mode}_*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("nerf::", "")
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, cs in acc.items():
    d = {c: sum(v) / len(v) for c, v in cs.items()}
    d["launches_sampled"] = max(len(v) for v in cs.values())
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        d["hbm_bytes_per_launch"] = int((2 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024)
        d["hbm_read_bytes_corrected"] = int(2 * d["FETCH_SIZE"] * 1024)
        d["hbm_write_bytes"] = int(d["WRITE_SIZE"] * 1024)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "SQ_BUSY_CYCLES" in d and d["SQ_BUSY_CYCLES"]:
        d["mfma_busy_over_sq_busy"] = d["SQ_VALU_MFMA_BUSY_CYCLES"] / d["SQ_BUSY_CYCLES"]
    key = k
    if ("k_field_fwd_reg" in k or k.startswith("k_field_fwd<")) and "<true" not in k:  # the dominant kernel of bench.py's headline leg (inference instantiation)
        key = "k_field_fwd"
    out[key] = d
os.makedirs(os.path.join(root, "profiles"), exist_ok=True)
p = os.path.join(root, "profiles", f"{tag}_{mode}_pmc.json")
json.dump(out, open(p, "w"), indent=1, sort_keys=True)
# the summaries bench.py reads for `roofline.traffic` (one per leg)
latest = {"forward": "pmc_latest.json", "fwd_f32": "pmc_latest.json", "train_f32": "pmc_train_latest.json",
          "fwd_bf16": "pmc_bf16_fwd_latest.json", "train_bf16": "pmc_bf16_train_latest.json", "split_fwd": "pmc_split_fwd_latest.json", "train_split": "pmc_split_train_latest.json"}
if mode in latest:
    json.dump(out, open(os.path.join(root, "profiles", latest[mode]), "w"), indent=1, sort_keys=True)
for k in sorted(out):
    if "field" in k or k.startswith("k_dw"):
        print(k, json.dumps(out[k], sort_keys=True))
