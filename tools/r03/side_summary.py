"""Per-kernel summary of tools/r03/collect_side.sh: mean duration (kernel trace), HBM bytes per launch from the FETCH_SIZE / WRITE_SIZE passes
(FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md; both counters are in KB), and -- MCEM chain kernel -- MFMA-busy share."""
import csv, glob, json, os, sys
from collections import defaultdict
O = sys.argv[1]

def rows(pattern):
    for f in glob.glob(os.path.join(O, pattern), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                yield r

def short(n):
    return n.split("(")[0].replace("void ", "").replace("dvae::", "")

out = {}
for what in ("stft", "mcem"):
    dur = defaultdict(list)
    for r in rows(f"{what}_stats/**/*kernel_trace.csv"):
        dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
    cnt = {"FETCH_SIZE": defaultdict(list), "WRITE_SIZE": defaultdict(list)}
    for tag, key in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        for r in rows(f"{what}_{tag}/**/*counter_collection.csv"):
            if r.get("Counter_Name") == key:
                cnt[key][short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    res = {}
    for k, v in dur.items():
        if not any(s in k for s in ("stft", "mcem", "mstep", "wiener", "target", "frames", "transpose")):
            continue
        e = {"launches": len(v), "avg_us": sum(v) / len(v), "max_us": max(v), "min_us": min(v)}
        if cnt["FETCH_SIZE"].get(k):
            f = cnt["FETCH_SIZE"][k]; e["fetch_MB_avg"] = 2 * 1024 * sum(f) / len(f) / 1e6; e["fetch_MB_max"] = 2 * 1024 * max(f) / 1e6
        if cnt["WRITE_SIZE"].get(k):
            w = cnt["WRITE_SIZE"][k]; e["write_MB_avg"] = 1024 * sum(w) / len(w) / 1e6; e["write_MB_max"] = 1024 * max(w) / 1e6
        res[k] = e
    out[what] = res
sq = defaultdict(lambda: defaultdict(float)); n = defaultdict(int)
for r in rows("mcem_sq/**/*counter_collection.csv"):
    k = short(r["Kernel_Name"]); sq[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, c in sq.items():
    if "mcem_mh" in k and c.get("SQ_BUSY_CYCLES"):
        out["mcem"].setdefault(k, {})["sq"] = {kk: vv for kk, vv in c.items()}
        wc = c.get("SQ_WAVE_CYCLES", 0.0)
        if wc:
            out["mcem"][k]["wave_share"] = {"parked": c.get("SQ_WAIT_ANY", 0) / wc, "issue_stalled": c.get("SQ_WAIT_INST_ANY", 0) / wc, "issuing": c.get("SQ_ACTIVE_INST_ANY", 0) / wc}
print(json.dumps(out, indent=1))
