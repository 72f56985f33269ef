"""Summarise the passes of tools/pmc_collect.sh into one JSON: per step kernel the rocprofv3 average duration, HBM bytes per launch
(FETCH_SIZE x 2 per the gfx950 note of MI355X_MICROARCH.md, WRITE_SIZE x 1), SQ wait / issue / MFMA shares, L2 hit rate.

    python tools/pmc_summary.py gpurun_out/pmc_<tag> profiles/<name>.json [model y_dim batch precision]
"""
import csv, glob, json, sys
from collections import defaultdict

KEYS = ("vae_rows2_kernel", "vae_rows_kernel", "wgrad4_kernel", "wgrad_lds_kernel", "wgrad_kernel", "apply_kernel", "slab_sum_kernel", "slab_reduce_kernel", "elbo")


def short(name):
    for k in KEYS:
        if k in name:
            return k
    return None


def counters(d):
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = short(row["Kernel_Name"])
            if k:
                acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return acc


def mean_tail(v):
    v = v[len(v) // 4:] if len(v) >= 8 else v
    return sum(v) / len(v) if v else None


def main():
    root, out = sys.argv[1], sys.argv[2]
    cfg = {}
    if len(sys.argv) >= 7:
        cfg = {"model": sys.argv[3], "y_dim": int(sys.argv[4]), "batch": int(sys.argv[5]), "precision": sys.argv[6]}
    res = {"config": cfg, "kernels": {},
           "method": "rocprofv3, one counter set per pass (tools/pmc_collect.sh); FETCH_SIZE doubled (gfx950 counts 64 B per 128-B request on "
                     "wide streaming reads), WRITE_SIZE as is, both in KB; SQ_* means per launch, WAVE_CYCLES / WAIT_* / ACTIVE_INST_* in quad-cycles"}
    dur = defaultdict(list)
    for f in glob.glob(root + "/stats/**/*kernel_trace.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = short(row["Kernel_Name"])
            if k:
                dur[k].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-3)
    sets = {n: counters(root + "/" + n) for n in ("fetch", "write", "sq", "l2", "inst")}
    for k, v in dur.items():
        e = {"launches": len(v), "avg_us": mean_tail(v), "min_us": min(v)}
        f, w = sets["fetch"].get(k, {}).get("FETCH_SIZE"), sets["write"].get(k, {}).get("WRITE_SIZE")
        if f and w:
            fa, wa = mean_tail(f), mean_tail(w)
            e.update({"FETCH_SIZE_raw_KB": fa, "WRITE_SIZE_raw_KB": wa, "read_bytes_corrected": 2 * fa * 1024, "write_bytes": wa * 1024,
                      "hbm_bytes_per_launch": 2 * fa * 1024 + wa * 1024})
        sq = {c: mean_tail(x) for c, x in sets["sq"].get(k, {}).items()}
        if sq.get("SQ_WAVE_CYCLES"):
            wc = sq["SQ_WAVE_CYCLES"]
            e["sq"] = sq
            e["sq_shares"] = {"parked_wait_any": sq.get("SQ_WAIT_ANY", 0) / wc, "issue_stall_wait_inst_any": sq.get("SQ_WAIT_INST_ANY", 0) / wc,
                              "issuing_active_inst_any": sq.get("SQ_ACTIVE_INST_ANY", 0) / wc,
                              "lds_conflict_of_lds_active": (sq.get("SQ_LDS_BANK_CONFLICT", 0) / sq["SQ_LDS_IDX_ACTIVE"]) if sq.get("SQ_LDS_IDX_ACTIVE") else None}
            if e.get("avg_us"):
                # MFMA busy cycles (sum over SIMDs) against 1024 SIMDs x kernel cycles at 2.1 GHz
                e["mfma_busy_frac_of_1024_simds"] = sq.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024 * e["avg_us"] * 2100.0)
        l2 = {c: mean_tail(x) for c, x in sets["l2"].get(k, {}).items()}
        if l2.get("TCC_HIT_sum") is not None and l2.get("TCC_MISS_sum") is not None and (l2["TCC_HIT_sum"] + l2["TCC_MISS_sum"]) > 0:
            e["l2"] = l2
            e["l2_hit_rate"] = l2["TCC_HIT_sum"] / (l2["TCC_HIT_sum"] + l2["TCC_MISS_sum"])
        inst = {c: mean_tail(x) for c, x in sets["inst"].get(k, {}).items()}
        if inst:
            e["inst"] = inst
        res["kernels"][k] = e
    json.dump(res, open(out, "w"), indent=1)
    for k, e in res["kernels"].items():
        print(k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in e.items() if a in ("avg_us", "hbm_bytes_per_launch", "l2_hit_rate", "mfma_busy_frac_of_1024_simds", "sq_shares")})


if __name__ == "__main__":
    main()
