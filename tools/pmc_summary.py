"""Summarise the three rocprofv3 --pmc passes written by tools/pmc_collect.sh for one kernel (substring match)."""
import csv, json, sys, collections
d, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "k_gemm_nt_256")
out = {}
dur = []
for p in ("sq", "fetch", "write"):
    rows = [r for r in csv.DictReader(open(f"{d}/{p}_counters.csv")) if pat in r["Kernel_Name"]]
    per = collections.defaultdict(list)
    for r in rows:
        per[r["Counter_Name"]].append(float(r["Counter_Value"]))
        if p == "sq" and r["Counter_Name"] == "SQ_WAVE_CYCLES":
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k, v in per.items():
        v = v[1:] if len(v) > 1 else v                      # drop the first (cold) launch
        out[k if k not in out else k + "_" + p] = sum(v) / len(v)
    if rows: out["kernel_name"] = rows[0]["Kernel_Name"][:90]
# kernel-trace durations from the (counter-free) trace of the same pass are inflated by counter collection; report both
out["duration_us_under_pmc"] = sum(dur[1:]) / max(1, len(dur) - 1) if dur else None
if "SQ_VALU_MFMA_BUSY_CYCLES" in out and "GRBM_GUI_ACTIVE" in out:
    # MFMA busy is summed over the 4 SIMDs x 256 CUs; GRBM_GUI_ACTIVE is summed over the 8 XCDs
    cyc = out["GRBM_GUI_ACTIVE"] / 8.0
    out["active_cycles_per_xcd"] = cyc
    out["mfma_util"] = out["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 256 * 4)
    if out["duration_us_under_pmc"]:
        out["clock_GHz_under_pmc"] = cyc / out["duration_us_under_pmc"] / 1e3
if "FETCH_SIZE" in out:
    out["fetch_bytes_x2_gfx950"] = out["FETCH_SIZE"] * 1024 * 2
if "WRITE_SIZE" in out:
    out["write_bytes"] = out["WRITE_SIZE"] * 1024
if "fetch_bytes_x2_gfx950" in out and "write_bytes" in out:
    out["hbm_bytes_per_launch"] = out["fetch_bytes_x2_gfx950"] + out["write_bytes"]
print(json.dumps(out, indent=1))
