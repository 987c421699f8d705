"""gpurun_out/r03/ (tools/profile_r03.sh) -> profiles/r03_*: the summaries that are judged.  The gate_up PMC json keeps its documented
schema (bench.py reads `rows`, `algorithmic_bytes`, `hbm_bytes_per_launch` from it)."""
import json, os, shutil
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "gpurun_out", "r03"), os.path.join(root, "profiles")
for a, b in (("kernel_summary.txt", "r03_kernel_summary.txt"), ("kernel_stats.csv", "r03_kernel_stats.csv"), ("gate_up_launches.csv", "r03_gate_up_launches.csv"),
             ("sq_counters.csv", "r03_pmc_gate_up_sq.csv"), ("fetch_counters.csv", "r03_pmc_gate_up_fetch.csv"), ("write_counters.csv", "r03_pmc_gate_up_write.csv"),
             ("pmc_attn_summary.txt", "r03_pmc_attn.txt"), ("render_kernel_summary.txt", "r03_render_kernel_summary.txt")):
    shutil.copy(os.path.join(src, a), os.path.join(dst, b))
new = json.load(open(os.path.join(src, "pmc_gate_up_summary.json")))
pj = os.path.join(dst, "r03_pmc_gate_up.json")
out = json.load(open(pj))
out.update({"FETCH_SIZE_KB": new["FETCH_SIZE"], "WRITE_SIZE_KB": new["WRITE_SIZE"], "fetch_bytes_x2_gfx950": new["fetch_bytes_x2_gfx950"],
            "write_bytes": new["write_bytes"], "hbm_bytes_per_launch": new["hbm_bytes_per_launch"], "GRBM_GUI_ACTIVE_sum_over_8_xcd": new["GRBM_GUI_ACTIVE"]})
for k in ("SQ_VALU_MFMA_BUSY_CYCLES", "mfma_util", "duration_us_under_pmc", "clock_GHz_under_pmc", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY",
          "SQ_ACTIVE_INST_ANY", "SQ_LDS_BANK_CONFLICT", "SQ_WAIT_INST_LDS"):
    out[k] = new[k]
assert {"rows", "algorithmic_bytes", "hbm_bytes_per_launch"} <= set(out)
json.dump(out, open(pj, "w"), indent=1)
print("profiles/ updated;", "gate_up mfma_util", round(out["mfma_util"], 3), "clock", round(out["clock_GHz_under_pmc"], 3))
