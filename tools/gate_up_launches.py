"""One CSV row per TIMED gate_up launch of a rocprofv3 kernel trace of `python bench.py` (profiles/r03_gate_up_launches.csv): lets
`roofline.frac` be recomputed by hand as sum(2 * real_tokens * 3072 * 16384) / sum(duration) / 2.5e15.  The gate_up projection of a step is
the k_gemm_nt_256<.., 6 (SwiGLU), ..> launch (+ a k_gemm_nt<.., 6, ..> launch on the last rows when the 256-tile grid's last round is at most
half full): 31 per step (the 32nd layer runs on the 8 last rows only and is not timed by bench.py either)."""
import csv, json, re, sys
trace, bench_json = sys.argv[1], sys.argv[2]
line = [l for l in open(bench_json) if l.startswith("{")][-1]
b = json.loads(line)
tokens = b["config"]["real_tokens_per_timed_step"]
steps = b["steps"]
rows = sorted(csv.DictReader(open(trace)), key=lambda r: int(r["Start_Timestamp"]))
gu = [r for r in rows if re.search(r"k_gemm_nt(_256)?<(true|1), ?6[,>]", r["Kernel_Name"].replace("(anonymous namespace)::", ""))]
main = [r for r in gu if "k_gemm_nt_256" in r["Kernel_Name"]]
per_step = 31
n_steps_total = len(main) // per_step
first_timed = n_steps_total - steps
print("step,memory_step,launch_in_step,real_tokens,grid_workgroups,start_ns,duration_ns,tail128_duration_ns")
tot_ns = tot_fl = 0.0
for s in range(first_timed, n_steps_total):
    tk = tokens[s - first_timed]
    for j in range(per_step):
        r = main[s * per_step + j]
        t0, t1 = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        # a SwiGLU 128-tile launch right behind it (before the next 256 launch) belongs to the same projection
        nxt = int(main[s * per_step + j + 1]["Start_Timestamp"]) if s * per_step + j + 1 < len(main) else 1 << 62
        tail = [x for x in gu if "k_gemm_nt_256" not in x["Kernel_Name"] and t1 <= int(x["Start_Timestamp"]) < nxt]
        td = sum(int(x["End_Timestamp"]) - int(x["Start_Timestamp"]) for x in tail[:1])
        grid = r.get("Grid_Size_X") or r.get("Grid_Size") or ""
        wg = r.get("Workgroup_Size_X") or r.get("Workgroup_Size") or "512"
        try:
            nwg = int(grid) // int(wg)
        except ValueError:
            nwg = ""
        print(f"{s - first_timed},{b['config']['memory_steps_timed'][0] + s - first_timed},{j},{tk},{nwg},{t0},{t1 - t0},{td}")
        tot_ns += (t1 - t0) + td
        tot_fl += 2.0 * tk * 3072 * 16384
print(f"# launches {steps * per_step}; sum FLOP {tot_fl:.6e}; sum duration {tot_ns:.0f} ns; achieved {tot_fl / tot_ns / 1e3:.1f} TFLOP/s = {tot_fl / tot_ns / 1e3 / 2500:.4f} of 2.5 PF "
      f"(PROFILED run; bench.py reports the un-profiled HIP-event figure.  The two differ by box and by clock: round 4's profiled run was the SLOWER one -- "
      f"0.507 here against 0.542 from HIP events on another box, PMC shader clock 1.74-1.76 GHz -- so no direction is claimed; compare runs of ONE gpurun call)")
