#!/usr/bin/env python
"""Headline benchmark: nav steps/s, RGB-D observation -> action logits, batch 8 per GPU (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One "step" = one pass of the whole hot path (depth prep, CLIP ViT-L/14@336, frustum
delete, 3D-token update, agent-frame query, prefix MLPs, llava vision tower, Phi-3-mini prefill to the logits
of the first generated token) over a batch of 8 synthetic 224x224 posed RGB-D observations already resident
in HBM.  The memory is first advanced 8 untimed steps (the "warm" operating point of SURVEY.md 8d).
Prints ONE JSON line (rank 0)."""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_DENSE_TFLOPS = 2500.0     # /opt/skills/guides/MI355X_MICROARCH.md: ~2.5 PF dense bf16/fp16 MFMA
PEAK_HBM_GBS = 8000.0


def step_flops(cfg, lengths, n_img, pruned_last_layer=False):
    """Algorithmic FLOPs of one batch step (BASELINE.md section 3 formula, evaluated exactly for the configs and
    the real (unpadded, causal) sequence lengths)."""
    v, l = cfg.vit, cfg.llm
    L, W = v.tokens, v.width
    layer = 2 * L * W * (3 * W + W + 2 * v.mlp) + 4 * L * L * W
    embed = 2 * (L - 1) * 3 * v.patch * v.patch * W
    clip = v.layers * layer + embed + 2 * L * W * v.out_dim
    llava = (v.layers - 1) * layer + embed + 2 * (L - 1) * (W * v.proj_dim + v.proj_dim * v.proj_dim)
    qkv_tok = 2 * l.hidden * (l.heads + 2 * l.kv_heads) * l.head_dim
    tail_tok = 2 * (l.heads * l.head_dim * l.hidden + 3 * l.hidden * l.mlp)          # o_proj + MLP: row-wise, behind the attention
    per_tok = qkv_tok + tail_tok
    # the product evaluates the LAST layer's o_proj / MLP on each prompt's last row only (towers.py PRUNE_LAST_LAYER: only that row's
    # logits are read and everything behind the attention is row-wise) -- counted as executed, not as the reference executes it
    last_tail_rows = (lambda S: 1) if pruned_last_layer else (lambda S: S)
    phi = sum((l.layers - 1) * per_tok * S + qkv_tok * S + tail_tok * last_tail_rows(S) + l.layers * 2 * S * S * l.heads * l.head_dim
              + 2 * l.hidden * l.vocab for S in lengths)
    tok3d = n_img * 17e9        # SURVEY 8a row a7 (2-layer set encoder over 576+n tokens), informational
    return dict(clip=n_img * clip, llava=n_img * llava, phi3=phi, tokens3d=tok3d, total=n_img * (clip + llava) + phi + tok3d)


def _physical_cores():
    """Physical cores of the host (unique (socket, core) pairs of /proc/cpuinfo); logical CPUs if that cannot be read."""
    try:
        pairs, phys, core = set(), None, None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                core = line.split(":")[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    pairs.add((phys, core))
                phys = core = None
        if pairs:
            return len(pairs)
    except OSError:
        pass
    return os.cpu_count() or 1


def _best_threads(limit):
    """torch's fp32 GEMM does not always scale to every core of a large host: take the fastest of a few thread counts up to
    `limit` (1 s probe)."""
    best, best_t = min(8, limit), float("inf")
    a = torch.randn(2048, 2048)
    cand = sorted({n for n in (8, 16, 32, 64, 96, 128, 192, limit) if n <= limit})
    for n in cand:
        torch.set_num_threads(n)
        a @ a
        t0 = time.time()
        for _ in range(3):
            a @ a
        t = time.time() - t0
        if t < best_t:
            best, best_t = n, t
    return best


def cpu_baseline(cfg, seed, B, warm_steps, gpu_lengths=None):
    """The whole-step float32 oracle ("port", oracle/step_oracle.py) MEASURED on the host cores at the benchmark's operating point:
    B environments, memory advanced `warm_steps` steps (3D-memory oracle on seeded unit-norm grid features -- the towers do not
    touch the memory), then ONE full step timed end to end: both ViT-L/14@336 towers on B frames, the 3D-token builder, the prefix
    and the Phi-3-mini prefill over all 32 layers.  Plus the n = 8 threads point SURVEY.md 8d asks for, on a bounded sample."""
    import dataclasses
    from dynam3d_amd.policy import SyntheticTokenizer, synth_policy_weights
    from dynam3d_amd.synthetic import INSTRUCTION_64, SyntheticEpisodes
    from oracle.step_oracle import StepOracle
    phys = _physical_cores()
    n_threads = _best_threads(phys)
    torch.set_num_threads(n_threads)
    t_w = time.time()
    sd = synth_policy_weights(cfg, seed)                              # the full 4.4 B-parameter model, float32, CPU generator
    t_w = time.time() - t_w
    tok = SyntheticTokenizer(cfg.llm.vocab)
    orc = StepOracle(sd, cfg.vit, cfg.llm, B, tok)
    ep = SyntheticEpisodes(B, seed=seed)
    rng = np.random.default_rng(seed + 7)
    t_warm = time.time()
    for _ in range(warm_steps):
        fr = ep.next()
        grid = rng.standard_normal((B, 576, 768)).astype(np.float32)
        grid /= np.linalg.norm(grid, axis=-1, keepdims=True)
        orc.advance_memory(fr.depth, [p.tolist() for p in fr.positions], list(fr.headings), fr.patch_segm, grid)
    t_warm = time.time() - t_warm
    fr = ep.next()
    instr = [INSTRUCTION_64] * B
    t0 = time.time()
    orc.forward_logits(fr.rgb, fr.depth, instr, [p.tolist() for p in fr.positions], list(fr.headings), fr.patch_segm)
    measured = time.time() - t0
    st = {k: round(v, 3) for k, v in orc.timing.items()}
    out = dict(value=round(B / measured, 5), unit="env-steps/s", cores=n_threads, physical_cores=phys, logical_cpus=os.cpu_count(), kind="port",
               sample=("ONE full warm step measured end to end, %d environments, float32 oracle (oracle/step_oracle.py): CLIP ViT-L/14@336 + llava ViT-L on %d frames, "
                       "3D-token builder, prefix, Phi-3-mini prefill over all %d layers at S=%s (right-padded to %d); %d torch threads "
                       "(fastest of a 1 s GEMM probe over 8..%d = the physical cores).  Operating point: the SAME synthetic episodes and the SAME memory step as the "
                       "first step the GPU leg times (memory advanced %d steps); the untimed advance feeds the 3D memory seeded unit-norm grid features instead of "
                       "running CLIP on the host for every advance step (9 s each), so merge decisions -- and with them Ni/Nz and S -- differ from the GPU leg's: "
                       "S here sums to %d tokens, the GPU leg's first timed step to %s"
                       % (B, B, cfg.llm.layers, orc.last_lengths, max(orc.last_lengths), n_threads, phys, warm_steps, sum(orc.last_lengths),
                          sum(gpu_lengths) if gpu_lengths else "n/a")),
               memory_steps_advanced=warm_steps,
               seconds_measured=round(measured, 2), stages=st, seconds_weights=round(t_w, 1), seconds_memory_warmup=round(t_warm, 1))
    # n = 8 threads (comparability with the survey container's probe), bounded: one environment's frame through both towers,
    # the B-environment memory step, Phi-3 on 2 of the layers; the full-step figure is the sum scaled to B frames / all layers.
    try:
        torch.set_num_threads(8)
        sub = dataclasses.replace(cfg, llm=dataclasses.replace(cfg.llm, layers=2))
        o8 = StepOracle(sd, sub.vit, sub.llm, 1, tok)
        f1 = SyntheticEpisodes(1, seed=seed).next()
        o8.forward_logits(f1.rgb, f1.depth, [INSTRUCTION_64], [f1.positions[0].tolist()], list(f1.headings), f1.patch_segm)
        s8 = dict(o8.timing)
        est = B * (s8["vit_clip"] + s8["vit_llava"]) + st["tokens_3d"] * 0 + B * s8["tokens_3d"] + B * s8["phi3_prefill"] * (cfg.llm.layers / 2.0) * (sum(orc.last_lengths) / B / o8.last_lengths[0])
        out["n8"] = dict(threads=8, seconds_estimated_full_step=round(est, 1), value=round(B / est, 5),
                         sample="1 environment, cold, 2 of %d Phi-3 layers at S=%d; scaled to %d environments / all layers / the warm lengths" % (cfg.llm.layers, o8.last_lengths[0], B),
                         stages={k: round(v, 3) for k, v in s8.items()})
    except Exception as e:   # the primary measurement above stands on its own
        out["n8"] = {"error": repr(e)}
    return out


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_guard(gpus: int):
    """`--gpus N` must mean N ranks on N distinct GPUs, or no number at all.
      * N > 1 without a launcher (WORLD_SIZE unset): re-launch this very command under `torch.distributed.run` with N ranks -- if the
        box has N GPUs; otherwise exit non-zero (a plain `python bench.py --gpus 8` on one GPU used to print 8x one GPU's rate);
      * under a launcher whose WORLD_SIZE differs from --gpus: exit non-zero.
    Runs before anything is allocated."""
    world_env = os.environ.get("WORLD_SIZE")
    if world_env is not None:
        if int(world_env) != gpus:
            sys.exit(f"bench.py: --gpus {gpus} but the launcher started WORLD_SIZE={world_env} ranks; refusing to report a number for a job that is not the one named")
        return
    if gpus == 1:
        return
    n_dev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if os.environ.get("D3D_SHARE_DEVICE0") == "1":
        n_dev = max(n_dev, gpus if n_dev >= 1 else 0)          # test hook: several ranks on cuda:0 (see dynam3d_amd/dist.py)
    if n_dev < gpus:
        sys.exit(f"bench.py: --gpus {gpus} needs {gpus} GPUs, this machine shows {n_dev}; refusing to report a {gpus}-GPU number "
                 f"(launch: python -m torch.distributed.run --nnodes=1 --nproc-per-node {gpus} --master-addr 127.0.0.1 --master-port P bench.py --gpus {gpus} ...)")
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    print("bench.py: --gpus %d without a launcher -> %s" % (gpus, " ".join(cmd)), file=sys.stderr, flush=True)
    sys.exit(subprocess.call(cmd))


def device_identity(index: int) -> str:
    """Something that tells two GPUs of one node apart: the device UUID where torch exposes it, else PCI bus id, else the index."""
    try:
        p = torch.cuda.get_device_properties(index)
        for attr in ("uuid", "pci_bus_id"):
            v = getattr(p, attr, None)
            if v not in (None, ""):
                return f"{attr}:{v}"
    except Exception:       # noqa: BLE001
        pass
    return f"index:{index}"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--warm-steps", type=int, default=8, help="untimed trajectory steps before warmup (warm memory)")
    ap.add_argument("--cpu-baseline", default="auto", choices=["auto", "on", "off"])
    ap.add_argument("--hip-dense", default="all",
                    help="comma list of dense primitives on hand-written HIP kernels (linear,layer_norm,rms_norm,rope,swiglu,resize_normalize), 'all' or 'none'")
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    launch_guard(a.gpus)

    from dynam3d_amd import dense_ops as D
    from dynam3d_amd import dist as DD
    from dynam3d_amd.policy import Dynam3D_VLN, PolicyConfig, synth_policy_weights
    from dynam3d_amd.profiling import TIMER
    from dynam3d_amd.synthetic import INSTRUCTION_64, SyntheticEpisodes

    rank, local, world = DD.init_from_env()
    assert world == a.gpus, (world, a.gpus)                      # launch_guard() made sure of it
    torch.cuda.set_device(local)
    dev = f"cuda:{local}"
    # --hip-dense alone decides the dense backend: "all" (default) = every primitive on a hand-written kernel AND strict mode (a
    # PyTorch fallback raises); a subset / "none" = the A/B baselines kept under profiles/ (PyTorch-ROCm libraries for the rest).
    cfg = PolicyConfig(hip_dense=False)
    if a.hip_dense != "none":
        D.enable_hip_kernels(a.hip_dense.split(","))
    D.strict(a.hip_dense == "all")
    B = a.batch
    sd = synth_policy_weights(cfg, a.seed, device=dev)
    net = Dynam3D_VLN(cfg, sd, device=dev, batch_size=B, max_steps=a.warm_steps + a.warmup + a.steps + 2)
    del sd
    net.feature_fields.initialize_camera_setting(90.0, 90.0)
    ep = SyntheticEpisodes(B, seed=a.seed + 1000 * rank)       # independent episodes per rank (VLN-TR:141)
    instr = [INSTRUCTION_64] * B
    total = a.warm_steps + a.warmup + a.steps
    frames = []
    for _ in range(total):                                      # inputs resident in HBM before the timed region
        fr = ep.next()
        frames.append((dict(rgb=torch.from_numpy(fr.rgb).to(dev), depth=torch.from_numpy(fr.depth).to(dev)),
                       [p.tolist() for p in fr.positions], list(fr.headings), fr.patch_segm))

    def run(i):
        obs, pos, hd, segm = frames[i]
        return net.forward_logits(obs, instr, pos, hd, patch_segm=segm)

    hp = None
    if os.environ.get("D3D_BENCH_HP_STREAM") == "1":
        # experiment knob: the step's main stream at HIGH priority, so that the llava tower on the (default-priority) side stream only
        # fills what the critical path leaves idle (DESIGN.md section 8 "stream priorities")
        hp = torch.cuda.Stream(device=dev, priority=-1)
        torch.cuda.set_stream(hp)
    for i in range(a.warm_steps + a.warmup):
        run(i)
    torch.cuda.synchronize()
    DD.barrier()
    TIMER.enabled = True
    D.reset_counts()
    lengths_seen = []
    t0 = time.perf_counter()
    for i in range(a.warm_steps + a.warmup, total):
        lo = run(i)
        lengths_seen.append(list(net.last_lengths))
    torch.cuda.synchronize()
    DD.barrier()
    dt_own = time.perf_counter() - t0
    dt = DD.max_over_ranks(dt_own, device=dev)
    TIMER.enabled = False
    assert torch.isfinite(lo).all()
    # who actually worked: every rank reports its device and its own time (one all_gather_object); N ranks must be N distinct GPUs
    shared_hook = os.environ.get("D3D_SHARE_DEVICE0") == "1"
    rank_info = DD.gather_objects(dict(rank=rank, local_rank=local, device=device_identity(local), ms_per_step=round(dt_own / a.steps * 1e3, 3),
                                       env_steps=B * a.steps, finite=bool(torch.isfinite(lo).all())))
    devices_seen = len({r["device"] for r in rank_info})
    if rank == 0 and not shared_hook and devices_seen != world:
        sys.exit(f"bench.py: {world} ranks ran on {devices_seen} distinct GPU(s) {sorted({r['device'] for r in rank_info})}; not a {world}-GPU measurement")

    if rank == 0:
        ms = dt / a.steps * 1e3
        pruned = bool(getattr(net.llm, "PRUNE_LAST_LAYER", False))
        # Algorithmic FLOPs are accumulated PER TIMED STEP (the prompts grow by ~30 tokens per step as the memory fills): the step figures
        # are sums over the timed steps divided by the timed wall time, the launch figures sums over the timed launches.
        fl_steps = [step_flops(cfg, L, B, pruned_last_layer=pruned) for L in lengths_seen]
        fl = {k: sum(f[k] for f in fl_steps) / len(fl_steps) for k in fl_steps[0]}          # mean per step
        tokens_steps = [sum(L) for L in lengths_seen]
        rows_gemm = getattr(net.llm, "last_packed_rows", None) or B * max(lengths_seen[-1])   # rows the last step's GEMMs processed
        tsum = TIMER.summary()
        n_gu, ms_gu_raw = tsum.get("phi3.gate_up_proj", (0, float("nan")))
        # A HIP-event bracket on the launching stream measures the launch PLUS what the two event records cost there (each waits for the
        # work in front of it and writes a timestamp): an EMPTY bracket issued right behind every timed launch measures that cost under
        # the same conditions, and it is subtracted.  The rocprofv3 kernel trace of the same command (profiles/) is the cross-check.
        _, ms_empty = tsum.get("phi3.event_pair_overhead", (0, 0.0))
        ms_gu = ms_gu_raw - ms_empty
        l = cfg.llm
        recs = TIMER.records("phi3.gate_up_proj")                                           # (ms, {rows: real tokens of THAT launch})
        gu_flops_total = sum(2.0 * r.get("rows", tokens_steps[-1]) * l.hidden * 2 * l.mlp for _, r in recs)   # ALGORITHMIC: real tokens only
        gu_flops = gu_flops_total / max(len(recs), 1)                                       # mean per launch
        achieved = gu_flops / (ms_gu * 1e-3) / 1e12 if n_gu else float("nan")
        st = net.feature_fields.state
        traffic, traffic_note = None, None
        import glob
        pjs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_gate_up.json")))       # the latest round's PMC passes
        pj = pjs[-1] if pjs else ""
        if os.path.isfile(pj) and dict(D.BACKEND)["linear"] == "hip":
            # Fabric-side bytes per launch of this kernel: counters cannot be read inside an un-profiled run, so this is the figure of THIS
            # ROUND's rocprofv3 --pmc passes over the same kernel (profiles/rNN_pmc_gate_up.json: 2 * FETCH_SIZE + WRITE_SIZE, separate
            # passes; FETCH_SIZE counts 128-B requests at 64 B on gfx950, MI355X_MICROARCH.md), scaled by the rows of this run.
            try:
                pm = json.load(open(pj))
                traffic = int(pm["hbm_bytes_per_launch"] * rows_gemm / float(pm["rows"]))
                traffic_note = ("from_profile: profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, the kernel at M = %d), "
                                "scaled by launched rows; includes Infinity-Cache hits (algorithmic bytes %d)" % (os.path.basename(pj), pm["rows"], int(pm["algorithmic_bytes"] * rows_gemm / float(pm["rows"]))))
            except (KeyError, ValueError, OSError) as e:            # a profile file that lacks the fields is reported, it does not stop the benchmark
                traffic, traffic_note = None, "profiles/%s unusable (%s: %s)" % (os.path.basename(pj), type(e).__name__, e)
        out = {
            "metric": "nav steps/sec (RGB-D obs->action logits) at batch=8", "value": round(sum(r["env_steps"] for r in rank_info) / dt, 3), "unit": "env-steps/s",
            "n_gpus": world, "ranks_seen": len(rank_info), "devices_seen": devices_seen,
            "per_rank": [dict(rank=r["rank"], device=r["device"], ms_per_step=r["ms_per_step"]) for r in sorted(rank_info, key=lambda r: r["rank"])], "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "configs[2]: full Dynam3D-VLN step (3D tokens + llava-phi-3-mini prefill -> action logits), batch=8 synthetic 224x224 RGB-D, 1 MI355X per rank",
                       "batch_per_gpu": B, "operating_point": (f"warm: the timed steps are memory steps {a.warm_steps + a.warmup}..{total - 1} of the synthetic episodes "
                                           f"({a.warm_steps} untimed advance steps + {a.warmup} warm-up steps before them)"),
                       "memory_steps_timed": [a.warm_steps + a.warmup, total - 1],
                       "S_tokens_first_timed_step": lengths_seen[0], "S_tokens_last_timed_step": lengths_seen[-1], "S_tokens": lengths_seen[-1],
                       "real_tokens_per_timed_step": tokens_steps, "mean_real_tokens_per_step": round(sum(tokens_steps) / len(tokens_steps), 1),
                       "Ni": net.last_counts["Ni"], "Nz": net.last_counts["Nz"], "rows_per_env": st.count(0, st.ROWS),
                       "instances_per_env": st.count(0, st.LIVE), "clip_dtype": str(cfg.clip_dtype), "llm_dtype": str(cfg.llava_dtype),
                       "token_builder_dtype": "float32", "parallelism": f"episode-parallel x{world} (no data-path collective)" + (" [test hook: ranks share cuda:0]" if shared_hook else ""),
                       "dense_backend": dict(D.BACKEND), "strict_hip": bool(D.STRICT),
                       "dense_dispatch_per_step": {k: round(v / a.steps, 2) for k, v in D.counts()["hip"].items()},
                       "fallbacks": int(sum(D.counts()["fallback"].values()))},
            "roofline": {"bound": "mfma", "kernel": "phi3.gate_up_proj GEMM + fused SwiGLU (sum(S_b) x 3072 x 16384, bf16): k_gemm_nt_256<bf16,SwiGLU> on the full rounds (+ k_gemm_nt<bf16,SwiGLU> on the last rows when the last round is at most half full); avg_launch_ms covers the whole projection", "gemm_rows_launched_last_step": rows_gemm, "mean_real_tokens_per_launch": round(gu_flops / (2.0 * l.hidden * 2 * l.mlp), 1),
                         "algorithmic_gflop_per_launch_mean": round(gu_flops / 1e9, 2), "achieved": round(achieved, 1),
                         "peak": PEAK_BF16_DENSE_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / PEAK_BF16_DENSE_TFLOPS, 4), "traffic": traffic, "traffic_note": traffic_note,
                         "launches_timed": n_gu, "avg_launch_ms": round(ms_gu, 4), "avg_launch_ms_event_bracket": round(ms_gu_raw, 4),
                         "event_pair_overhead_ms": round(ms_empty, 4),
                         "accounting": "achieved = (sum over the timed launches of 2 * real_tokens * 3072 * 16384) / (sum of their HIP-event durations - event-pair overhead); "
                                       "step_* = mean over the timed steps of the algorithmic FLOPs at that step's own S_b",
                         "step_total_tflop": round(fl["total"] / 1e12, 2), "step_frac_of_peak": round(fl["total"] / (ms * 1e-3) / 1e12 / PEAK_BF16_DENSE_TFLOPS, 4),
                         "step_flop_split_tflop": {k: round(v / 1e12, 3) for k, v in fl.items() if k != "total"}},
        }
        do_cpu = a.cpu_baseline == "on" or (a.cpu_baseline == "auto" and world == 1)
        if do_cpu:
            try:
                out["cpu_baseline"] = cpu_baseline(cfg, a.seed, B, a.warm_steps + a.warmup, lengths_seen[0])
            except Exception as e:  # never lose the GPU line because the host baseline failed
                out["cpu_baseline"] = {"value": None, "error": repr(e)}
        print(json.dumps(out), flush=True)
    DD.barrier()
    DD.shutdown()


if __name__ == "__main__":
    main()
