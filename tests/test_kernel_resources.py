"""CPU suite: the built library's kernels against a scratch budget, read from the code objects' metadata notes (tools/kernel_resources.py).

Round 6 shipped -- for a few commits -- a build in which a harmless-looking source change (a one-trip `for` around the 256 x 256 GEMM's
body) made hipcc spill 9-56 VGPRs to scratch in every instantiation.  It passed the GEMM parity tests and three golden-parity bench runs,
and then produced wrong logits or a GPU memory access fault in about every second process of tests/test_gpu_full_step.py
(profiles/r06_gemm_scratch_regression.txt).  A register-starved hand-scheduled kernel that starts to spill is a different program: this
test makes that visible on the CPU box, at build time, kernel by kernel."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

# kernel family -> bytes of scratch per lane it may use (the state of the validated round-5 build; everything else: none)
BUDGET = {
    "k_gemm_nt_256": 12,            # a few training / diagnostic instantiations spill 1-2 registers; the step's (EPI 0/1/4/5/6, DMAV 2) none
    "k_flash_attn_dma": 20,         # <*, 96, causal, 4 waves>: 4 registers (Phi-3 prefill; since round 4)
    "k_ffdev_plan_merge": 100,      # device planner: one workgroup, local arrays
    "k_ffdev_plan_zones": 40,
    "k_group_stats4": 56,
    "k_phi3_decode_token": 460,     # opt-in persistent decode kernel (D3D_DECODE_PERSISTENT=1), not on the default path
}
# the step's own GEMM instantiations <BF16, EPI, KFULL, SPLIT, DMAV>: no scratch at all
STEP_GEMM_EPI = {"0", "1", "4", "5", "6"}


@pytest.fixture(scope="module")
def table():
    from kernel_resources import LIB, kernel_table
    if not os.path.exists(LIB):
        pytest.skip("libdynam3d_hip.so is not built")
    rows = kernel_table(LIB)
    assert len(rows) > 300, len(rows)                       # (all sixteen code objects were found and parsed)
    return rows


def test_no_kernel_uses_scratch_beyond_its_budget(table):
    over = []
    for r in table:
        fam = r["short"].split("<")[0]
        budget = BUDGET.get(fam, 0)
        if fam == "k_flash_attn_dma" and not r["short"].startswith(("k_flash_attn_dma<0,96,1,4", "k_flash_attn_dma<1,96,1,4")):
            budget = 0                                          # only the Phi-3 prefill instantiation (head_dim 96, causal, 4 waves) has its 4 registers
        if fam == "k_gemm_nt_256":
            a = r["short"][len("k_gemm_nt_256<"):-1].split(",")
            if a[1] in STEP_GEMM_EPI and a[2] == "1" and a[4] == "2":
                budget = 0                                      # the step's own instantiations: none
        if r["scratch"] > budget or r["dynamic_stack"]:
            over.append((r["short"], r["scratch"], r["spills"]))
    assert not over, "kernels with scratch beyond the budget (a spilling build -- look at the last change of that kernel): " + repr(over)


def test_the_steps_gemm_and_attention_instantiations_do_not_spill(table):
    seen = 0
    for r in table:
        if r["short"].startswith("k_gemm_nt_256<"):
            bf16, epi, kfull, split, dmav = r["short"][len("k_gemm_nt_256<"):-1].split(",")[:5]
            if epi in STEP_GEMM_EPI and kfull == "1" and dmav == "2":
                seen += 1
                assert r["scratch"] == 0 and r["spills"] == 0, r
        if r["short"].startswith("k_gemm_nt<") or r["short"].startswith("k_flash_attn_dma<0,64") or r["short"].startswith("k_flash_attn_dma<1,64"):
            seen += 1
            assert r["scratch"] == 0 and r["spills"] == 0, r
        if r["short"].startswith("k_flash_attn_pipe<"):                      # (opt-in attn4 kernel)
            assert r["scratch"] == 0, r
    assert seen >= 20, seen
