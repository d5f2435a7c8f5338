"""CPU suite: the assembly filter of the library's build (zkevm-circuits_amd/build.py) removes the wait states that only an
empty asm statement asked for, and nothing else."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_spec = importlib.util.spec_from_file_location("zkhip_build", os.path.join(ROOT, "zkevm-circuits_amd", "build.py"))
build = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(build)

VEIL = "\t;;#ASMSTART\n\t;;#ASMEND\n"


def test_wait_states_behind_empty_asm_statements_are_removed():
    text = ("\tv_mad_u64_u32 v[16:17], s[2:3], v19, s38, v[16:17]\n" + VEIL + "\ts_nop 0\n"
            "\tv_ashrrev_i64 v[16:17], 29, v[16:17]\n" + VEIL + VEIL + "\ts_nop 0\n"
            "\tv_mad_i64_i32 v[16:17], s[2:3], v66, v54, v[16:17]\n")
    out, removed = build.trim_wait_states(text)
    assert removed == 2 and "s_nop" not in out
    assert out.count("v_mad") == 2 and out.count(";;#ASMSTART") == 3


def test_wait_states_that_may_cover_a_real_hazard_stay():
    cases = [
        # a DPP consumer: two wait states after a vector write are the hardware's
        "\tv_and_b32_e32 v5, 0x1fffffff, v4\n" + VEIL + "\ts_nop 1\n\tv_mov_b32_dpp v6, v5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n",
        "\tv_and_b32_e32 v5, 0x1fffffff, v4\n" + VEIL + "\ts_nop 0\n\tv_mov_b32_dpp v6, v5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n",
        # consumers and producers outside the list of plain vector instructions
        "\tv_mad_u64_u32 v[16:17], s[2:3], v19, s38, v[16:17]\n" + VEIL + "\ts_nop 0\n\tv_readlane_b32 s4, v16, 3\n",
        "\tv_mad_u64_u32 v[16:17], s[2:3], v19, s38, v[16:17]\n" + VEIL + "\ts_nop 0\n\tv_cndmask_b32_e32 v3, v16, v2, vcc\n",
        "\tv_rcp_f32_e32 v5, v4\n" + VEIL + "\ts_nop 0\n\tv_and_b32_e32 v6, 1, v5\n",
        "\ts_cbranch_execz .LBB0_3\n" + VEIL + "\ts_nop 0\n\tv_and_b32_e32 v6, 1, v5\n",
        # a wait state that no asm statement stands in front of
        "\tv_mad_u64_u32 v[16:17], s[2:3], v19, s38, v[16:17]\n\ts_nop 0\n\tv_ashrrev_i64 v[16:17], 29, v[16:17]\n",
        # a label between the wait state and the next instruction
        "\tv_mad_u64_u32 v[16:17], s[2:3], v19, s38, v[16:17]\n" + VEIL + "\ts_nop 0\n.LBB0_4:\n\tv_and_b32_e32 v6, 1, v5\n",
    ]
    for text in cases:
        out, removed = build.trim_wait_states(text)
        assert removed == 0 and out == text, text


def test_every_hip_source_goes_through_the_filter():
    hip = sorted(s for s in build.SOURCES if s.endswith(".hip"))
    assert sorted(build.TRIMMED) == hip
