"""The sixteen-lane kernels read neighbours' registers through DPP operands written as inline assembly (cdkf_lpe_kernels.h:
lpe_fmac_bcast and the 32-bit v_fmac_f32_dpp / v_mul_f32_dpp forms), which the compiler's hazard recogniser cannot see into.
gfx9 rule: a VGPR written by a VALU instruction must not be read as a DPP source within the next two wait states.  This test
disassembles the gfx950 code objects the library is linked from and checks EVERY DPP instruction against that rule, so a compiler
release that schedules differently fails here, on the CPU, instead of silently corrupting sweeps on the GPU.  CPU only."""
import glob
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
UNITS = ["launch_ekf", "launch_ukf", "launch_eks", "launch_grad", "launch_w8"]  # the translation units with DPP code


def _regs(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def _disassemble(obj, tmp):
    fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "dev.co")
    subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat])
    subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={fat}",
                           "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"])
    return subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", co], capture_output=True, text=True, check=True).stdout


def _violations(text):
    """(function, instruction, producer) for every DPP source register written by a VALU fewer than two wait states earlier.  A
    virtual clock counts issue slots: one per instruction, N + 1 for `s_nop N`; between a producer issued at tp and a consumer at
    tc lie tc - tp - 1 wait states."""
    out, func, recent, clock, n_dpp = [], "?", [], 0, 0  # recent: (issue time, written VGPRs, text) of the latest VALU writers
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            func, recent = m.group(1), []
            continue
        line = line.split("//")[0].strip()
        if not line:
            continue
        if line.endswith(":"):  # a label: predecessors unknown (fall-through and branch targets are the recogniser's business)
            recent = []
            continue
        parts = line.split(None, 1)
        op, ops = parts[0], [t.strip() for t in (parts[1] if len(parts) > 1 else "").replace(" row_", ", row_").replace(" quad_", ", quad_").split(",")]
        if op == "s_nop":
            clock += int(ops[0], 0) + 1
            continue
        clock += 1
        if op.endswith("_dpp"):
            n_dpp += 1
            src = _regs(ops[1]) if len(ops) > 1 else set()
            for tp, written, txt in recent:
                if written & src and clock - tp - 1 < 2:
                    out.append((func, line, txt))
        if op.startswith("v_") and not op.startswith(("v_cmp", "v_readlane", "v_readfirstlane", "v_accvgpr_write")):
            written = _regs(ops[0]) if ops and ops[0] else set()
            if written:
                recent = [(tp, w, t) for tp, w, t in recent if clock - tp < 3] + [(clock, written, line)]
    return out, n_dpp


@pytest.mark.parametrize("unit", UNITS)
def test_no_dpp_read_within_two_wait_states_of_its_producer(unit, tmp_path):
    obj = os.path.join(ROOT, "build", "csrc", unit + ".o")
    if not os.path.exists(obj) or shutil.which("objcopy") is None or not os.path.exists(os.path.join(LLVM, "llvm-objdump")):
        pytest.skip("needs the library's object files (make -C cd_dynamax_amd/csrc) and the ROCm LLVM tools")
    bad, n_dpp = _violations(_disassemble(obj, str(tmp_path)))
    assert n_dpp > (100 if unit != "launch_w8" else 4), f"{unit}: only {n_dpp} DPP instructions found -- the disassembly did not come out as expected"
    assert not bad, f"{len(bad)} DPP reads inside the hazard window, e.g. {bad[:3]}"


def test_the_checker_flags_a_planted_hazard():
    text = """0000000000001000 <k>:
\tv_add_f64 v[2:3], v[4:5], v[6:7]          // 1
\tv_mov_b32_e32 v9, v8                       // 2
\tv_fmac_f64_dpp v[10:11], v[2:3], v[12:13] row_newbcast:3 row_mask:0xf bank_mask:0xf // 3
\tv_add_f64 v[2:3], v[4:5], v[6:7]          // 4
\ts_nop 1                                    // 5
\tv_fmac_f64_dpp v[10:11], v[2:3], v[12:13] row_newbcast:3 row_mask:0xf bank_mask:0xf // 6
\tv_mul_f32_e32 v20, v21, v22                // 7
\tv_mul_f32_dpp v23, v20, v24 row_ror:4 row_mask:0xf bank_mask:0xf // 8
\tv_mul_f32_e32 v30, v21, v22                // 9
\ts_nop 0                                    // 10
\tv_mul_f32_dpp v23, v30, v24 row_ror:4 row_mask:0xf bank_mask:0xf // 11: one wait state only
\tv_mul_f32_e32 v31, v21, v22                // 12
\ts_nop 0                                    // 13
\tv_mov_b32_e32 v40, v41                     // 14
\tv_mul_f32_dpp v23, v31, v24 row_ror:4 row_mask:0xf bank_mask:0xf // 15: two wait states
"""
    bad, n = _violations(text)
    assert n == 5 and len(bad) == 3 and "// 11" not in bad[2][1] and "// 3" not in bad[0][1] and bad[0][1].startswith("v_fmac_f64_dpp") and bad[1][1].startswith("v_mul_f32_dpp")
