"""The sixteen-lane kernels read neighbours' registers through DPP operands written as inline assembly (cdkf_lpe_kernels.h:
lpe_fmac_bcast and the 32-bit v_fmac_f32_dpp / v_mul_f32_dpp forms), which the compiler's hazard recogniser cannot see into.
gfx9 rule: a VGPR written by a VALU instruction must not be read as a DPP source within the next two wait states.  This test
disassembles the gfx950 code objects the library is linked from and checks EVERY DPP instruction against that rule, so a compiler
release that schedules differently fails here, on the CPU, instead of silently corrupting sweeps on the GPU.  CPU only."""
import glob
import gzip
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


def _listing_dir(obj):
    """build/disasm/ beside build/csrc/ (listed in .gpurunignore: compressed listings the GPU box has no use for)."""
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(obj))), "disasm")
    os.makedirs(d, exist_ok=True)
    return d


def _disassemble(obj, tmp):
    keep = os.path.join(_listing_dir(obj), "%s_%d.s.gz" % (os.path.basename(obj), int(os.path.getmtime(obj))))   # (shared with
    if os.path.exists(keep):                                                                                             #  scripts/check_exec_prologue.py)
        return gzip.open(keep, "rt", errors="replace").read()
    text = _disassemble_now(obj, tmp)
    try:
        gzip.open(keep, "wt", compresslevel=1).write(text)
    except OSError:
        pass
    return text


def _disassemble_now(obj, tmp):
    fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "dev.co")
    subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat])
    subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={fat}",
                           "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"])
    return subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", co], capture_output=True, text=True, check=True).stdout


_BRANCH = re.compile(r"^s_(?:branch|cbranch_\w+)$")


def _parse(text):
    """[(function, [instruction])], an instruction = dict(op, ops, line, addr): addr from objdump's trailing `// ADDR: ENCODING` (None
    in hand-written listings without one)."""
    funcs, cur = [], None
    for raw in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", raw)
        if m:
            cur = []
            funcs.append((m.group(1), cur))
            continue
        body, _, tail = raw.partition("//")
        line = body.strip()
        if not line or line.endswith(":") or cur is None:
            continue
        am = re.match(r"\s*([0-9A-Fa-f]{6,16}):", tail)
        parts = line.split(None, 1)
        ops = [t.strip() for t in (parts[1] if len(parts) > 1 else "").replace(" row_", ", row_").replace(" quad_", ", quad_").split(",")]
        cur.append({"op": parts[0], "ops": ops, "line": line, "addr": int(am.group(1), 16) if am else None})
    return funcs


def _violations(text):
    """(function, instruction, producer) for every DPP source register written by a VALU fewer than two wait states earlier.  A
    virtual clock counts issue slots: one per instruction, N + 1 for `s_nop N`; between a producer issued at tp and a consumer at
    tc lie tc - tp - 1 wait states.  CONTROL FLOW (round 5): the producers still inside the window at the end of
    every predecessor -- the fall-through AND each branch whose target the instruction is, loop back edges included (s_branch /
    s_cbranch_* carry a dword offset relative to the next instruction) -- enter the window of the block they lead to, as if the
    consumer issued right after the branch (a taken branch costs more than one slot; counting one is the conservative choice)."""
    out, n_dpp = [], 0
    for func, ins in _parse(text):
        index_of = {i["addr"]: k for k, i in enumerate(ins) if i["addr"] is not None}
        sources = {}   # target instruction index -> [branch instruction index]
        for k, i in enumerate(ins):
            if _BRANCH.match(i["op"]) and i["addr"] is not None and i["ops"] and re.fullmatch(r"\d+", i["ops"][0]):
                off = int(i["ops"][0])
                off = off - 65536 if off >= 32768 else off
                tgt = index_of.get(i["addr"] + 4 + 4 * off)
                if tgt is not None:
                    sources.setdefault(tgt, []).append(k)
        after = [None] * len(ins)   # window after instruction k: [(slots since the producer issued, written VGPRs, text)]
        for sweep in range(3):      # (a window reaches two slots back: the second sweep sees every back edge's source, the third confirms)
            found, dpp = [], 0
            window = []             # producers with their age in slots at the point BEFORE the next instruction issues
            for k, i in enumerate(ins):
                for src in sources.get(k, []):
                    if after[src] is not None:
                        window = window + [w for w in after[src] if w not in window]
                op, ops = i["op"], i["ops"]
                cost = (int(ops[0], 0) + 1) if op == "s_nop" else 1
                if op.endswith("_dpp"):
                    dpp += 1
                    src_regs = _regs(ops[1]) if len(ops) > 1 else set()
                    for age, written, txt in window:
                        if written & src_regs and age < 2:   # age = wait states between producer and this consumer
                            found.append((func, i["line"], txt))
                window = [(age + cost, w, t) for age, w, t in window if age + cost < 3]
                if op.startswith("v_") and not op.startswith(("v_cmp", "v_readlane", "v_readfirstlane", "v_accvgpr_write")):
                    written = _regs(ops[0]) if ops and ops[0] else set()
                    if written:
                        window.append((0, written, i["line"]))
                after[k] = list(window)
                if op in ("s_branch", "s_endpgm", "s_setpc_b64"):   # no fall-through
                    window = []
        out += found
        n_dpp += dpp
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


def test_the_checker_follows_branches_and_back_edges():
    """A producer in the last slot of a loop body and a DPP read in the first slot of the loop head (reached by the back edge only), a
    forward branch over enough instructions, and a clean loop: the first two are reported, the third is not."""
    text = """0000000000001000 <k>:
\tv_mov_b32_e32 v1, v0                       // 000000001000: 7E020300
\ts_nop 2                                    // 000000001004: BF800002
\tv_mul_f32_dpp v23, v20, v24 row_ror:4 row_mask:0xf bank_mask:0xf // 000000001008: 0A2E30FA FF012414
\tv_mov_b32_e32 v9, v8                       // 000000001010: 7E120308
\tv_mul_f32_e32 v20, v21, v22                // 000000001014: 0A282D15
\ts_cbranch_scc1 65531                       // 000000001018: BF85FFFB
\tv_mul_f32_e32 v30, v21, v22                // 00000000101C: 0A3C2D15
\ts_cbranch_vccz 2                           // 000000001020: BF860002
\tv_mov_b32_e32 v2, v3                       // 000000001024: 7E040303
\tv_mov_b32_e32 v4, v5                       // 000000001028: 7E080305
\tv_mul_f32_dpp v23, v30, v24 row_ror:4 row_mask:0xf bank_mask:0xf // 00000000102C: 0A2E30FA FF01241E
\tv_mul_f32_e32 v40, v21, v22                // 000000001034: 0A502D15
\ts_nop 1                                    // 000000001038: BF800001
\ts_cbranch_scc1 65533                       // 00000000103C: BF85FFFD
\ts_endpgm                                   // 000000001040: BF810000
"""
    # back edge 0x1018 -> 0x1008: v20 written at 0x1014, one slot (the branch) before the DPP read at the loop head
    # forward branch 0x1020 -> 0x102C: v30 written at 0x101C, the branch in between: one wait state
    bad, n = _violations(text)
    assert n == 2, n
    assert len(bad) == 2 and bad[0][2].startswith("v_mul_f32_e32 v20") and bad[1][2].startswith("v_mul_f32_e32 v30"), bad


def test_run_time_compiled_code_objects_obey_the_rule_too(tmp_path):
    """The hipRTC code objects of the in-tree cache (cd_dynamax_amd/lib/rtc_cache/*.co: a 16-byte header, the lowered name, the code
    object -- launch_custom.hip:rtc_cache_store): whatever DPP instruction the compiler put there is checked against the same rule
    (their sources hold no hand-written DPP: the count may be zero).  A sample of the cache, newest first; skipped without one."""
    import struct
    files = sorted(glob.glob(os.path.join(ROOT, "cd_dynamax_amd", "lib", "rtc_cache", "*.co")), key=os.path.getmtime, reverse=True)[:12]
    if not files or not os.path.exists(os.path.join(LLVM, "llvm-objdump")):
        pytest.skip("no run-time compiled code objects in the tree (they are built on first use on a GPU box)")
    checked = 0
    for path in files:
        raw = open(path, "rb").read()
        magic, nname, lo, hi = struct.unpack("<4I", raw[:16])
        if magic != 0x43524b43:
            continue
        co = tmp_path / "k.co"
        co.write_bytes(raw[16 + nname:16 + nname + (hi << 32 | lo)])
        text = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", str(co)], capture_output=True, text=True, check=True).stdout
        bad, _ = _violations(text)
        assert not bad, (os.path.basename(path), bad[:3])
        checked += 1
    assert checked > 0
