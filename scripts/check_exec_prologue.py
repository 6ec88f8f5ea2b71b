"""Checker for the ROCm 7.2 miscompile behind rounds 3 - 5's optimisation-level-dependent wrong results (NOTES.md R5.1, profiles/
r05_j_root_cause.txt): vector spill code placed at the top of a structured-control-flow FLOW block, BEFORE the `s_or_saveexec_b64` that
re-enables the lanes of the other arm -- so the stores (and VGPR -> AGPR copies) run under the then-arm's execution mask and the lanes that
took the other arm keep stale spill slots.  In a correct build a block that restores the mask starts with that restore (scalar
instructions may precede it).  Rule checked on a disassembly (llvm-objdump -d): between a block's first instruction -- a branch target,
the instruction after an unconditional branch / s_setpc / s_endpgm, or a function's entry -- and an `s_or_saveexec_b64` in that block
there is no vector store (scratch_ / global_ / buffer_ / flat_ / ds_ store), no v_accvgpr_write and no other lane-masked VALU write.
The shape matched is the lowered if / else itself: `s_and_saveexec_b64 sX, c; s_xor_b64 sY, exec, sX; s_cbranch_execz FLOW` and, from
FLOW on in a straight line, the `s_or_saveexec_b64 .., sY` that consumes the SAME saved mask: whatever lane-masked instruction sits between
FLOW and that restore runs for the wrong lanes (a then-arm that falls through into its flow block ends BEFORE the label and is not judged).
The plain `if` is matched the same way: `s_and_saveexec_b64 sX, c; s_cbranch_execz JOIN` and, from JOIN on, `s_or_b64 exec, exec, sX`.

  python scripts/check_exec_prologue.py <code object or .o with a .hip_fatbin or .s listing> [...]      exit code 1 when something is found
Also used by tests/test_exec_prologue.py over the library's objects and the in-tree hipRTC cache."""
import gzip
import os
import re
import struct
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
_UNMASKED = ("v_readlane", "v_writelane", "v_readfirstlane")   # do not depend on the execution mask


def _listing_dir(obj):
    """build/disasm/ beside build/csrc/ (listed in .gpurunignore: compressed listings the GPU box has no use for)."""
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(obj))), "disasm")
    os.makedirs(d, exist_ok=True)
    return d


def disassemble(path, tmp):
    raw = open(path, "rb").read()
    if path.endswith(".s"):
        return raw.decode(errors="replace")
    if path.endswith(".o"):      # (the library's objects are tens of MB of device code: the listing is kept beside the build, keyed by the object's mtime)
        keep = os.path.join(_listing_dir(path), "%s_%d.s.gz" % (os.path.basename(path), int(os.path.getmtime(path))))
        if os.path.exists(keep):
            return gzip.open(keep, "rt", errors="replace").read()
        text = _disassemble(path, raw, tmp)
        try:
            gzip.open(keep, "wt", compresslevel=1).write(text)
        except OSError:
            pass
        return text
    return _disassemble(path, raw, tmp)


def _disassemble(path, raw, tmp):
    co = path
    if raw[:4] == b"CKRC":      # an entry of the library's hipRTC cache: header, name, code object
        magic, nname, lo, hi = struct.unpack("<4I", raw[:16])
        co = os.path.join(tmp, "k.co")
        open(co, "wb").write(raw[16 + nname:16 + nname + (hi << 32 | lo)])
    elif path.endswith(".o"):
        fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "dev.co")
        subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", path, fat])
        if subprocess.call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={fat}",
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], stderr=subprocess.DEVNULL) != 0:
            return ""      # (no device code in this object)
    return subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", co], capture_output=True, text=True, check=True).stdout


def violations(text):
    """[(function, address of the s_or_saveexec, [offending instructions])]"""
    funcs, cur = [], None
    for raw in text.splitlines():
        m = re.match(r"^([0-9a-f]+) <(.+)>:$", raw)
        if m:
            cur = {"name": m.group(2), "start": int(m.group(1), 16), "ins": []}
            funcs.append(cur)
            continue
        body, _, tail = raw.partition("//")
        line = body.strip()
        if not line or cur is None or line.endswith(":"):
            continue
        am = re.match(r"\s*([0-9A-Fa-f]+):", tail)
        if not am:
            continue
        tm = re.search(r"<[^>+]+\+0x([0-9a-f]+)>", tail)
        cur["ins"].append((int(am.group(1), 16), line, (cur["start"] + int(tm.group(1), 16)) if tm else None))
    out = []
    for f in funcs:
        ins = f["ins"]
        index = {addr: k for k, (addr, _, _) in enumerate(ins)}
        for k, (addr, line, target) in enumerate(ins):
            # the lowered `if (c) { A } else { B }`:  s_and_saveexec_b64 sX, c ; s_xor_b64 sY, exec, sX ; s_cbranch_execz FLOW ; A ... ;
            #                                FLOW:  s_or_saveexec_b64 sZ, sY ; ...        (sY: the lanes of the other arm)
            if line.split()[0] != "s_cbranch_execz" or target is None or target <= addr or target not in index:
                continue
            sy = sx = None
            for a2, l2, _ in ins[max(0, k - 3):k]:
                m = re.match(r"s_xor_b64 (s\[\d+:\d+\]), exec, s\[\d+:\d+\]", l2)
                if m:
                    sy = m.group(1)
                m = re.match(r"s_and_saveexec_b64 (s\[\d+:\d+\]),", l2)
                if m:
                    sx = m.group(1)
            if sy is None and sx is None:
                continue
            if sy is None:
                sy = sx      # a plain `if`: the skip lands on the JOIN block, whose first instruction is  s_or_b64 exec, exec, sX
            j = index[target]
            bad = []
            for a2, l2, _ in ins[j:j + 400]:
                o2 = l2.split()[0]
                toks = l2.replace(",", " ").split()
                if (o2 == "s_or_saveexec_b64" and toks[2] == sy) or (o2 == "s_or_b64" and toks[1:4] == ["exec", "exec", sy]):
                    if bad:
                        out.append((f["name"], a2, bad))
                    break
                if o2.startswith(("s_branch", "s_cbranch", "s_setpc", "s_endpgm")) or (o2.startswith("s_") and sy in l2.replace(",", " ").split()[1:2]):
                    break      # (control flow, or the saved mask is redefined: not the simple shape this rule is about)
                if re.match(r"^(scratch|global|buffer|flat)_store|^ds_write|^ds_store", o2) or o2.startswith("v_accvgpr_write") or \
                        (o2.startswith("v_") and not o2.startswith(_UNMASKED) and not o2.startswith("v_cmp")):
                    bad.append("%x: %s" % (a2, l2))
    return out


def main():
    rc = 0
    with tempfile.TemporaryDirectory() as tmp:
        for path in sys.argv[1:]:
            v = violations(disassemble(path, tmp))
            print("%s: %d block(s) with lane-masked instructions before their s_or_saveexec_b64" % (path, len(v)))
            for name, addr, bad in v[:10]:
                print("   %s, restore at %#x: %d instruction(s), e.g. %s" % (name[:70], addr, len(bad), "; ".join(bad[:3])))
            rc |= bool(v)
    sys.exit(rc)


if __name__ == "__main__":
    main()
