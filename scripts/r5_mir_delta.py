"""Round 5: WHICH of the pre-RA si-shrink-instructions rewrites flips a run-time compiled kernel from right to wrong.

The pass bisection of scripts/r5_o3_probe.py names the pass; this script finds the instruction.  For the kernel a case names:
  1. the generated source (launch_custom.hip) is compiled offline to machine IR stopped BEFORE the first si-shrink-instructions
     (k_pre.mir) and that pass alone is run on it (k_post.mir): the two files differ in a few thousand lines -- VOP3 -> VOP2 encodings,
     commuted compares, register-allocation hints;
  2. a HYBRID = k_pre.mir with a chosen subset of those line changes applied is pushed through the rest of the -O3 pipeline (llc
     -start-after=si-shrink-instructions, ld.lld), loaded in place of the library's own build of that kernel (CDKF_RTC_OVERRIDE_CO)
     and checked against the oracle on the GPU (scripts/r5_o3_probe.py case <name>);
  3. delta debugging (ddmin) over the changed lines finds a minimal set whose application makes the result wrong.
Output: gpurun_out/r5_o3/<case>_delta.txt (+ the minimal hybrid and the two code objects).
  gpurun -- 'python scripts/r5_mir_delta.py d2grad'"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "scripts")]
OUT = os.path.join(ROOT, "gpurun_out", "r5_o3")
LLVM = "/opt/rocm/lib/llvm/bin"
CSRC = os.path.join(ROOT, "cd_dynamax_amd", "csrc")

import r5_o3_probe as probe  # noqa: E402


def sh(cmd, **kw):
    p = subprocess.run(cmd, capture_output=True, text=True, **kw)
    if p.returncode != 0:
        raise RuntimeError(" ".join(cmd) + "\n" + p.stderr[-3000:])
    return p


def generated_source(name, work):
    """The .hip unit of the case's kernel, as launch_custom.hip generates it (cdkf_custom_drift_compile with CDKF_CUSTOM_DUMP: no GPU)."""
    from cd_dynamax_amd import _ffi
    os.environ["CDKF_CUSTOM_DUMP"] = work
    try:
        if name == "d2grad":
            from test_custom_drift import NL_F
            k = _ffi.register_custom_drift(2, 3, NL_F, None, None)
            rc = _ffi.lib().cdkf_custom_drift_compile(k, 8, 1, 3, 1, 0)
        else:
            raise SystemExit("no generator for case " + name)
        assert rc == 0, _ffi.lib().cdkf_last_error().decode()
    finally:
        del os.environ["CDKF_CUSTOM_DUMP"]
    srcs = sorted(f for f in os.listdir(work) if f.endswith(".hip"))
    assert srcs, os.listdir(work)
    return os.path.join(work, srcs[0])


def hunks(pre, post):
    """[(pre_lo, pre_hi, [post lines])]: the line ranges of `pre` a normal diff replaces / deletes / inserts after; one-to-one
    replacements of several lines are split into single lines."""
    p = subprocess.run(["diff", pre, post], capture_output=True, text=True)
    post_lines = open(post).read().split("\n")
    hs = []
    for ln in p.stdout.split("\n"):
        if not ln or ln[0] in "<>-":
            continue
        op = [c for c in ln if c in "acd"][0]
        l, r = ln.split(op)
        l0, l1 = (int(x) for x in (l.split(",") * 2)[:2])
        r0, r1 = (int(x) for x in (r.split(",") * 2)[:2])
        if op == "a":
            hs.append((l0, l0, post_lines[r0 - 1:r1]))          # insert after pre line l0 (1-based): empty range at index l0
        elif op == "d":
            hs.append((l0 - 1, l1, []))
        elif l1 - l0 == r1 - r0:
            for k in range(l1 - l0 + 1):
                hs.append((l0 - 1 + k, l0 + k, [post_lines[r0 - 1 + k]]))
        else:
            hs.append((l0 - 1, l1, post_lines[r0 - 1:r1]))
    return hs


def main():
    name = sys.argv[1]
    os.makedirs(OUT, exist_ok=True)
    rep = open(os.path.join(OUT, f"{name}_delta.txt"), "a")

    def say(*a):
        s = " ".join(str(x) for x in a)
        print(s, flush=True)
        rep.write(s + "\n")
        rep.flush()
    work = tempfile.mkdtemp(prefix="r5_delta_")
    src = generated_source(name, work)
    pre, post = os.path.join(work, "k_pre.mir"), os.path.join(work, "k_post.mir")
    sh(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-I" + CSRC, "-Wno-pass-failed", "-x", "hip",
        "-mllvm", "-stop-before=si-shrink-instructions", "-S", src, "-o", pre])
    llc = [os.path.join(LLVM, "llc"), "-O3", "-mtriple=amdgcn-amd-amdhsa", "-mcpu=gfx950"]
    sh(llc + ["-run-pass=si-shrink-instructions", pre, "-o", post])
    # (llc re-serialises: compare like with like)
    pre_rt = os.path.join(work, "k_pre_rt.mir")
    sh(llc + ["-run-pass=none", pre, "-o", pre_rt])
    pre = pre_rt
    pre_lines = open(pre).read().split("\n")
    H = hunks(pre, post)
    say("case", name, "source", os.path.basename(src), "lines", len(pre_lines), "changed-line hunks", len(H))
    count = [0]

    def test(subset, keep=None):
        """True = GOOD.  subset: indices into H applied on top of pre."""
        count[0] += 1
        chosen = sorted((H[i] for i in subset), key=lambda h: (h[0], h[1]))
        out, pos = [], 0
        for lo, hi, new in chosen:
            out.extend(pre_lines[pos:lo])
            out.extend(new)
            pos = hi
        out.extend(pre_lines[pos:])
        mir = os.path.join(work, "hybrid.mir")
        open(mir, "w").write("\n".join(out))
        obj, co = os.path.join(work, "hybrid.o"), os.path.join(work, "hybrid.co")
        try:
            sh(llc + ["-start-after=si-shrink-instructions", mir, "-filetype=obj", "-o", obj])
        except RuntimeError as e:
            say("  llc failed on a hybrid of", len(subset), "hunks:", str(e)[-300:].replace("\n", " | "))
            return None
        sh([os.path.join(LLVM, "ld.lld"), "-shared", obj, "-o", co])
        env = dict(os.environ, CDKF_RTC_OVERRIDE_CO=co, CDKF_RTC_EXTRA_OPTS_ONLY=probe.TAGS[name])
        p = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "r5_o3_probe.py"), "case", name], capture_output=True, text=True, env=env, timeout=3000)
        res = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT")]
        good = bool(res) and res[-1].split()[1] == "GOOD"
        if keep:
            for f, suffix in ((mir, ".mir"), (co, ".co")):
                with open(f, "rb") as a, open(os.path.join(OUT, f"{name}_{keep}{suffix}"), "wb") as b:
                    b.write(a.read())
        return good
    none_ok = test([], keep="none")
    all_ok = test(range(len(H)), keep="all")
    say("no shrink rewrite applied:", "GOOD" if none_ok else "BAD", "| all applied:", "GOOD" if all_ok else "BAD")
    if not none_ok or all_ok:
        say("the pre-RA shrink rewrites do not separate GOOD from BAD in this pipeline: stop")
        return
    # by KIND of rewrite first: the pass changes instruction encodings (VOP3 -> VOP2, commuted compares) AND leaves register-allocation
    # hints in the function's register table (`preferred-register:` -- e.g. $vcc for the carry / condition operands it would like there)
    is_hint = lambda h: any("preferred-register" in ln for ln in h[2]) or any("preferred-register" in ln for ln in pre_lines[h[0]:h[1]])
    hints = [i for i, h in enumerate(H) if is_hint(h)]
    instrs = [i for i, h in enumerate(H) if not is_hint(h)]
    fmac = [i for i in instrs if any("V_FMAC_F64_e32" in ln for ln in H[i][2])]
    say("hint lines", len(hints), "| instruction lines", len(instrs), "| of them V_FMAC_F64 e64 -> e32", len(fmac))
    for label, sub in (("register-allocation hints only", hints), ("instruction rewrites only", instrs), ("V_FMAC_F64_e32 only", fmac),
                       ("everything but V_FMAC_F64_e32", [i for i in range(len(H)) if i not in set(fmac)]),
                       ("everything but the hints", instrs)):
        r = test(sub)
        say("  %-34s %5d lines: %s" % (label, len(sub), "GOOD" if r else ("BAD" if r is False else "llc failed")))
    if len(sys.argv) > 2 and sys.argv[2] == "kinds":
        return
    # ddmin: smallest subset that is still BAD
    cur = list(range(len(H)))
    n = 2
    while len(cur) >= 2:
        size = max(1, len(cur) // n)
        chunks = [cur[i:i + size] for i in range(0, len(cur), size)]
        reduced = False
        for ch in chunks:                       # a chunk alone BAD?
            r = test(ch)
            if r is False:
                cur, n, reduced = ch, 2, True
                break
        if not reduced:
            for k, ch in enumerate(chunks):    # a complement BAD?
                comp = [x for c2 in chunks[:k] + chunks[k + 1:] for x in c2]
                if not comp:
                    continue
                r = test(comp)
                if r is False:
                    cur, n, reduced = comp, max(n - 1, 2), True
                    break
        say("  tests", count[0], "subset size", len(cur), "granularity", n)
        if not reduced:
            if n >= len(cur):
                break
            n = min(len(cur), 2 * n)
    say("minimal BAD subset:", len(cur), "rewrites after", count[0], "tests")
    for i in cur[:40]:
        lo, hi, new = H[i]
        say("  line", lo + 1, "\n    -", "\n    - ".join(x.strip()[:260] for x in pre_lines[lo:hi]), "\n    +", "\n    + ".join(x.strip()[:260] for x in new))
    test(cur, keep="minimal")


if __name__ == "__main__":
    main()
