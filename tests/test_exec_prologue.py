"""The ROCm 7.2 register-allocation defect behind rounds 3 - 5's optimisation-level-dependent wrong results, as a checkable SHAPE in the
machine code (NOTES.md R5.1, profiles/r05_j_root_cause.txt): vector spill code / VGPR -> AGPR copies at the top of a flow or join block,
IN FRONT of the `s_or_saveexec_b64` / `s_or_b64 exec, exec, sX` that re-enables the other arm's lanes.  Two implementations of one rule:
scripts/check_exec_prologue.py over llvm-objdump listings, and the library's own (launch_custom.hip: rtc_exec_prologue_defect, comgr's
disassembler) that decides whether a run-time compiled kernel is kept at -O3.  No GPU needed."""
import ctypes as C
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import check_exec_prologue as chk  # noqa: E402
from cd_dynamax_amd import _ffi  # noqa: E402

CACHE = os.path.join(ROOT, "cd_dynamax_amd", "lib", "rtc_cache")

WRONG = """0000000000001000 <k>:
\ts_and_saveexec_b64 s[2:3], s[4:5]                          // 000000001000: BE822004
\ts_xor_b64 s[4:5], exec, s[2:3]                             // 000000001004: 8884027E
\ts_cbranch_execz 3                                          // 000000001008: BF880003 <k+0x18>
\tv_mul_f64 v[16:17], v[20:21], s[48:49]                     // 00000000100C: D2810110 00006114
\tv_rndne_f64_e32 v[16:17], v[16:17]                         // 000000001014: 7E203310
\tscratch_store_dwordx2 off, v[198:199], off offset:56       // 000000001018: DC744038 007FC600
\ts_or_saveexec_b64 s[2:3], s[4:5]                           // 000000001020: BE822104
\ts_xor_b64 exec, exec, s[2:3]                               // 000000001024: 88FE027E
\ts_endpgm                                                   // 000000001028: BF810000
"""
RIGHT = WRONG.replace("\tv_rndne_f64_e32 v[16:17], v[16:17]                         // 000000001014: 7E203310\n\tscratch_store_dwordx2 off, v[198:199], off offset:56       // 000000001018: DC744038 007FC600\n",
                      "\tscratch_store_dwordx2 off, v[198:199], off offset:56       // 000000001014: DC744038 007FC600\n\ts_nop 0                                                    // 00000000101C: BF800000\n")


def test_the_rule_on_planted_listings():
    """The store behind the skip's landing point and in front of the restore is flagged; the same store at the END of the then-arm (the
    landing point is the restore itself) is not."""
    bad = chk.violations(WRONG)
    assert len(bad) == 1 and "scratch_store_dwordx2" in bad[0][2][0], bad
    right = RIGHT.replace("s_cbranch_execz 3 ", "s_cbranch_execz 5 ").replace("<k+0x18>", "<k+0x20>")
    assert chk.violations(right) == []


def test_library_objects_do_not_show_the_shape():
    objs = sorted(glob.glob(os.path.join(ROOT, "build", "csrc", "launch_*.o")))
    if not objs or shutil.which("objcopy") is None or not os.path.exists(os.path.join(chk.LLVM, "llvm-objdump")):
        pytest.skip("needs the library's object files (make -C cd_dynamax_amd/csrc) and the ROCm LLVM tools")
    with tempfile.TemporaryDirectory() as tmp:
        for o in objs:
            v = chk.violations(chk.disassemble(o, tmp))
            assert not v, (os.path.basename(o), v[:2])   # (launch_wg8.o is the -O1 build: its -O3 build shows it twice, profiles/r05_j_root_cause.txt)


def test_cache_objects_both_implementations_agree_and_only_forced_o3_builds_show_it():
    """Every code object of the in-tree hipRTC cache through the library's detector and through the script: same verdicts; what is flagged
    is exactly what the MANIFEST records as a FORCED -O3 build past the spill limit (the canary of tests/test_gpu_toolchain.py: the
    known-wrong kernel, kept on purpose) -- the shipped policy would have rebuilt it at -O1."""
    files = sorted(glob.glob(os.path.join(CACHE, "*.co")))
    manifest = os.path.join(CACHE, "MANIFEST")
    if not files or not os.path.exists(manifest) or not os.path.exists(os.path.join(chk.LLVM, "llvm-objdump")):
        pytest.skip("no run-time compiled code objects in the tree")
    L = _ffi.lib()
    L.cdkf_debug_exec_prologue_check.argtypes = [C.c_void_p, C.c_int64, C.c_char_p, C.c_char_p, C.c_int64]
    what = {ln.split()[0]: ln for ln in open(manifest)}
    flagged, checked_by_script = [], 0
    with tempfile.TemporaryDirectory() as tmp:
        for path in files:
            raw = open(path, "rb").read()
            if raw[:4] != b"CKRC":
                continue
            import struct
            magic, nname, lo, hi = struct.unpack("<4I", raw[:16])
            code = raw[16 + nname:16 + nname + (hi << 32 | lo)]
            buf = C.create_string_buffer(256)
            rc = L.cdkf_debug_exec_prologue_check(code, len(code), b"gfx950", buf, 256)
            assert rc in (0, 1), (path, rc)     # (-1: no disassembler in the process -- the policy would then fall back to the spill rule alone)
            if rc or len(flagged) + checked_by_script < 24 or hash(path) % 6 == 0:   # (llvm-objdump on every flagged object and on a sample of the rest)
                checked_by_script += 1
                script = bool(chk.violations(chk.disassemble(path, tmp)))
                assert script == bool(rc), (os.path.basename(path), rc, script, buf.value)
            if rc:
                flagged.append(os.path.basename(path)[:-3])
    for key in flagged:
        m = re.search(r"algo=\d -O3 \(vgpr spills at -O3: (\d+)", what.get(key, ""))
        assert m and int(m.group(1)) > 300, (key, what.get(key))   # a forced -O3 build (CDKF_RTC_POLICY=o3), well past the limit
    assert len(flagged) <= 2, flagged


def test_policy_rebuilds_a_kernel_that_shows_the_shape(tmp_path):
    """The d = 2 forward-sensitivity kernel (the reproducer) through the shipped policy with the spill limit lifted out of the way: the -O3
    build is discarded BECAUSE of the shape in its machine code, the kept build is -O1 and clean, the MANIFEST says why."""
    code = (
        "import os, sys\n"
        "sys.path[:0] = [%r, %r, %r]\n"
        "from cd_dynamax_amd import _ffi\n"
        "from test_custom_drift import NL_F\n"
        "k = _ffi.register_custom_drift(2, 3, NL_F, None, None)\n"
        "rc = _ffi.lib().cdkf_custom_drift_compile(k, 8, 1, 3, 1, 0)\n"
        "print('RC', rc, _ffi.lib().cdkf_last_error().decode())\n" % (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")))
    cache = tmp_path / "cache"
    cache.mkdir(mode=0o755)
    env = dict(os.environ, CDKF_RTC_CACHE_DIR=str(cache), CDKF_RTC_SPILL_LIMIT="100000")
    env.pop("CDKF_RTC_POLICY", None)
    env["CDKF_RTC_CACHE"] = "1"      # (tests/conftest.py switches the cache off where no cache directory is given)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=900)
    assert "RC 0" in p.stdout, p.stdout + p.stderr
    lines = open(cache / "MANIFEST").read().splitlines()
    assert len(lines) == 1 and "algo=2 -O1" in lines[0] and "exec-prologue defect at -O3" in lines[0], lines
    with tempfile.TemporaryDirectory() as tmp:
        (obj,) = glob.glob(str(cache / "*.co"))
        assert chk.violations(chk.disassemble(obj, tmp)) == []
