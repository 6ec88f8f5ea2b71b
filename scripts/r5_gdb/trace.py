# rocgdb Python: single-step ONE wavefront of the kernel from a start offset and log (pc offset, instruction, destination value of lane 0).
# usage inside rocgdb:  source scripts/r5_gdb/trace.py   (environment: R5_TRACE_START, R5_TRACE_STEPS, R5_TRACE_P, R5_TRACE_OUT)
import gdb, os, re, time
START = int(os.environ.get("R5_TRACE_START", "0x504"), 16)
STEPS = int(os.environ.get("R5_TRACE_STEPS", "2000"))
WANT_P = int(os.environ.get("R5_TRACE_P", "1"))
OUT = os.environ.get("R5_TRACE_OUT", "gpurun_out/r5_trace.txt")
gdb.execute("set pagination off"); gdb.execute("set confirm off"); gdb.execute("set breakpoint pending on"); gdb.execute("set height 0")
gdb.execute("break cdkf_custom_kernel")
gdb.execute("run")
gdb.execute("delete 1")
base = int(gdb.parse_and_eval("(long)&cdkf_custom_kernel"))
gdb.execute("break *%d" % (base + START))
# run to the start offset in a wavefront whose parameter index (v28 lane 0) is the wanted one
while True:
    gdb.execute("continue")
    p = int(gdb.parse_and_eval("$v28[0]"))
    if p == WANT_P:
        break
gdb.execute("delete")
gdb.execute("set scheduler-locking on")
def reg(tok):
    m = re.match(r"^v(\d+)$", tok)
    if m: return "%08x" % (int(gdb.parse_and_eval("$v%s[0]" % m.group(1))) & 0xffffffff)
    m = re.match(r"^v\[(\d+):(\d+)\]$", tok)
    if m: return ":".join("%08x" % (int(gdb.parse_and_eval("$v%d[0]" % k)) & 0xffffffff) for k in range(int(m.group(2)), int(m.group(1)) - 1, -1))
    m = re.match(r"^a(\d+)$", tok)
    if m: return "%08x" % (int(gdb.parse_and_eval("$a%s[0]" % m.group(1))) & 0xffffffff)
    m = re.match(r"^a\[(\d+):(\d+)\]$", tok)
    if m: return ":".join("%08x" % (int(gdb.parse_and_eval("$a%d[0]" % k)) & 0xffffffff) for k in range(int(m.group(2)), int(m.group(1)) - 1, -1))
    m = re.match(r"^s(\d+)$", tok)
    if m: return "%08x" % (int(gdb.parse_and_eval("$s%s" % m.group(1))) & 0xffffffff)
    m = re.match(r"^s\[(\d+):(\d+)\]$", tok)
    if m: return ":".join("%08x" % (int(gdb.parse_and_eval("$s%d" % k)) & 0xffffffff) for k in range(int(m.group(2)), int(m.group(1)) - 1, -1))
    if tok == "vcc": return "%016x" % (int(gdb.parse_and_eval("$vcc")) & 0xffffffffffffffff)
    return ""
t0 = time.time()
with open(OUT, "w") as f:
    for k in range(STEPS):
        pc = int(gdb.parse_and_eval("$pc"))
        txt = gdb.execute("x/i $pc", to_string=True)
        ins = txt.split(":", 1)[1].strip() if ":" in txt else txt.strip()
        ins = re.sub(r"^<[^>]*>\s*", "", ins)
        gdb.execute("stepi", to_string=True)
        toks = ins.replace(",", " ").split()
        dst = toks[1] if len(toks) > 1 else ""
        val = ""
        if not toks[0].startswith(("scratch_store", "global_store", "s_cbranch", "s_branch", "s_waitcnt", "s_nop", "s_setpc", "s_cmp", "v_cmp")) or dst == "vcc":
            try: val = reg(dst)
            except Exception as e: val = "?"
        f.write("%06x  %-70s %s\n" % (pc - base, ins, val))
        if k % 500 == 0:
            f.flush()
    f.write("# %d steps in %.1f s\n" % (STEPS, time.time() - t0))
gdb.execute("kill")
gdb.execute("quit")
