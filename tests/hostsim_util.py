"""Host builds of the device templates under the CPU sanitizers (cd_dynamax_amd/csrc/hostsim/cdkf_hostsim.h): compile a kernel's
translation unit for x86-64 with -DCDKF_HOST_SIM, run it on the argument blocks the library's launcher forms (cdkf_debug_* entry
points, no GPU), and hand the outputs back for comparison with the oracle.  Test infrastructure only."""
import ctypes as C
import hashlib
import os
import shutil
import subprocess
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "cd_dynamax_amd", "csrc")
HARNESS = os.path.join(ROOT, "tests", "hostsim")
BUILD = os.path.join(ROOT, "build", "hostsim")

SANITIZERS = {
    # -ftrivial-auto-var-init=pattern under ASan: an uninitialised private array reads as 0xAA.. (a huge / NaN real), so a result that
    # depends on one differs from the oracle instead of happening to be right
    "asan": ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-ftrivial-auto-var-init=pattern"],
    "msan": ["-fsanitize=memory", "-fsanitize-memory-track-origins=2", "-fno-sanitize-recover=all"],
    "tsan": ["-fsanitize=thread"],
    "plain": [],
}


def clang():
    for c in ("/opt/rocm/lib/llvm/bin/clang++", shutil.which("amdclang++"), shutil.which("clang++")):
        if c and os.path.exists(c):
            return c
    return None


def build(harness: str, include_src: str, san: str, opt: str = "-O1", defines=()):
    """Compile tests/hostsim/<harness> with `include_src` force-included; returns the executable's path (cached by content)."""
    cc = clang()
    if cc is None:
        raise RuntimeError("no clang++ for the host build")
    os.makedirs(BUILD, exist_ok=True)
    h = hashlib.sha1()
    for p in [os.path.join(HARNESS, harness), include_src] + sorted(
            os.path.join(dp, f) for dp, _, fs in os.walk(CSRC) for f in fs if f.endswith((".h", ".inc"))):
        h.update(open(p, "rb").read())
    h.update(" ".join([san, opt] + list(defines)).encode())
    exe = os.path.join(BUILD, f"{os.path.splitext(harness)[0]}_{san}_{h.hexdigest()[:16]}")
    if os.path.exists(exe):
        return exe
    cmd = [cc, "-x", "c++", "-std=c++17", opt, "-g", "-fno-omit-frame-pointer", "-DCDKF_HOST_SIM=1", "-I", CSRC, "-include", include_src,
           "-Wno-unused-value", "-Wno-pass-failed", "-Wno-unknown-attributes", "-Wno-unused-function", "-ffp-contract=off",
           *[f"-D{d}" for d in defines], *SANITIZERS[san], os.path.join(HARNESS, harness), "-o", exe + ".tmp", "-lpthread", "-lm"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=1800)
    if out.returncode != 0:
        raise RuntimeError("host build failed:\n" + " ".join(cmd) + "\n" + out.stderr[-6000:])
    os.replace(exe + ".tmp", exe)
    return exe


def run(exe, infile, outfile, timeout=1800):
    env = dict(os.environ)
    env["ASAN_OPTIONS"] = "detect_leaks=0:halt_on_error=1:detect_stack_use_after_return=0"
    env["UBSAN_OPTIONS"] = "print_stacktrace=1:halt_on_error=1"
    env["TSAN_OPTIONS"] = "halt_on_error=0:report_signal_unsafe=0:history_size=4"
    env["MSAN_OPTIONS"] = "halt_on_error=1"
    return subprocess.run([exe, infile, outfile], capture_output=True, text=True, timeout=timeout, env=env)


def sanitizer_reports(stderr: str):
    keys = ("ERROR: AddressSanitizer", "runtime error:", "WARNING: MemorySanitizer", "WARNING: ThreadSanitizer", "ERROR: ThreadSanitizer")
    return [ln for ln in stderr.splitlines() if any(k in ln for k in keys)]


def dump_custom_source(kind: int, nbytes: int, m: int, algo: int, state_order: int = 2, emission_kind: int = 0):
    """The translation unit launch_custom.hip generates for this variant (cdkf_custom_drift_compile with CDKF_CUSTOM_DUMP; cross-compiles
    for gfx950 on the way, no GPU needed).  Returns the path of the .hip file."""
    from cd_dynamax_amd import _ffi
    d = tempfile.mkdtemp(prefix="cdkf_dump_")
    os.environ["CDKF_CUSTOM_DUMP"] = d
    try:
        rc = _ffi.lib().cdkf_custom_drift_compile(kind, nbytes, m, algo, state_order, emission_kind)
        if rc:
            raise RuntimeError(_ffi.lib().cdkf_last_error().decode())
    finally:
        del os.environ["CDKF_CUSTOM_DUMP"]
    src = [f for f in os.listdir(d) if f.endswith(".hip")]
    assert src, os.listdir(d)
    # (algo 2 writes the filter's and the smoother's unit; the caller names the one it wants by suffix)
    return d, sorted(src)


def reg_run(src_path, mdl, opts, t, y, algo, dtype, san, outs, opt="-O1", inputs=None, preload=None):
    """Run a generated register-resident kernel on the host.  t [N,T], y [N,T,m] (layout NT in, `opts.layout` out as set by the
    caller); outs: lengths (in reals) of (ll, o1, o2, o3, o4, status, sm, sP).  Returns the list of output arrays."""
    from cd_dynamax_amd import _ffi
    dtype = np.dtype(dtype)
    N, T = t.shape
    par = np.zeros(4096, dtype)
    ip = np.zeros(27, np.int64)
    n = _ffi.lib().cdkf_debug_custom_reg_blob(C.byref(mdl.c), C.byref(opts), C.c_int64(N), C.c_int64(T), algo, dtype.itemsize, par.ctypes.data_as(C.c_void_p),
                                               C.c_int64(par.nbytes), ip.ctypes.data_as(C.POINTER(C.c_int64)))
    assert n > 0, _ffi.lib().cdkf_last_error().decode()
    par = par[:n]
    head = np.zeros(16, np.int64)
    tt, yy = np.ascontiguousarray(t, dtype), np.ascontiguousarray(y, dtype)
    head[:5] = [ip[26], n, 26, tt.size, yy.size]
    head[5:13] = outs
    uu = None if inputs is None else np.ascontiguousarray(inputs, dtype)
    head[13] = 0 if uu is None else uu.size
    head[14] = 0 if preload is None else 1   # preload = (fm, fP) flat in the sweep's layout: the smoother's backward kernel reads them
    exe = build("reg_harness.cpp", src_path, san, opt)
    with tempfile.TemporaryDirectory() as d:
        fin, fout = os.path.join(d, "in.bin"), os.path.join(d, "out.bin")
        with open(fin, "wb") as f:
            f.write(head.tobytes()); f.write(par.tobytes()); f.write(ip[:26].tobytes()); f.write(tt.tobytes()); f.write(yy.tobytes())
            if uu is not None:
                f.write(uu.tobytes())
            if preload is not None:
                f.write(np.ascontiguousarray(preload[0], dtype).tobytes()); f.write(np.ascontiguousarray(preload[1], dtype).tobytes())
        res = run(exe, fin, fout)
        reports = sanitizer_reports(res.stderr)
        if res.returncode != 0 or reports:
            raise AssertionError(f"host run ({san}) rc={res.returncode}\n" + res.stderr[-8000:])
        raw = open(fout, "rb").read()
    arrs, off = [], 0
    for k, ln in enumerate(outs):
        if k == 0 and preload is not None:
            arrs.append(None)
            continue
        dt = np.dtype(np.int32) if k == 5 else dtype
        arrs.append(np.frombuffer(raw, dt, ln, off).copy() if ln else None)
        off += ln * dt.itemsize
    return arrs


def wg_run(include_src, mdl, opts, t, y, dtype, san, *, ukf=False, kind=-1, smoother=False, filtered=None, opt="-O1", timeout=3000, inputs=None):
    """Run one instantiation of the workgroup-per-trajectory kernels on the host.  include_src: the translation unit
    (tests/hostsim/wg_builtin_tu.h or a generated custom-drift source).  t [N,T], y [N,T,m]; opts.layout = TN (outputs [T,N,...]) and
    layout_in = NT are set here.  Returns dict of outputs in the reference shapes [N,T,...]."""
    from cd_dynamax_amd import _ffi
    dtype = np.dtype(dtype)
    N, T = t.shape
    d, m = mdl.state_dim, mdl.emission_dim
    opts.layout = _ffi.LAYOUT_TN
    opts.layout_in = _ffi.LAYOUT_NT
    opts.t_shared = 0
    args = np.zeros(8192, np.uint8)
    blob = np.zeros(1 << 18, dtype)
    geom = np.zeros(6, np.int64)
    n = _ffi.lib().cdkf_debug_wg_args(C.byref(mdl.c), C.byref(opts), C.c_int64(N), C.c_int64(T), dtype.itemsize, int(ukf), int(smoother), args.ctypes.data_as(C.c_void_p),
                                       C.c_int64(args.nbytes), blob.ctypes.data_as(C.c_void_p), C.c_int64(blob.nbytes), geom.ctypes.data_as(C.POINTER(C.c_int64)))
    assert n > 0, _ffi.lib().cdkf_last_error().decode()
    ept, threads, lds, asz = (int(v) for v in geom[:4])
    tt, yy = np.ascontiguousarray(t, dtype), np.ascontiguousarray(y, dtype)
    nm, nP = N * T * d, N * T * d * d
    head = np.zeros(16, np.int64)
    head[:6] = [N, threads, asz, n, tt.size, yy.size]
    head[6:14] = [N, nm, nP, 0 if smoother else nm, 0 if smoother else nP, N, nm if smoother else 0, nP if smoother else 0]
    uu = None if inputs is None else np.ascontiguousarray(inputs, dtype)
    head[14] = 0 if uu is None else uu.size
    defines = [f"HS_REAL={'double' if dtype.itemsize == 8 else 'float'}", f"HS_EPT={ept}", f"HS_UKF={int(ukf)}", f"HS_KIND={kind}",
               f"HS_SMOOTHER={int(smoother)}"]
    if not open(include_src).read().count("CDKF_WG_STATIC_LDS"):
        defines.append(f"CDKF_WG_STATIC_LDS={lds}")
    exe = build("wg_harness.cpp", include_src, san, opt, defines)
    with tempfile.TemporaryDirectory() as dd:
        fin, fout = os.path.join(dd, "in.bin"), os.path.join(dd, "out.bin")
        with open(fin, "wb") as f:
            f.write(head.tobytes()); f.write(args[:asz].tobytes()); f.write(blob[:n].tobytes()); f.write(tt.tobytes()); f.write(yy.tobytes())
            if smoother:   # filtered moments in the sweep's own layout [T,N,...]
                f.write(np.ascontiguousarray(np.swapaxes(filtered[0], 0, 1), dtype).tobytes())
                f.write(np.ascontiguousarray(np.swapaxes(filtered[1], 0, 1), dtype).tobytes())
            if uu is not None:
                f.write(uu.tobytes())
        res = run(exe, fin, fout, timeout)
        reports = sanitizer_reports(res.stderr)
        if res.returncode != 0 or reports:
            raise AssertionError(f"host run ({san}, ept {ept}, threads {threads}) rc={res.returncode}\n" + res.stderr[-12000:])
        raw = open(fout, "rb").read()
    off = 0

    def take(count, dt=dtype):
        nonlocal off
        a = np.frombuffer(raw, dt, count, off).copy()
        off += count * np.dtype(dt).itemsize
        return a
    TN = lambda a, shape: np.swapaxes(a.reshape((T, N) + shape), 0, 1)
    if smoother:
        sm, sP = take(nm), take(nP)
        return {"smoothed_means": TN(sm, (d,)), "smoothed_covariances": TN(sP, (d, d)), "status": take(N, np.int32), "geom": (ept, threads, lds)}
    ll, fm, fP, pm, pP = take(N), take(nm), take(nP), take(nm), take(nP)
    return {"marginal_loglik": ll, "filtered_means": TN(fm, (d,)), "filtered_covariances": TN(fP, (d, d)), "predicted_means": TN(pm, (d,)),
            "predicted_covariances": TN(pP, (d, d)), "status": take(N, np.int32), "geom": (ept, threads, lds)}


def awg_run(include_src, mdl, opts, t, y, dtype, san, forward, *, inputs=None, n_theta=None, opt="-O1", timeout=3000):
    """Run the reverse sweep ekf_adjoint_wg_kernel on the host.  include_src: the generated custom-drift unit of the reverse sweep (or
    a header with the library's instantiation); forward: the dict wg_run returned for the filter on the same problem (its four moment
    arrays are what the reverse sweep reads).  Returns (grad [N, n_theta], grad_model [N, d + 2 d^2 + m d + m + m^2], status)."""
    from cd_dynamax_amd import _ffi
    dtype = np.dtype(dtype)
    N, T = t.shape
    d, m = mdl.state_dim, mdl.emission_dim
    opts.layout = _ffi.LAYOUT_TN
    opts.layout_in = _ffi.LAYOUT_NT
    opts.t_shared = 0
    args = np.zeros(8192, np.uint8)
    blob = np.zeros(1 << 18, dtype)
    geom = np.zeros(6, np.int64)
    n = _ffi.lib().cdkf_debug_wg_args(C.byref(mdl.c), C.byref(opts), C.c_int64(N), C.c_int64(T), dtype.itemsize, 0, 2, args.ctypes.data_as(C.c_void_p),
                                       C.c_int64(args.nbytes), blob.ctypes.data_as(C.c_void_p), C.c_int64(blob.nbytes), geom.ctypes.data_as(C.POINTER(C.c_int64)))
    assert n > 0, _ffi.lib().cdkf_last_error().decode()
    ne, threads, lds, asz, ws_stride, cap = (int(v) for v in geom)
    n_theta = int(mdl.c.n_theta) if n_theta is None else n_theta
    ngm = d + 2 * d * d + m * d + m + m * m
    tt, yy = np.ascontiguousarray(t, dtype), np.ascontiguousarray(y, dtype)
    uu = None if inputs is None else np.ascontiguousarray(inputs, dtype)
    nm, nP = N * T * d, N * T * d * d
    head = np.zeros(16, np.int64)
    head[:13] = [N, threads, asz, n, tt.size, yy.size, nm, nP, N * n_theta, N * ngm, ws_stride, cap, 0 if uu is None else uu.size]
    defines = [f"HS_REAL={'double' if dtype.itemsize == 8 else 'float'}", f"HS_NE={ne}"]
    if not open(include_src).read().count("CDKF_WG_STATIC_LDS"):
        defines.append(f"CDKF_WG_STATIC_LDS={lds}")
    exe = build("awg_harness.cpp", include_src, san, opt, defines)
    TNf = lambda a: np.ascontiguousarray(np.swapaxes(a, 0, 1), dtype).tobytes()
    with tempfile.TemporaryDirectory() as dd:
        fin, fout = os.path.join(dd, "in.bin"), os.path.join(dd, "out.bin")
        with open(fin, "wb") as f:
            f.write(head.tobytes()); f.write(args[:asz].tobytes()); f.write(blob[:n].tobytes()); f.write(tt.tobytes()); f.write(yy.tobytes())
            for k in ("filtered_means", "filtered_covariances", "predicted_means", "predicted_covariances"):
                f.write(TNf(forward[k]))
            if uu is not None:
                f.write(uu.tobytes())
        res = run(exe, fin, fout, timeout)
        reports = sanitizer_reports(res.stderr)
        if res.returncode != 0 or reports:
            raise AssertionError(f"host run of the reverse sweep ({san}, ne {ne}) rc={res.returncode}\n" + res.stderr[-12000:])
        raw = open(fout, "rb").read()
    off = N * dtype.itemsize
    grad = np.frombuffer(raw, dtype, N * n_theta, off).reshape(N, n_theta).copy()
    off += N * n_theta * dtype.itemsize
    gm = np.frombuffer(raw, dtype, N * ngm, off).reshape(N, ngm).copy()
    off += N * ngm * dtype.itemsize
    return grad, gm, np.frombuffer(raw, np.int32, N, off).copy()


def ut_run(mdl, opts, t, y, dtype, san, *, every_leaf=True, opt="-O1", inputs=None, timeout=3000, ekf=False, value_only=False):
    """Run the tangent sweep of the literal unscented recursion (cdkf_ukf_tangent_kernels.h) on the host for this model: the
    translation unit launch_custom.hip generates (cdkf_ukf_tangent_compile with CDKF_CUSTOM_DUMP; cross-compiles for gfx950 on the way).
    ekf=True: the extended filter's sweep (cdkf_ekf_tangent_compile; opts.state_order / num_iter apply).
    t [N,T], y [N,T,m]; returns (ll [N], grad [N, n_theta], grad_model [N, .] or None, status [N]).
    value_only: the kernels' value mode (the FILTER through them, a lane per trajectory): returns (ll, status, filtered means [N,T,d],
    covariances [N,T,d,d], predicted means, covariances)."""
    from cd_dynamax_amd import _ffi
    dtype = np.dtype(dtype)
    N, T = t.shape
    d, m, nth = mdl.state_dim, mdl.emission_dim, int(mdl.c.n_theta)
    dump = tempfile.mkdtemp(prefix="cdkf_dump_")
    os.environ["CDKF_CUSTOM_DUMP"] = dump
    try:
        rc = (_ffi.lib().cdkf_ekf_tangent_compile if ekf else _ffi.lib().cdkf_ukf_tangent_compile)(C.byref(mdl.c), C.byref(opts), dtype.itemsize)
        if rc:
            raise RuntimeError(_ffi.lib().cdkf_last_error().decode())
    finally:
        del os.environ["CDKF_CUSTOM_DUMP"]
    src = [os.path.join(dump, f) for f in os.listdir(dump) if f.endswith(".hip")]
    assert len(src) == 1, os.listdir(dump)
    args = np.zeros(512, np.uint8)
    par = np.zeros(65536, dtype)
    n = _ffi.lib().cdkf_debug_ukf_tangent_args(C.byref(mdl.c), C.byref(opts), C.c_int64(N), C.c_int64(T), dtype.itemsize, 2 if value_only else (1 if every_leaf else 0),
                                                args.ctypes.data_as(C.c_void_p), C.c_int64(args.nbytes), par.ctypes.data_as(C.c_void_p), C.c_int64(par.nbytes))
    assert n > 0, _ffi.lib().cdkf_last_error().decode()
    par = par[:n]
    nargs = 8 * 8 + 8 * 11 + dtype.itemsize * 6 + 4 * 4   # 8 pointers, 11 longs, 6 reals, 4 ints ...
    nargs = (nargs + 7) // 8 * 8 + 8 * 4 + 8 * 6       # ... padded to the pointers' alignment, 4 pointers, 6 longs (value mode)
    npd, npm = d * (d + 1) // 2, m * (m + 1) // 2
    nleaf = 1 if value_only else (nth + d + 2 * npd + m * d + m + npm if every_leaf else max(nth, 1))
    gm_len = N * (d + 2 * d * d + m * d + m + m * m) if every_leaf and not value_only else 0
    tt, yy = np.ascontiguousarray(t, dtype), np.ascontiguousarray(y, dtype)
    uu = None if inputs is None else np.ascontiguousarray(inputs, dtype)
    head = np.array([(N * nleaf + 63) // 64, nargs, n, tt.size, yy.size, 0 if uu is None else uu.size, N * nth, gm_len], np.int64)
    exe = build("ut_harness.cpp", src[0], san, opt)
    with tempfile.TemporaryDirectory() as dd:
        fin, fout = os.path.join(dd, "in.bin"), os.path.join(dd, "out.bin")
        with open(fin, "wb") as f:
            f.write(head.tobytes()); f.write(args[:nargs].tobytes()); f.write(par.tobytes()); f.write(tt.tobytes()); f.write(yy.tobytes())
            if uu is not None:
                f.write(uu.tobytes())
        res = run(exe, fin, fout, timeout=timeout)
        reports = sanitizer_reports(res.stderr)
        if res.returncode != 0 or reports:
            raise AssertionError(f"host run ({san}) rc={res.returncode}\n" + res.stderr[-8000:])
        raw = open(fout, "rb").read()
    off = 0
    ll = np.frombuffer(raw, dtype, N, off).copy(); off += N * dtype.itemsize
    grad = np.frombuffer(raw, dtype, N * nth, off).reshape(N, nth).copy(); off += N * nth * dtype.itemsize
    gm = np.frombuffer(raw, dtype, gm_len, off).reshape(N, -1).copy() if gm_len else None; off += gm_len * dtype.itemsize
    status = np.frombuffer(raw, np.int32, N, off).copy(); off += N * 4
    if value_only:
        mom = []
        for k in range(4):
            cnt = N * T * d * (d if k & 1 else 1)
            mom.append(np.frombuffer(raw, dtype, cnt, off).reshape((N, T, d, d) if k & 1 else (N, T, d)).copy()); off += cnt * dtype.itemsize
        return (ll, status) + tuple(mom)
    return ll, grad, gm, status


def em_run(mdl, opts, ukf, t, mu, P, dtype, san, *, inputs=None, opt="-O1", timeout=3000):
    """Run the emission-moments kernel generated around this model's emission statements (launch_custom.hip: kEmissionMomentsKernel) on the
    host: the translation unit cdkf_custom_emission_moments_compile dumps (it cross-compiles for gfx950 on the way).
    t [rows], mu [rows, d], P [rows, d, d] or None; returns (ym [rows, m], yc [rows, m, m] or None)."""
    from cd_dynamax_amd import _ffi
    dtype = np.dtype(dtype)
    d, m = mdl.state_dim, mdl.emission_dim
    rows = mu.shape[0]
    dump = tempfile.mkdtemp(prefix="cdkf_dump_")
    os.environ["CDKF_CUSTOM_DUMP"] = dump
    try:
        rc = _ffi.lib().cdkf_custom_emission_moments_compile(C.byref(mdl.c), C.byref(opts), dtype.itemsize)
        if rc:
            raise RuntimeError(_ffi.lib().cdkf_last_error().decode())
    finally:
        del os.environ["CDKF_CUSTOM_DUMP"]
    src = [os.path.join(dump, f) for f in os.listdir(dump) if f.endswith(".hip")]
    assert len(src) == 1, os.listdir(dump)
    par = np.concatenate([np.asarray(mdl.H, np.float64).ravel(), np.asarray(mdl.h_bias, np.float64).ravel(), np.asarray(mdl.R, np.float64).ravel()]).astype(dtype)
    alpha, n = float(opts.ukf_alpha), float(d)
    lamb = alpha * alpha * (n + float(opts.ukf_kappa)) - n
    w = np.array([np.sqrt(n + lamb), lamb / (n + lamb), lamb / (n + lamb) + (1 - alpha * alpha + float(opts.ukf_beta)), 1 / (2 * (n + lamb))], dtype)
    head = np.array([rows, 1 if ukf else 0, 0 if P is None else 1, par.size], np.int64)
    exe = build("em_harness.cpp", src[0], san, opt)
    with tempfile.TemporaryDirectory() as dd:
        fin, fout = os.path.join(dd, "in.bin"), os.path.join(dd, "out.bin")
        with open(fin, "wb") as f:
            f.write(head.tobytes()); f.write(w.tobytes()); f.write(par.tobytes()); f.write(np.ascontiguousarray(t, dtype).tobytes())
            if inputs is not None:
                f.write(np.ascontiguousarray(inputs, dtype).tobytes())
            f.write(np.ascontiguousarray(mu, dtype).tobytes())
            if P is not None:
                f.write(np.ascontiguousarray(P, dtype).tobytes())
        res = run(exe, fin, fout, timeout=timeout)
        reports = sanitizer_reports(res.stderr)
        if res.returncode != 0 or reports:
            raise AssertionError(f"host run ({san}) rc={res.returncode}\n" + res.stderr[-8000:])
        raw = open(fout, "rb").read()
    ym = np.frombuffer(raw, dtype, rows * m, 0).reshape(rows, m).copy()
    yc = None if P is None else np.frombuffer(raw, dtype, rows * m * m, rows * m * dtype.itemsize).reshape(rows, m, m).copy()
    return ym, yc
