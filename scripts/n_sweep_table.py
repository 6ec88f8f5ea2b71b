"""Throughput against the batch size N for the Lorenz-63 sweeps, on BOTH mappings (sixteen lanes per trajectory / one lane per
trajectory), to place the dispatcher's crossover (cdkf_lpe_kernels.h: lpe_batch_is_small) and to fill DESIGN.md section 3.1c.

    python3 scripts/n_sweep_table.py [out.json]          (GPU box)

Each mapping runs in a child process (CDKF_LPE_MAX_N=0: never the grid, =huge: always the grid, unset: the dispatcher's choice);
per N and sweep: ms per call (one HIP-event pair around back-to-back launches), trajectories/s, fraction of the 8 TB/s roofline
by SURVEY 8d's algorithmic bytes, and the kernel the library reports."""
import ctypes as C, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]

NS = [2048, 4096, 5120, 6144, 8192, 12288, 16384, 24576, 32768, 65536, 131072]
T = 1000


def child():
    import numpy as np
    import bench
    import cd_dynamax_amd as cd
    from cd_dynamax_amd import _ffi
    from cd_dynamax_amd.models import _model_block
    from cd_dynamax_amd._ffi import DeviceArray
    lib = _ffi.lib()
    stream = C.c_void_p()
    _ffi.check(lib.cdkf_stream_create(C.byref(stream)))
    timer = bench.Timer(lib, _ffi, stream)
    blk = _model_block(bench.l63_params(cd))
    t_h, y_h = bench.make_batch(0, 4096, T)
    rows = []
    for N in [int(x) for x in os.environ.get("NSWEEP_NS", ",".join(map(str, NS))).split(",")]:
        rep = (N + 4095) // 4096
        tt = np.tile(np.ascontiguousarray(t_h.T), (1, rep))[:, :N]
        yy = np.tile(np.ascontiguousarray(y_h.transpose(1, 2, 0)), (1, 1, rep))[:, :, :N]
        reps = 8 if N <= 16384 else 3
        for sweep in ("ekf_filter_f64", "ukf_filter_f32", "ekf_smoother_f64", "loglik_grad_f64", "loglik_grad_all_f64"):
            npd = np.float32 if sweep.endswith("f32") else np.float64
            s = npd().itemsize
            opts = _ffi.default_opts()
            opts.layout, opts.layout_in = _ffi.LAYOUT_TN, _ffi.LAYOUT_TCN
            t_d = DeviceArray.from_numpy(np.ascontiguousarray(tt, dtype=npd))
            y_d = DeviceArray.from_numpy(np.ascontiguousarray(yy, dtype=npd))
            ll = DeviceArray((N,), npd)
            st = DeviceArray.from_numpy(np.zeros(N, np.int32))
            bufs = []
            if "grad" in sweep:
                opts.layout, opts.layout_in = _ffi.LAYOUT_TCN, _ffi.LAYOUT_SAME
                g = DeviceArray((N, 3), npd)
                bufs = [g]
                if "all" in sweep:
                    gm = DeviceArray((N, _ffi.model_grad_size(3, 3)), npd)
                    bufs.append(gm)
                    run = lambda: _ffi.check(lib.cdkf_ekf_loglik_grad_all_f64_dev(C.byref(blk.c), C.byref(opts), N, T, t_d.ptr, y_d.ptr,
                                                                                  ll.ptr, g.ptr, gm.ptr, st.ptr, stream))
                else:
                    run = lambda: _ffi.check(lib.cdkf_ekf_loglik_grad_f64_dev(C.byref(blk.c), C.byref(opts), N, T, t_d.ptr, y_d.ptr,
                                                                              ll.ptr, g.ptr, st.ptr, stream))
                alg_bytes = s * (1 + 3)  # t + y read; the moments stay in the workspace
            else:
                bufs = [DeviceArray((T * N * w,), npd) for w in (3, 9, 3, 9)]
                fn = getattr(lib, f"cdkf_{sweep}_dev")
                run = lambda: _ffi.check(fn(C.byref(blk.c), C.byref(opts), N, T, t_d.ptr, y_d.ptr, ll.ptr, *[b.ptr for b in bufs],
                                            st.ptr, stream))
                alg_bytes = s * ((1 + 3) + 2 * (3 + 9)) + (s * (1 + 2 * (3 + 9)) if "smoother" in sweep else 0)
            try:
                ms = timer.ms_per_call(run, reps)
                kern = lib.cdkf_last_kernel().decode()
                bad = int(np.count_nonzero(st.numpy()))
            except Exception as e:  # (out of workspace at the largest sizes of the gradient)
                ms, kern, bad = None, f"error: {e}", -1
            rows.append({"N": N, "sweep": sweep, "ms": ms, "kernel": kern, "status_flags": bad,
                         "traj_per_s": None if ms is None else N / (ms * 1e-3),
                         "hbm_frac": None if ms is None else alg_bytes * N * T / (ms * 1e-3) / 8e12})
            for a in [t_d, y_d, ll, st] + bufs:
                a.free()
    print("NSWEEP_JSON " + json.dumps(rows), flush=True)


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "n_sweep.json")
    table = {}
    for name, env in (("dispatcher", None), ("grid_16_lanes_per_trajectory", "1000000000"), ("lane_per_trajectory", "0")):
        e = dict(os.environ)
        e.pop("CDKF_LPE_MAX_N", None)
        if env is not None:
            e["CDKF_LPE_MAX_N"] = env
        e["NSWEEP_CHILD"] = "1"
        r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=e, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("NSWEEP_JSON ")]
        if not line:
            print(name, "failed:", r.stderr[-2000:])
            continue
        table[name] = json.loads(line[0][len("NSWEEP_JSON "):])
    json.dump(table, open(out, "w"), indent=1)
    # compact view
    for sweep in ("ekf_filter_f64", "ukf_filter_f32", "ekf_smoother_f64", "loglik_grad_f64", "loglik_grad_all_f64"):
        print(sweep)
        for N in NS:
            cell = lambda v: next((f"{r['ms']:.3f}" if r["ms"] else "fail" for r in table.get(v, []) if r["N"] == N and r["sweep"] == sweep), "-")
            print(f"  N={N:7d}  grid {cell('grid_16_lanes_per_trajectory'):>8s}  lane {cell('lane_per_trajectory'):>8s}  dispatcher {cell('dispatcher'):>8s} ms")


if __name__ == "__main__":
    child() if os.environ.get("NSWEEP_CHILD") else main()
