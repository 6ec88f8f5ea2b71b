#!/usr/bin/env python3
"""Headline benchmark: BASELINE.json config 2 -- Lorenz-63 CDNLGSSM EKF, d_x = 3, fp64,
4096 trajectories x 1000 irregular observations PER GPU (weak scaling), all four output fields.

A "step" is one full filter sweep of the hot path over one batch of synthetic trajectories that is
already resident in HBM (time-major layout), followed by the on-device sum of the per-trajectory
log-likelihoods and -- for N > 1 GPUs -- the RCCL all-reduce of that ONE scalar, which is the only
collective the path has (independent trajectories).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line (contract in the task statement) with two extra objects:
  roofline     -- the sweep kernel against the HBM roofline: algorithmic bytes per launch
                  (B_f = 8 [(1 + m) + 2 (d + d^2)] = 224 B per trajectory-step) divided by
                  the kernel's average duration measured with HIP events on the launch stream.
  cpu_baseline -- the C restatement of the reference algorithm (oracle/cdkf_oracle.c, OpenMP over
                  trajectories) timed on this host's cores on a bounded sample of the same workload.
torch is used for device buffers, streams/events and torch.distributed only; the arithmetic is in
cd_dynamax_amd/lib/libcdkf_hip.so, called through the C ABI.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_PER_GPU = 4096
T_STEPS = 1000
D, M = 3, 3
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)


def make_batch(rank, n, T):
    """Synthetic Lorenz-63 batch: per-trajectory irregular grids with mean gap 0.005
    (recipe of the reference's simulation_utils.py:46-49), observations y = x + N(0, I) around a noisy
    Lorenz-63 path integrated with Euler-Maruyama.  Seeded per rank."""
    rng = np.random.default_rng(1234 + rank)
    u = rng.uniform(0.0, 1.0, size=(n, T))
    s = np.cumsum(u, axis=1)
    t = s / s[:, -1:] * (0.005 * T)
    x = rng.standard_normal((n, 3)) * np.sqrt(5.0)
    y = np.empty((n, T, 3))
    tc = t[:, 0].copy()
    sig, rho, beta = 10.0, 28.0, 8.0 / 3.0
    for k in range(T):
        h = (t[:, k] - tc)[:, None]
        f = np.stack([sig * (x[:, 1] - x[:, 0]), x[:, 0] * (rho - x[:, 2]) - x[:, 1], x[:, 0] * x[:, 1] - beta * x[:, 2]], 1)
        x = x + h * f + np.sqrt(h) * rng.standard_normal((n, 3))
        tc = t[:, k]
        y[:, k] = x + rng.standard_normal((n, 3))
    return t, y


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-saturation", action="store_true", help="skip the extra N=131072 (HBM-bound regime) measurement")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from cd_dynamax_amd import _ffi, distributed as D_
    import cd_dynamax_amd as cd
    from cd_dynamax_amd.models import _model_block

    rank, local_rank, world = D_.init_process_group("nccl" if args.gpus > 1 else None)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    lib = _ffi.lib()  # raises if the HIP library is missing: there is no fallback

    params = cd.ParamsCDNLGSSM(
        initial=cd.ParamsLGSSMInitial(cd.LearnableVector(np.zeros(3)), cd.LearnableMatrix(5.0 * np.eye(3))),
        dynamics=cd.ParamsCDNLGSSMDynamics(cd.LearnableLorenz63(10.0, 28.0, 8.0 / 3.0), cd.LearnableMatrix(np.eye(3)),
                                           cd.LearnableMatrix(np.eye(3)), 2.0),
        emissions=cd.ParamsCDNLGSSMEmissions(cd.LearnableLinear(np.eye(3), np.zeros(3)), cd.LearnableMatrix(np.eye(3))))
    blk = _model_block(params)
    opts = _ffi.default_opts()
    opts.layout = _ffi.LAYOUT_TCN  # engine-native layout [T, component, N]: every access coalesces
    N, T = N_PER_GPU, T_STEPS

    t_h, y_h = make_batch(rank, N, T)
    t_d = torch.from_numpy(np.ascontiguousarray(t_h.T)).to(dev)                   # [T,N]
    y_d = torch.from_numpy(np.ascontiguousarray(y_h.transpose(1, 2, 0))).to(dev)  # [T,m,N]
    ll = torch.empty(N, dtype=torch.float64, device=dev)
    fm = torch.empty(T, D, N, dtype=torch.float64, device=dev)
    fP = torch.empty(T, D, D, N, dtype=torch.float64, device=dev)
    pm = torch.empty_like(fm)
    pP = torch.empty_like(fP)
    status = torch.zeros(N, dtype=torch.int32, device=dev)
    ll_sum = torch.zeros(1, dtype=torch.float64, device=dev)
    p = lambda x: C.c_void_p(x.data_ptr())
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)  # events below are recorded on this stream

    def sweep():
        _ffi.check(lib.cdkf_ekf_filter_f64_dev(C.byref(blk.c), C.byref(opts), N, T, p(t_d), p(y_d), p(ll), p(fm), p(fP),
                                               p(pm), p(pP), p(status), stream))

    def step():
        sweep()
        _ffi.check(lib.cdkf_ll_sum_f64_dev(p(ll), N, p(ll_sum), stream))
        if world > 1:
            dist.all_reduce(ll_sum, op=dist.ReduceOp.SUM)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        elapsed = float(el.item())
    total_ll = float(ll_sum.item())
    n_bad = int((status != 0).sum().item())

    # kernel-only duration of the sweep kernel: HIP events on the launch stream, same K launches
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    for a, b in evs:
        a.record()
        sweep()
        b.record()
    torch.cuda.synchronize()
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in evs]))

    if rank == 0:
        bytes_per_launch = N * T * 8 * ((1 + M) + 2 * (D + D * D))  # 224 B per trajectory-step
        traffic, traffic_src = pmc_traffic(bytes_per_launch)
        achieved = bytes_per_launch / (kern_ms * 1e-3) / 1e9
        out = {
            "metric": "trajectories_per_sec", "value": world * N * args.steps / elapsed, "unit": "trajectories/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "Lorenz-63 CDNLGSSM EKF (state_order=second), d_x=3, d_y=3, 4096 trajectories x 1000 "
                                   "irregular obs per GPU, fp64, 4 output fields, native [T,comp,N] layout, Dopri5 dt0=0.01",
                       "trajectories_per_gpu": N, "num_timesteps": T, "parallelism": f"dp{world}"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": "filter_lpe_l63_kernel<double,OUT=all> (sixteen lanes per trajectory; the N=131072 line below runs filter_reg_kernel)",
                         "kernel_ms": kern_ms, "algorithmic_bytes_per_launch": bytes_per_launch},
            "marginal_loglik_sum": total_ll, "status_flags_raised": n_bad,
        }
        if not args.no_cpu_baseline and world == 1:
            out.update(cpu_baseline_and_error(t_h, y_h, ll, fm))
        if not args.no_saturation and world == 1:
            out["value_and_grad"] = value_and_grad(lib, blk, opts, N, T, t_d, y_d, ll, dev, torch, stream)
            del fm, fP, pm, pP
            out["saturated_regime"] = saturated(lib, blk, opts, dev, torch)
            torch.cuda.empty_cache()
            out["other_configs"] = other_configs(lib, dev, torch, t_h, y_h)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def pmc_traffic(algorithmic_bytes):
    """HBM bytes per launch of the sweep kernel from the committed rocprofv3 PMC passes of THIS command
    (profiles/*_pmc_summary.json: FETCH_SIZE x2 per MI355X_MICROARCH.md + WRITE_SIZE, separate --pmc runs).
    Counters cannot be read from inside an un-profiled run, so the number is the latest committed measurement
    for the same workload; null if there is none."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.json"))):
        try:
            rec = json.load(open(path))
        except Exception:
            continue
        if rec.get("algorithmic_bytes_per_launch") == algorithmic_bytes:
            best = (rec["hbm_traffic_bytes_per_launch"], os.path.relpath(path, ROOT))
    return best if best else (None, None)


def value_and_grad(lib, blk, opts, N, T, t_d, y_d, ll, dev, torch, stream, reps=5):
    """The SGD objective on the same resident batch: log-likelihood and its gradient w.r.t. (sigma, rho, beta) in one
    sweep (cdkf_ekf_loglik_grad_f64_dev: forward sensitivities, a lane per (trajectory, parameter)).  Informational --
    the headline metric stays the filter sweep."""
    from cd_dynamax_amd import _ffi
    grad = torch.empty(N, 3, dtype=torch.float64, device=dev)
    st = torch.zeros(N, dtype=torch.int32, device=dev)
    p = lambda x: C.c_void_p(x.data_ptr())
    run = lambda: _ffi.check(lib.cdkf_ekf_loglik_grad_f64_dev(C.byref(blk.c), C.byref(opts), N, T, p(t_d), p(y_d), p(ll),
                                                              p(grad), p(st), stream))
    run()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        a.record()
        run()
        b.record()
    torch.cuda.synchronize()
    ms = float(np.mean([a.elapsed_time(b) for a, b in evs]))
    return {"workload": "same batch: marginal log-likelihood + d/d(sigma, rho, beta) per trajectory, fp64",
            "kernel": "ekf_grad_reg_kernel<double,3,3,DriftLorenz63>", "kernel_ms": ms,
            "trajectories_per_sec": N / (ms * 1e-3), "grad_sum": [float(v) for v in grad.sum(0).tolist()]}


def saturated(lib, blk, opts, dev, torch, n=131072, reps=5):
    """Same per-trajectory workload, 32x the trajectories (lane-per-trajectory kernel, two wavefronts per SIMD; 29 GB
    per sweep): the regime where the sweep is bounded by HBM rather than by the T-long dependency chain per wavefront
    (scripts/n_sweep.py: 65 536: 4.9 TB/s, 131 072: 5.2 TB/s, 262 144: 4.8 TB/s).  Reported beside the headline number, never instead of it."""
    from cd_dynamax_amd import _ffi
    T = T_STEPS
    t_h, y_h = make_batch(99, 4096, T)
    reps_n = n // 4096
    t_d = torch.from_numpy(np.ascontiguousarray(t_h.T)).to(dev).repeat(1, reps_n)
    y_d = torch.from_numpy(np.ascontiguousarray(y_h.transpose(1, 2, 0))).to(dev).repeat(1, 1, reps_n)
    f64 = dict(dtype=torch.float64, device=dev)
    ll, st = torch.empty(n, **f64), torch.zeros(n, dtype=torch.int32, device=dev)
    fm, fP = torch.empty(T, D, n, **f64), torch.empty(T, D, D, n, **f64)
    pm, pP = torch.empty_like(fm), torch.empty_like(fP)
    p = lambda x: C.c_void_p(x.data_ptr())
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    run = lambda: _ffi.check(lib.cdkf_ekf_filter_f64_dev(C.byref(blk.c), C.byref(opts), n, T, p(t_d), p(y_d), p(ll), p(fm),
                                                         p(fP), p(pm), p(pP), p(st), stream))
    run()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        a.record()
        run()
        b.record()
    torch.cuda.synchronize()
    ms = float(np.mean([a.elapsed_time(b) for a, b in evs]))
    gbs = n * T * 224 / (ms * 1e-3) / 1e9
    return {"trajectories": n, "num_timesteps": T, "kernel_ms": ms, "trajectories_per_sec": n / (ms * 1e-3),
            "achieved_GBps": gbs, "frac_of_hbm_peak": gbs / HBM_PEAK_GBS}


def other_configs(lib, dev, torch, t_h, y_h):
    """Informational: the other BASELINE.json configurations (their per-GPU slices where the config spans 8 GPUs) through
    the same C ABI, timed with HIP events on the launch stream -- device-resident inputs, native layouts, synthetic data
    of SURVEY.md section 8d.  Not part of the headline metric."""
    import cd_dynamax_amd as cd
    from cd_dynamax_amd import _ffi
    from cd_dynamax_amd.models import _model_block
    p = lambda x: None if x is None else C.c_void_p(x.data_ptr())
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def timed(run, reps=3):
        run()
        torch.cuda.synchronize()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for a, b in evs:
            a.record()
            run()
            b.record()
        torch.cuda.synchronize()
        return float(np.mean([a.elapsed_time(b) for a, b in evs]))

    def grids(rng, n, T):
        u = rng.uniform(0.0, 1.0, size=(n, T))
        s = np.cumsum(u, axis=1)
        return s / s[:, -1:] * (0.005 * T)

    def case(params, t, y, dtype, layout, algos, outputs=True, grad=False, state_order=2):
        blk = _model_block(params)
        opts = _ffi.default_opts()
        opts.layout = layout
        opts.state_order = state_order
        n, T, m = y.shape
        d = blk.state_dim
        tdt = torch.float64 if dtype == "f64" else torch.float32
        t_d = torch.from_numpy(np.ascontiguousarray(t.T)).to(dev, tdt)
        y_d = torch.from_numpy(np.ascontiguousarray(y.transpose(1, 2, 0) if layout == _ffi.LAYOUT_TCN else y.transpose(1, 0, 2))).to(dev, tdt)
        ll = torch.empty(n, dtype=tdt, device=dev)
        st = torch.zeros(n, dtype=torch.int32, device=dev)
        bufs = [torch.empty(n * T * w, dtype=tdt, device=dev) if outputs else None for w in (d, d * d, d, d * d)]
        res = {}
        for algo in algos:
            fn = getattr(lib, f"cdkf_{algo}_{dtype}_dev")
            res[algo + "_ms"] = timed(lambda: _ffi.check(fn(C.byref(blk.c), C.byref(opts), n, T, p(t_d), p(y_d), p(ll),
                                                            *[p(b) for b in bufs], p(st), stream)))
        if grad:
            del bufs
            g = torch.empty(n, blk.theta.size, dtype=tdt, device=dev)
            gm = torch.empty(n, _ffi.model_grad_size(d, m), dtype=tdt, device=dev)
            fn = getattr(lib, f"cdkf_ekf_loglik_grad_all_{dtype}_dev")
            opts.layout = _ffi.LAYOUT_TCN
            y_g = torch.from_numpy(np.ascontiguousarray(y.transpose(1, 2, 0))).to(dev, tdt)
            res["loglik_and_grad_all_ms"] = timed(lambda: _ffi.check(fn(C.byref(blk.c), C.byref(opts), n, T, p(t_d), p(y_g), p(ll),
                                                                        p(g), p(gm), p(st), stream)))
        res["status_flags_raised"] = int((st != 0).sum().item())
        return res

    eye = np.eye
    out = {}
    l63 = cd.ParamsCDNLGSSM(
        initial=cd.ParamsLGSSMInitial(cd.LearnableVector(np.zeros(3)), cd.LearnableMatrix(5.0 * eye(3))),
        dynamics=cd.ParamsCDNLGSSMDynamics(cd.LearnableLorenz63(10.0, 28.0, 8.0 / 3.0), cd.LearnableMatrix(eye(3)), cd.LearnableMatrix(eye(3)), 2.0),
        emissions=cd.ParamsCDNLGSSMEmissions(cd.LearnableLinear(eye(3), np.zeros(3)), cd.LearnableMatrix(eye(3))))
    out["config3_lorenz63_ukf_fp32_4096x1000"] = case(l63, t_h, y_h, "f32", _ffi.LAYOUT_TCN, ["ukf_filter"])
    out["config2_with_smoother_fp64_4096x1000"] = case(l63, t_h, y_h, "f64", _ffi.LAYOUT_TCN, ["ekf_smoother"])
    rng = np.random.default_rng(1)
    d = 40
    l96 = cd.ParamsCDNLGSSM(
        initial=cd.ParamsLGSSMInitial(cd.LearnableVector(8.0 * np.ones(d)), cd.LearnableMatrix(eye(d))),
        dynamics=cd.ParamsCDNLGSSMDynamics(cd.LearnableLorenz96(8.0), cd.LearnableMatrix(eye(d)), cd.LearnableMatrix(eye(d)), 2.0),
        emissions=cd.ParamsCDNLGSSMEmissions(cd.LearnableLinear(eye(d), np.zeros(d)), cd.LearnableMatrix(eye(d))))
    n, T = 2048, 500
    out["config4_slice_lorenz96_d40_fp64_2048x500"] = case(l96, grids(rng, n, T), 8.0 + rng.standard_normal((n, T, d)), "f64",
                                                          _ffi.LAYOUT_TN, ["ekf_filter", "ekf_smoother"])
    rng = np.random.default_rng(2)
    d, m, h = 8, 4, 64
    mlp = cd.LearnableMLP(rng.standard_normal((h, d)) / np.sqrt(d), 0.1 * rng.standard_normal(h),
                          rng.standard_normal((h, h)) / np.sqrt(h), 0.1 * rng.standard_normal(h),
                          rng.standard_normal((d, h)) / np.sqrt(h), 0.1 * rng.standard_normal(d))
    c5 = cd.ParamsCDNLGSSM(
        initial=cd.ParamsLGSSMInitial(cd.LearnableVector(np.zeros(d)), cd.LearnableMatrix(eye(d))),
        dynamics=cd.ParamsCDNLGSSMDynamics(mlp, cd.LearnableMatrix(eye(d)), cd.LearnableMatrix(0.5 * eye(d)), 1.0),
        emissions=cd.ParamsCDNLGSSMEmissions(cd.LearnableLinear(eye(d)[:m], np.zeros(m)), cd.LearnableMatrix(0.5 * eye(m))))
    n, T = 1024, 1000
    # state_order 'first': the order the reverse (gradient) sweep supports for the MLP drift
    out["config5_slice_mlp_d8_fp64_1024x1000_first_order"] = case(c5, grids(rng, n, T), rng.standard_normal((n, T, m)), "f64",
                                                                 _ffi.LAYOUT_TN, ["ekf_filter"], outputs=False, grad=True, state_order=1)
    return out


def cpu_baseline_and_error(t_h, y_h, ll_dev, fm_dev):
    """Time the C port of the reference algorithm on a bounded sample of the same batch (all host cores),
    and report the HIP path's error against it on that sample (the 'marginal-LL error' of the metric)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import cdkf_oracle as o
    import cdkf_oracle_c as oc
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    ns = t_h.shape[0]
    mdl = o.lorenz63_model(3)
    oc.ekf_filter(mdl, t_h[:8], y_h[:8], nthreads=cores)  # build / warm-up
    # pick the thread count that serves the CPU best (containers often expose more CPUs than their quota)
    cand = sorted({c for c in (cores, cores // 2, cores // 4, 64, 32, 16, 8) if 1 <= c <= cores}, reverse=True)
    best = min(cand, key=lambda c: min(oc.ekf_filter(mdl, t_h[:ns], y_h[:ns], nthreads=c)["_seconds"] for _ in range(2)))
    cores = best
    ref = oc.ekf_filter(mdl, t_h[:ns], y_h[:ns], nthreads=cores)
    # bounded sample: repeat the 4096 x 1000 batch until ~3 s of wall time have been spent (>= 2 passes)
    reps = int(max(2, min(200, np.ceil(3.0 / max(ref["_seconds"], 1e-4)))))
    el = sum(oc.ekf_filter(mdl, t_h[:ns], y_h[:ns], nthreads=cores)["_seconds"] for _ in range(reps)) / reps
    ll = ll_dev[:ns].cpu().numpy()
    fm = fm_dev[:, :, :ns].cpu().numpy().transpose(2, 0, 1)
    return {
        "cpu_baseline": {"value": ns / el, "unit": "trajectories/s", "cores": cores, "kind": "port",
                         "sample": f"the same {ns} trajectories x 1000 steps batch, fp64, all four outputs written, C/OpenMP "
                                   f"restatement of the reference EKF (oracle/cdkf_oracle.c); mean of {reps} passes, "
                                   f"{el:.3f} s wall each on {cores} threads"},
        "marginal_ll_max_rel_err": float(np.max(np.abs(ll - ref["marginal_loglik"]) / np.abs(ref["marginal_loglik"]))),
        "filtered_mean_max_rel_err": float(np.max(np.abs(fm - ref["filtered_means"])) / np.max(np.abs(ref["filtered_means"]))),
    }


if __name__ == "__main__":
    main()
