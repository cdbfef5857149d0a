#!/usr/bin/env python3
"""bench_fftfilt.py -- BASELINE.json configs[2]: bandpass_fir_fft_cc overlap-add at a fixed 65536-point transform, taps swept 63 -> 4095
(apply_fir_fft_cc libcsdr.c:814-849 inside the CLI loop csdr.c:1846-1880; taps = firdes_bandpass_c(taps, -0.1, 0.2, HAMMING)).

One step = one csdr_amd_fftfilt_process call: `--streams` independent complexf streams x `--blocks` blocks of input_size = 65537 - taps
samples, inputs resident in HBM, overlap carried inside the object.  Algorithmic bytes = 16 B per input sample (8 in + 8 out; taps_fft is
cache resident).  The headline `value` is at `--taps` (default 1023); `sweep` holds the other tap counts.

The interface keeps the reference's framing (blocks of 65537 - taps samples, state carried); behind it the library serves filters of <= 4096 taps with
ONE pass over HBM (fftfilt_lds.hip: the same linear convolution by overlap-save with 4096 / 8192 / 16384-point transforms that fit a CU's LDS -- a 65536-point
transform cannot, and needs three passes).  The line says which path ran (`config.method`, `roofline.kernel`); `full_size_transform` is the same step
through the literal 65536-point block transform (three passes, fft64k.hip; `--method full` makes that one the headline).

    python bench_fftfilt.py [--gpus N] [--steps K] [--warmup W] [--streams 64] [--blocks 16] [--taps 1023] [--method auto|full] [--no-sweep] [--verify]
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import bench_common as bc  # noqa: E402

SWEEP = [63, 127, 255, 511, 1023, 2047, 4095]
FFT = 65536


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--streams", type=int, default=64)
    ap.add_argument("--blocks", type=int, default=16)
    ap.add_argument("--taps", type=int, default=1023)
    ap.add_argument("--method", choices=["auto", "full"], default="auto")
    ap.add_argument("--no-sweep", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--verify", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0")); local_rank = int(os.environ.get("LOCAL_RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench_fftfilt.py needs an MI355X; there is no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    import csdr_amd
    ctx = csdr_amd.Context(local_rank)
    L = ctx.L
    S, nb = args.streams, args.blocks
    g = torch.Generator(device="cuda"); g.manual_seed(3 + rank)
    x = (torch.rand((S, nb * FFT, 2), device="cuda", generator=g) * 2 - 1).contiguous()      # rows hold nb*input_size samples (pitch nb*FFT)
    y = torch.empty((S, nb * FFT, 2), dtype=torch.float32, device="cuda")
    pitch = nb * FFT
    torch.cuda.synchronize()

    def run(ntaps, steps, warmup, verify=False, method="auto"):
        taps = ctx.firdes_bandpass_c(ntaps, -0.1, 0.2)
        if method == "full":
            os.environ["CSDR_AMD_FFTFILT_LDS_OFF"] = "1"                # read by csdr_amd_fftfilt_create
        f = L.csdr_amd_fftfilt_create(ctx.h, FFT, taps.ctypes.data_as(C.c_void_p), ntaps, S, nb)
        os.environ.pop("CSDR_AMD_FFTFILT_LDS_OFF", None)
        if not f:
            raise SystemExit("fftfilt_create: " + ctx.err())
        inp = L.csdr_amd_fftfilt_input_size(f)
        kname = L.csdr_amd_fftfilt_kernel_name(f).decode() or "k_f64_cols_fwd + k_f64_rows + k_f64_cols_inv_oa (whole call; split in profiles/)"
        window = L.csdr_amd_fftfilt_window(f)

        def step():
            rc = L.csdr_amd_fftfilt_process(f, x.data_ptr(), y.data_ptr(), nb, pitch, pitch)
            if rc < 0:
                raise SystemExit("fftfilt_process: " + ctx.err())
        for _ in range(warmup):
            step()
        ctx.sync(); torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        ctx.timer_start()
        for _ in range(steps):
            step()
        ev_ms = ctx.timer_stop_ms()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        if world > 1:
            dist.barrier()
            t = torch.tensor([wall], dtype=torch.float64, device="cuda"); dist.all_reduce(t, op=dist.ReduceOp.MAX); wall = float(t.item())
        ver = None
        if verify and rank == 0:
            import oracle
            port = oracle.port()
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import verify_configs as vc
            L.csdr_amd_fftfilt_reset(f)
            step(); ctx.sync()
            rows = vc.pick_rows(S, want=3)
            worst = 0.0
            for r in rows:
                xr = x[r, :nb * inp].cpu().numpy().view(np.complex64).ravel()
                want = port.bandpass_fir_fft_cc(xr, taps, FFT)
                got = y[r, :nb * inp].cpu().numpy().view(np.complex64).ravel()
                worst = max(worst, vc.relrms(got[:want.size], want))
            ver = {"rows": rows, "blocks": nb, "max_rel_rms": worst, "tolerance": 1e-5, "ok": bool(worst < 1e-5)}
        L.csdr_amd_fftfilt_destroy(f)
        return inp, wall, ev_ms, ver, kname, window

    inp, wall, ev_ms, ver, kname, window = run(args.taps, args.steps, args.warmup, args.verify, args.method)
    sweep = []
    other = None
    k4 = max(args.steps // 4, 5)
    if not args.no_sweep:
        for nt in SWEEP:
            if nt == args.taps:
                continue
            i2, w2, e2, _, kn2, win2 = run(nt, k4, 2, method=args.method)
            if rank == 0:
                sweep.append({"taps": nt, "input_size": i2, "window": win2 or FFT, "value": round(S * nb * i2 * k4 * world / w2 / 1e6, 1),
                              "frac": round(16.0 * S * nb * i2 / (e2 / k4 * 1e-3) / 1e9 / bc.HBM_PEAK_GBS, 4)})
        if args.method == "auto" and window:
            i2, w2, e2, v2, kn2, _ = run(args.taps, k4, 2, args.verify, "full")
            if rank == 0:
                other = {"kernel": kn2, "value": round(S * nb * i2 * k4 * world / w2 / 1e6, 1), "ms_per_step": round(w2 / k4 * 1e3, 4),
                         "frac": round(16.0 * S * nb * i2 / (e2 / k4 * 1e-3) / 1e9 / bc.HBM_PEAK_GBS, 4)}
                if v2 is not None:
                    other["verify"] = v2
    if rank == 0:
        samples = S * nb * inp * args.steps * world
        algo = 16.0 * S * nb * inp
        k_ms = ev_ms / args.steps
        res = {"metric": "complex MS/s in->out, bandpass_fir_fft_cc overlap-add @65536-point FFT", "value": round(samples / wall / 1e6, 1), "unit": "complex MS/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(wall / args.steps * 1e3, 4),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": "configs[2]: bandpass_fir_fft_cc overlap-add, fft 65536, taps %d (firdes_bandpass_c -0.1 0.2 HAMMING)" % args.taps,
                          "streams_per_gpu": S, "blocks_per_step": nb, "input_size": inp, "taps": args.taps,
                          "method": ("one pass: overlap-save windows of %d points in LDS behind the 65536-point framing" % window) if window else "65536-point transform, three passes",
                          "parallelism": "streams sharded, no data-path collective"},
               "roofline": {"bound": "hbm", "kernel": kname,
                            "achieved": round(algo / (k_ms * 1e-3) / 1e9, 1), "peak": bc.HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": round(algo / (k_ms * 1e-3) / 1e9 / bc.HBM_PEAK_GBS, 4), "traffic": None, "traffic_source": None,
                            "algorithmic_bytes_per_launch": algo, "kernel_avg_ms": round(k_ms, 4), "kernel_launches_timed": args.steps},
               "sweep": sweep}
        if other is not None:
            res["full_size_transform"] = other
        tr = bc.pmc_traffic(kname.split("<")[0] if window else "k_f64", {"streams_per_gpu": S, "blocks_per_step": nb, "taps": args.taps})
        if tr:
            res["roofline"]["traffic"], res["roofline"]["traffic_source"] = tr
        if ver is not None:
            res["verify"] = ver
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = bc.cpu_baseline("fftfilt", (args.taps,), single_amount=100, probe_amount=20, target_wall_s=8.0, fast_fft=True,
                                                  describe="the bandpass_fir_fft_cc loop (csdr.c:1846-1880) at fft 65536, taps %d, one complexf stream per thread" % args.taps)
        print(json.dumps(res))
        if ver is not None and not ver["ok"]:
            raise SystemExit("bench_fftfilt.py --verify failed")
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
