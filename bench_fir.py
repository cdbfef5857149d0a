#!/usr/bin/env python3
"""bench_fir.py -- BASELINE.json configs[0]: fir_decimate_cc (decimation 10, transition 0.05, HAMMING => 79 taps; libcsdr.c:528-549,
csdr.c:1123-1176) on 2.4 MS/s synthetic complexf streams.  The reference runs it as one process per stream on 16384-sample blocks
(that is the CPU leg below); the device batch API takes `--streams` independent streams x `--block` samples per call, inputs resident in
HBM, with the CLI's stream semantics (y[k] = sum_t h[t] x[10k+t], all k with 10k+79 <= n).

One step = one csdr_amd_fir_decimate_cc call over all streams.  Algorithmic bytes: 8 B in + 8/D B out per input sample (8.8 B).

    python bench_fir.py [--gpus N] [--steps K] [--warmup W] [--streams 256] [--block 2400256] [--decimation 10] [--tbw 0.05] [--verify]
`--decimation 50 --tbw 0.005` is the long-filter shape (801 taps) of the NFM / AM / SSB chains when they start from complexf.
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--spinup-ms", type=float, default=60.0, help=bc.SPINUP_HELP)
    ap.add_argument("--streams", type=int, default=256)
    ap.add_argument("--block", type=int, default=2344 * 1024)
    ap.add_argument("--decimation", type=int, default=10)
    ap.add_argument("--tbw", type=float, default=0.05)
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
        raise SystemExit("bench_fir.py needs an MI355X; there is no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    import csdr_amd
    ctx = csdr_amd.Context(local_rank)
    L = ctx.L
    S, T, D = args.streams, args.block, args.decimation
    nt = ctx.firdes_filter_len(args.tbw)
    taps_h = ctx.firdes_lowpass_f(nt, 0.5 / D, "HAMMING")                      # csdr.c:1144-1158
    taps = ctx.upload(taps_h)
    g = torch.Generator(device="cuda"); g.manual_seed(1234 + rank)
    pad = int(os.environ.get("CSDR_BENCH_PITCH_PAD", "0"))                       # experiments: extra samples of row pitch (channel / bank spreading of the 256 concurrent rows)
    xbuf = (torch.rand((S, T + pad, 2), device="cuda", generator=g) * 2 - 1).contiguous()
    x = xbuf
    in_pitch = T + pad
    x_ptr = x.data_ptr()
    if os.environ.get("CSDR_BENCH_ALLOC") == "contig":      # experiment: the input in physically contiguous memory (hipExtMallocWithFlags(hipDeviceMallocContiguous)) instead of torch's block
        hip = C.CDLL("libamdhip64.so")
        pc = C.c_void_p()
        nbytes = xbuf.numel() * 4
        rc = hip.hipExtMallocWithFlags(C.byref(pc), C.c_size_t(nbytes), C.c_uint(0x4))
        if rc != 0:
            raise SystemExit("hipExtMallocWithFlags(contiguous) failed: %d" % rc)
        hip.hipMemcpy(pc, C.c_void_p(xbuf.data_ptr()), C.c_size_t(nbytes), C.c_int(3))
        torch.cuda.synchronize()
        x_ptr = pc.value
    opitch = T // D + 8 + int(os.environ.get("CSDR_BENCH_OPITCH_PAD", "0"))
    y = torch.empty((S, opitch, 2), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()

    def step():
        n = L.csdr_amd_fir_decimate_cc(ctx.h, x_ptr, y.data_ptr(), S, T, in_pitch, opitch, D, taps.ptr, nt)
        if n < 0:
            raise SystemExit("fir_decimate_cc: " + ctx.err())
        return n

    spin_steps = bc.spinup(step, ctx.sync, args.spinup_ms)
    for _ in range(args.warmup):
        step()
    ctx.sync(); torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    ctx.timer_start()
    for _ in range(args.steps):
        nout = step()
    ev_ms = ctx.timer_stop_ms()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    if world > 1:
        dist.barrier()
        t = torch.tensor([wall], dtype=torch.float64, device="cuda"); dist.all_reduce(t, op=dist.ReduceOp.MAX); wall = float(t.item())
    if rank == 0:
        samples = S * T * args.steps * world
        algo = (8.0 + 8.0 / D) * S * T
        k_ms = ev_ms / args.steps                                             # one kernel per step: the HIP-event time per step IS the kernel's
        kname = L.csdr_amd_fir_last_kernel().decode()                          # what the timed calls launched (k_fir_poly, k_fir_mfma3, ...)
        res = {"metric": "complex MS/s in, fir_decimate_cc %d %g HAMMING @2.4 MS/s x N streams" % (D, args.tbw), "value": round(samples / wall / 1e6, 1),
               "unit": "complex MS/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "spinup_steps_before_warmup": spin_steps, "ms_per_step": round(wall / args.steps * 1e3, 4),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": "configs[0]: fir_decimate_cc (decim=%d, %g HAMMING, %d taps) on synthetic complexf, batched" % (D, args.tbw, nt),
                          "streams_per_gpu": S, "block_samples_per_stream": T, "decimation": D, "taps": nt, "stream_rate_sps": 2400000, "parallelism": "streams sharded, no data-path collective"},
               "roofline": {"bound": "hbm", "kernel": kname, "achieved": round(algo / (k_ms * 1e-3) / 1e9, 1), "peak": bc.HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": round(algo / (k_ms * 1e-3) / 1e9 / bc.HBM_PEAK_GBS, 4), "traffic": None, "traffic_source": None,
                            "algorithmic_bytes_per_launch": algo, "kernel_avg_ms": round(k_ms, 4), "kernel_launches_timed": args.steps},
               "outputs_per_step_per_stream": nout}
        tr = bc.pmc_traffic(kname, {"streams_per_gpu": S, "block_samples_per_stream": T, "decimation": D})
        if tr:
            res["roofline"]["traffic"], res["roofline"]["traffic_source"] = tr
        if args.verify:
            import oracle
            port = oracle.port()
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import verify_configs as vc
            rows = vc.pick_rows(S, want=6)
            worst = 0.0; ok = True
            for r in rows:
                xr = x[r].cpu().numpy().view(np.complex64).ravel()
                want = port.fir_decimate_cc(xr, D, taps_h)
                got = y[r, :nout].cpu().numpy().view(np.complex64).ravel()
                ok = ok and want.size == got.size
                worst = max(worst, vc.relrms(got[:want.size], want))
            res["verify"] = {"rows": rows, "max_rel_rms": worst, "tolerance": 1e-5, "ok": bool(ok and worst < 1e-5)}
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = bc.cpu_baseline("fir", (D, args.tbw), single_amount=100.0, probe_amount=4.0, target_wall_s=8.0,
                                                  describe="fir_decimate_cc %d %g on one 2.4 MS/s complexf stream per thread, 16384-sample blocks + refeed (csdr.c:1160-1176)" % (D, args.tbw))
        print(json.dumps(res), flush=True)
        if args.verify and not res["verify"]["ok"]:
            raise SystemExit("bench_fir.py --verify failed")
    taps.free()                                                       # device buffers go before their context
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
