#!/usr/bin/env python3
"""bench_nfm.py -- BASELINE.json configs[4]: the NFM receive chain (README.md:87)
    convert_u8_f | shift_addition_cc | fir_decimate_cc 50 0.005 HAMMING | fmdemod_quadri_cf | limit_ff | deemphasis_nfm_ff 48000 | fastagc_ff | convert_f_s16
on 4096 concurrent 25 kHz channels = 512 independent 2.4 MS/s u8 IQ streams per GPU on 8 GPUs (streams sharded, no data-path collective).

One step = one pass of the chain object (csdr_amd_nfm_*) over `--block` new samples of every stream, inputs resident in HBM.  Reports whole-job
complex MS/s in -> audio out, and for the front-end kernel (k_ddc_mfma) the HIP-event time, its algorithmic bytes (2 B in + 8/50 B out per input
sample) and the fraction of the 8 TB/s HBM roofline.

    python bench_nfm.py [--gpus N] [--steps K] [--warmup W] [--streams 512] [--block 2400256]
N > 1:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench_nfm.py --gpus N ...
Not part of the driver's bench contract (that is bench.py); same timing discipline (barrier + synchronize, max over ranks).
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
HBM_PEAK_GBS = bc.HBM_PEAK_GBS


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)        # ~0.3 s of timed work
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--spinup-ms", type=float, default=60.0, help=bc.SPINUP_HELP)
    ap.add_argument("--streams", type=int, default=512)
    ap.add_argument("--block", type=int, default=2344 * 1024)
    ap.add_argument("--uniform-rate", action="store_true", help="ONE shift rate (-0.05) for all channels: the shared-weights kernel of rounds 1-3 (csdr_amd_nfm_create); default: "
                    "a rate per channel (csdr_amd_nfm_create_rates), config 5 as SURVEY.md section 8d defines it")
    ap.add_argument("--front-end-only", action="store_true", help="time csdr_amd_ddc_process alone (convert | shift | fir_decimate)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--verify", action="store_true",
                    help="after the timed region: reset, one more pass over all channels with the same object / buffers, 16 full s16 rows spread "
                         "over all channel blocks against the CPU oracle's stage-by-stage chain on the same bytes")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0")); local_rank = int(os.environ.get("LOCAL_RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench_nfm.py needs an MI355X; there is no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    import csdr_amd
    ctx = csdr_amd.Context(local_rank)
    L = ctx.L
    S, T, D = args.streams, args.block, 50
    assert T % 1024 == 0
    taps = ctx.firdes_lowpass_f(ctx.firdes_filter_len(0.005), 0.5 / D, "HAMMING")      # csdr.c:1144-1158: 801 taps
    pitch = 2 * T
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import verify_configs as vc
    rates = np.full(S, -0.05, np.float32) if args.uniform_rate else vc.c5_rates(S)
    g = torch.Generator(device="cuda"); g.manual_seed(5000 + rank)
    x = torch.randint(0, 256, (S, pitch), dtype=torch.uint8, device="cuda", generator=g)
    strict_rows = []
    if args.verify and rank == 0 and not args.front_end_only:
        # rows held to +-1 LSB on every sample: a real narrow-band FM signal in a few rows spread over the batch, put there BEFORE the timed loop
        from tests_helpers import nfm_signal_u8
        strict_rows = sorted({r for r in (5, S // 3 + 1, (2 * S) // 3 + 2, S - 2) if 0 <= r < S})      # (with a rate per channel: the four drifting rates)
        for k, r in enumerate(strict_rows):
            x[r, :2 * T] = torch.from_numpy(nfm_signal_u8(7100 + k, T, offset=-float(rates[r]))).cuda()
    n_out_max = (T // D + 2048 + 63) // 64 * 64
    out_s16 = torch.empty((S, n_out_max), dtype=torch.int16, device="cuda")
    out_y = torch.empty((S, n_out_max, 2), dtype=torch.float32, device="cuda") if args.front_end_only else None
    torch.cuda.synchronize()
    tp, rp = taps.ctypes.data_as(C.c_void_p), rates.ctypes.data_as(C.c_void_p)
    if args.front_end_only:
        obj = L.csdr_amd_ddc_create(ctx.h, S, -0.05, D, tp, taps.size, T) if args.uniform_rate else L.csdr_amd_ddc_create_rates(ctx.h, S, rp, D, tp, taps.size, T)
        fe = obj
    else:
        obj = (L.csdr_amd_nfm_create(ctx.h, S, -0.05, D, tp, taps.size, 48000, 1024, 1.0, 1.0, T) if args.uniform_rate else
               L.csdr_amd_nfm_create_rates(ctx.h, S, rp, D, tp, taps.size, 48000, 1024, 1.0, 1.0, T))
        fe = L.csdr_amd_nfm_front_end(obj) if obj else None
    if not obj:
        raise SystemExit("create: " + ctx.err())

    def step():
        if args.front_end_only:
            n = L.csdr_amd_ddc_process(obj, x.data_ptr(), pitch, T, out_y.data_ptr(), n_out_max)
        else:
            n = L.csdr_amd_nfm_process(obj, x.data_ptr(), pitch, T, out_s16.data_ptr(), None, n_out_max)
        if n < 0:
            raise SystemExit("process: " + ctx.err())
        return n

    spin_steps = bc.spinup(step, ctx.sync, args.spinup_ms)
    for _ in range(args.warmup):
        step()
    ctx.sync(); torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    L.csdr_amd_ddc_set_profiling(fe, 1)
    t0 = time.perf_counter()
    ctx.timer_start()
    produced = 0
    for _ in range(args.steps):
        produced += step()
    ev_ms = ctx.timer_stop_ms()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    if world > 1:
        dist.barrier()
        t = torch.tensor([wall], dtype=torch.float64, device="cuda"); dist.all_reduce(t, op=dist.ReduceOp.MAX); wall = float(t.item())
    kms = C.c_double(); kl = C.c_long()
    L.csdr_amd_ddc_kernel_time(fe, C.byref(kms), C.byref(kl))
    kname = L.csdr_amd_ddc_kernel_name(fe).decode()
    if rank == 0:
        samples = S * T * args.steps * world
        k_avg_ms = kms.value / max(kl.value, 1)
        # front-end kernel: 2 B of u8 IQ in per sample; out per D samples: one complexf (the stand-alone front end), or -- inside the chain object, whose reducer
        # epilogue demodulates and limits -- the three digit bytes the de-emphasis FIR reads (CSDR_AMD_NFM_FUSE=0 restores the complexf store)
        fused = (not args.front_end_only) and os.environ.get("CSDR_AMD_NFM_FUSE", "1") != "0" and kname.startswith("k_ddc_mfma")
        algo = (2.0 + (3.0 if fused else 8.0) / D) * S * T
        if fused:
            kname += " (fused fmdemod_quadri_cf | limit_ff epilogue)"
        tr = bc.pmc_traffic("k_ddc_mfma", {"channels_per_gpu": S, "block_samples_per_channel": T, "shift_rates": "uniform" if args.uniform_rate else "per channel"})
        if not tr and args.uniform_rate:                      # (summaries older than the shift_rates key are the shared-rate kernel's)
            tr = bc.pmc_traffic("k_ddc_mfma<13, 2, true>", {"channels_per_gpu": S, "block_samples_per_channel": T})
        traffic, traffic_src = tr if tr else (None, None)
        if traffic and not (0.8 < traffic / algo < 1.25):
            traffic, traffic_src = None, None                 # a summary of the other (fused / unfused) variant
        res = {"metric": "complex MS/s in->out, NFM chain @2.4 MS/s x N channels", "value": round(samples / wall / 1e6, 1), "unit": "complex MS/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "spinup_steps_before_warmup": spin_steps, "ms_per_step": round(wall / args.steps * 1e3, 4),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": "configs[4]: NFM chain u8 IQ -> s16 audio (convert_u8_f|shift_addition_cc " + ("-0.05 for all channels" if args.uniform_rate else "<a rate per channel: %d distinct, -0.4999 .. 0.4999>" % len(set(rates.tolist()))) + "|fir_decimate_cc 50 0.005 HAMMING|fmdemod_quadri_cf|"
                                      "limit_ff|deemphasis_nfm_ff 48000|fastagc_ff|convert_f_s16)" + (" -- FRONT END ONLY (first three stages)" if args.front_end_only else ""),
                          "channels_per_gpu": S, "block_samples_per_channel": T, "channel_rate_sps": 2400000, "shift_rates": "uniform" if args.uniform_rate else "per channel",
                          "fallback": bool(L.csdr_amd_ddc_fallback(fe)),
                          "realtime_channels_equivalent": round(samples / wall / 2.4e6, 1), "parallelism": "channels sharded, no data-path collective"},
               "roofline": {"bound": "hbm", "kernel": kname, "achieved": round(algo / (k_avg_ms * 1e-3) / 1e9, 1) if k_avg_ms else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": round(algo / (k_avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if k_avg_ms else None, "traffic": traffic, "traffic_source": traffic_src,
                            "algorithmic_bytes_per_launch": algo, "kernel_avg_ms": round(k_avg_ms, 4), "kernel_launches_timed": kl.value,
                            "hip_event_ms_per_step_all_kernels": round(ev_ms / args.steps, 4),
                            "whole_chain": {"algorithmic_bytes_per_step": (2.0 + 2.0 / D) * S * T,
                                            "achieved": round((2.0 + 2.0 / D) * S * T / (ev_ms / args.steps * 1e-3) / 1e9, 1),
                                            "frac": round((2.0 + 2.0 / D) * S * T / (ev_ms / args.steps * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}},
               "outputs_per_step_per_channel": produced // max(args.steps, 1)}
        if args.verify and not args.front_end_only:
            L.csdr_amd_ddc_set_profiling(fe, 0)
            res["verify"] = vc.verify_nfm(ctx, obj, x, out_s16, S, T, pitch, n_out_max, shift_rate=rates, rows=[r for r in vc.pick_rows(S) if r not in strict_rows], strict_rows=strict_rows)
        if world == 1 and not args.no_cpu_baseline and not args.front_end_only:
            res["cpu_baseline"] = bc.cpu_baseline("nfm", unit="complex MS/s", single_amount=100.0, probe_amount=4.0, target_wall_s=8.0,
                                                  describe="config 5 NFM chain (README.md:87), one 2.4 MS/s u8 IQ channel per thread, in process with the CLI's block framing")
        print(json.dumps(res))
        if "verify" in res and not res["verify"]["ok"]:
            raise SystemExit("bench_nfm.py --verify failed: %s" % json.dumps(res["verify"]))
    if args.front_end_only:
        L.csdr_amd_ddc_destroy(obj)
    else:
        L.csdr_amd_nfm_destroy(obj)
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
