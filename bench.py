#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric: complex MS/s in->out for the WFM demod chain @2.4 MS/s x N streams,
with the % of the HBM roofline of the dominant kernel.

Workload (BASELINE.json configs[1]): the full WFM pipe
    convert_u8_f | shift_addition_cc -0.085 | fir_decimate_cc 10 0.05 HAMMING | fmdemod_quadri_cf |
    fractional_decimator_ff 5 | deemphasis_wfm_ff 48000 50e-6 | convert_f_s16
on 1024 parallel 2.4 MS/s u8 IQ streams per GPU, synthetic i.i.d. uniform u8 (SURVEY.md section 8d "throughput
signal"), inputs resident in HBM before the timed region.  One step = one block of 2344*1024 = 2 400 256 complex
samples (1.0001 s of signal) of every stream through the whole chain (state carried from step to step).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--streams S] [--block T] [--no-cpu-baseline] [--verify | --no-verify] [--no-other-configs]

The line checks itself: four rows of the batch carry a real FM signal (put there BEFORE the timed loop) and are held to +-1 LSB on every sample against the CPU
oracle after it, and 16 rows of the noise batch spread over all stream blocks go through the statistical gate of tests/verify_configs.py ("verify", default;
--strict-rows-only drops the noise rows; --no-verify skips).  The exit code is non-zero when the headline OR any operating point OR any other config's leg fails its
verify or crashes ("all_legs_verified").  At N = 1 the line also
carries the other BASELINE configs as short legs of their own bench scripts ("other_configs": C1 fir_decimate_cc, C3 at 1023 and 4095 taps, C4 fastddc, C5 NFM with a
rate per channel -- each with its own verify) and short-block operating points of this chain ("operating_points": the reference's 16384-sample block -- one launch per
block, and through the resident ring csdr_amd_wfm_ring_* --, and 65536 streams x 10 ms), so that every config's number exists on the driver's clock.

N > 1: launched by torch.distributed.run, one rank per GPU; streams are independent, so ranks share nothing on
the data path (replicas of the per-GPU workload, "scaling": "weak"); barrier + max-over-ranks timing over RCCL.
Prints ONE JSON line on rank 0.

Before the W warm-up steps the bench runs untimed steps for --spinup-ms (default 60 ms, reported as "spinup_steps_before_warmup"): after idle the GPU
takes about 25 launches to reach its running clocks (the chain kernel's first launches measure 990-1175 us, then settle at ~905: profiles/r3_notes.md),
and the metric is a stream's steady-state rate.  The K timed steps are untouched: every one runs the whole chain on the whole batch.  --spinup-ms 0
gives the cold number (about 3 % lower at --steps 20 --warmup 5).
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import bench_common as bc

ALGO_BYTES_PER_SAMPLE = 2.0 + 2.0 / 50.0      # u8 IQ in + s16 audio out per complex input sample
HBM_PEAK_GBS = bc.HBM_PEAK_GBS                 # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy ceiling)


def cpu_baseline():
    """Reference CPU path on the host cores, bounded sample (rank 0, N=1 only): the unmodified reference (oracle/_ref/cpu_bench_ref), one
    2.4 MS/s u8 IQ stream per thread through the 7-stage chain in process with the CLI's block framing; `cores` = threads used =
    min(CPU affinity, cgroup quota)."""
    return bc.cpu_baseline("wfm", unit="complex MS/s", single_amount=300.0, probe_amount=4.0, target_wall_s=12.0,
                           describe="config 2 WFM chain, one 2.4 MS/s u8 IQ stream per thread, in process with the CLI's block framing")


def pmc_traffic(kernel_name, streams, block):
    return bc.pmc_traffic(kernel_name, {"streams_per_gpu": streams, "block_samples_per_stream": block})


def other_configs():
    """The other BASELINE configs as short legs of their own bench scripts (child processes of this one: inside the driver's clock), each with its --verify:
    ms per step, the dominant kernel and its roofline fraction, whether the shape fell back, verify.ok."""
    legs = [("C1 fir_decimate_cc 10 0.05 HAMMING, 256 x 2.4 M complexf", ["bench_fir.py", "--steps", "60"]),
            ("C1' fir_decimate_cc 50 0.005 (801 taps), 64 x 2.4 M complexf", ["bench_fir.py", "--steps", "60", "--decimation", "50", "--tbw", "0.005", "--streams", "64"]),
            ("C3 bandpass_fir_fft_cc 1023 taps @65536 framing, 64 x 16 blocks", ["bench_fftfilt.py", "--steps", "60", "--no-sweep", "--taps", "1023"]),
            ("C3 bandpass_fir_fft_cc 4095 taps", ["bench_fftfilt.py", "--steps", "60", "--no-sweep", "--taps", "4095"]),
            ("C4 fastddc 256 channels x 64 blocks, fft 65536", ["bench_fastddc.py", "--steps", "120"]),
            ("C5 NFM chain, 512 channels x 2.4 M u8, a shift rate per channel", ["bench_nfm.py", "--steps", "60"]),
            ("C5 NFM chain, one shift rate for all channels", ["bench_nfm.py", "--steps", "60", "--uniform-rate"])]
    out = []
    for name, cmd in legs:
        t0 = time.perf_counter()
        try:
            r = subprocess.run([sys.executable, os.path.join(ROOT, cmd[0])] + cmd[1:] + ["--no-cpu-baseline", "--verify"], capture_output=True, text=True, timeout=240)
            lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
            d = json.loads(lines[-1])
            rf = d.get("roofline", {})
            e = {"config": name, "command": "python " + " ".join(cmd) + " --no-cpu-baseline --verify", "value": d.get("value"), "unit": d.get("unit"), "ms_per_step": d.get("ms_per_step"),
                 "kernel": rf.get("kernel"), "bound": rf.get("bound"), "frac": rf.get("frac"), "kernel_avg_ms": rf.get("kernel_avg_ms"),
                 "fallback": d.get("config", {}).get("fallback"), "verify_ok": d.get("verify", {}).get("ok"), "rc": r.returncode}
            if "full_size_transform" in d:
                e["full_size_transform_frac"] = d["full_size_transform"].get("frac")
        except Exception as ex:  # noqa: BLE001
            e = {"config": name, "error": str(ex)[:300]}
        e["wall_s"] = round(time.perf_counter() - t0, 1)
        out.append(e)
    return out


def operating_points(ctx, taps, verify=True, only=None):
    """The same chain object at the blocks a live receiver bank hands over (the reference's loop moves 16384 samples per read, csdr.c:189-193, 330-392): many streams,
    short blocks.  One step = one call over all streams; state carried; inputs resident in HBM."""
    import torch
    L = ctx.L
    pts = []
    import numpy as np
    if verify and os.path.join(ROOT, "tests") not in sys.path:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
    for i_pt, (S, T, steps, per_stream) in enumerate(((1024, 2344 * 1024, 100, True), (1024, 16384, 400, False), (65536, 24576, 60, False))):
        if only is not None and i_pt not in only:
            continue
        pitch = 2 * T
        x = torch.randint(0, 256, (S, pitch), dtype=torch.uint8, device="cuda")
        na_max = (T // 50 + 64 + 63) // 64 * 64
        out = torch.empty((S, na_max), dtype=torch.int16, device="cuda")
        rates = (-0.45 + 0.9 * (np.arange(S) + 0.5) / S).astype(np.float32) if per_stream else None
        strict_rows = []
        if verify:          # as in the headline: a real FM signal (at -rate of its stream) in four rows spread over the batch, put there before the timed loop
            from tests_helpers import wfm_signal_u8
            strict_rows = sorted({r for r in (5, S // 3 + 1, (2 * S) // 3 + 2, S - 2) if 0 <= r < S})
            for k, r in enumerate(strict_rows):
                x[r, :2 * T] = torch.from_numpy(wfm_signal_u8(7100 + k, T, offset=-float(rates[r]) if per_stream else 0.085)).cuda()
        torch.cuda.synchronize()
        if per_stream:      # the headline shape with a shift rate PER STREAM (csdr_amd_wfm_create_rates: one workgroup = one stream x 16 time segments)
            w = L.csdr_amd_wfm_create_rates(ctx.h, S, rates.ctypes.data_as(C.c_void_p), 10, taps.ctypes.data_as(C.c_void_p), taps.size, 5, 50e-6, 48000, T)
        else:
            w = L.csdr_amd_wfm_create(ctx.h, S, -0.085, 10, taps.ctypes.data_as(C.c_void_p), taps.size, 5, 50e-6, 48000, T)
        if not w:
            pts.append({"streams": S, "block": T, "error": ctx.err()}); continue
        for _ in range(steps // 2):
            L.csdr_amd_wfm_process(w, x.data_ptr(), pitch, T, out.data_ptr(), None, na_max)
        ctx.sync()
        L.csdr_amd_wfm_set_profiling(w, 1)
        t0 = time.perf_counter()
        for _ in range(steps):
            L.csdr_amd_wfm_process(w, x.data_ptr(), pitch, T, out.data_ptr(), None, na_max)
        ctx.sync()
        wall = time.perf_counter() - t0
        kms = C.c_double(0); kl = C.c_long(0)
        L.csdr_amd_wfm_kernel_time(w, C.byref(kms), C.byref(kl))
        k_ms = kms.value / max(kl.value, 1)
        algo = ALGO_BYTES_PER_SAMPLE * S * T
        pts.append({"streams": S, "shift_rates": "per stream (1024 distinct)" if per_stream else "one for all", "block_samples_per_stream": T, "block_ms_of_signal": round(T / 2400.0, 2), "steps": steps, "ms_per_step": round(wall / steps * 1e3, 4),
                    "value": round(S * T * steps / wall / 1e6, 1), "unit": "complex MS/s", "realtime_factor_per_stream": round((T / 2.4e6) / (wall / steps), 1),
                    "kernel": L.csdr_amd_wfm_kernel_name(w).decode(), "kernel_avg_ms": round(k_ms, 4),
                    "frac": round(algo / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if k_ms else None, "fallback": bool(L.csdr_amd_wfm_fallback(w))})
        if verify:          # after the timed loop: fresh-state pass of the same object / buffers, strict rows +-1 LSB and float audio <= 1e-5 against the oracle
            import verify_configs as vc
            L.csdr_amd_wfm_set_profiling(w, 0)
            outf = torch.empty((S, na_max), dtype=torch.float32, device="cuda")
            v = vc.verify_wfm(ctx, w, x, out, S, T, pitch, na_max, taps, shift_rate=rates if per_stream else -0.085, rows=[], strict_rows=strict_rows, out_f32=outf)
            pts[-1]["verify_ok"] = v["ok"]
            pts[-1]["verify"] = {k: v[k] for k in ("strict_rows", "strict_rows_max_abs_diff_lsb", "strict_rows_max_rel_rms", "strict_rows_samples_compared", "rows_expected_len", "rows_got_len")}
            del outf
        L.csdr_amd_wfm_destroy(w)
        del x, out; torch.cuda.empty_cache()
    return pts


def resident_point(ctx, taps, verify=True, S=1024, T=16384, n_slots=8, blocks=4000):
    """The reference's own block (16384 samples per read, csdr.c:189-193, 330-392) through the RESIDENT form of the chain (csdr_amd_wfm_ring_*: one persistent grid walks
    a ring of blocks, no launch per block): `blocks` consecutive blocks of S streams posted as fast as the ring takes them (inputs resident in the ring's slots).  Time per
    block from the DEVICE clock: completion of the last block minus completion of the first timed one, over the blocks between (there is no kernel launch to put events
    around); the host's wall clock beside it."""
    import numpy as np
    import torch
    L = ctx.L
    r = L.csdr_amd_wfm_ring_create(ctx.h, S, -0.085, 10, taps.ctypes.data_as(C.c_void_p), taps.size, 5, 50e-6, 48000, T, n_slots)
    if not r:
        return {"streams": S, "block_samples_per_stream": T, "resident": True, "error": ctx.err()}
    try:
        pitch = C.c_size_t(0)
        x = torch.randint(0, 256, (S, 2 * T), dtype=torch.uint8, device="cuda")
        for k in range(n_slots):
            pi = L.csdr_amd_wfm_ring_input(r, k, C.byref(pitch))
            blk = torch.zeros((S, pitch.value), dtype=torch.uint8, device="cuda"); blk[:, :2 * T] = x
            torch.cuda.synchronize()
            if L.csdr_amd_d2d(ctx.h, pi, blk.data_ptr(), blk.numel()) or L.csdr_amd_ctx_sync(ctx.h):
                return {"streams": S, "block_samples_per_stream": T, "resident": True, "error": ctx.err()}
        del x, blk; torch.cuda.empty_cache(); torch.cuda.synchronize()
        if L.csdr_amd_wfm_ring_replay(r, blocks // 4, None, None):                     # warm-up (clocks, first launch)
            return {"streams": S, "block_samples_per_stream": T, "resident": True, "error": ctx.err()}
        td = np.zeros(blocks, np.float64); tf = np.zeros(blocks, np.float64)
        l0 = L.csdr_amd_wfm_ring_launches(r)
        t0 = time.perf_counter()
        rc = L.csdr_amd_wfm_ring_replay(r, blocks, tf.ctypes.data_as(C.c_void_p), td.ctypes.data_as(C.c_void_p))
        wall = time.perf_counter() - t0
        if rc:
            return {"streams": S, "block_samples_per_stream": T, "resident": True, "error": ctx.err()}
        skip = 2 * n_slots
        per_block_us = (td[-1] - td[skip]) / (blocks - 1 - skip)
        algo = ALGO_BYTES_PER_SAMPLE * S * T
        e = {"streams": S, "shift_rates": "one for all", "block_samples_per_stream": T, "block_ms_of_signal": round(T / 2400.0, 2), "resident": True,
             "steps": blocks, "ms_per_step": round(wall / blocks * 1e3, 5), "value": round(S * T * blocks / wall / 1e6, 1), "unit": "complex MS/s",
             "realtime_factor_per_stream": round((T / 2.4e6) / (wall / blocks), 1),
             "kernel": "k_wfm_mfma_seq<false, true> (resident grid of %d workgroups, %d slots; %d launches during the %d timed blocks)"
                       % (L.csdr_amd_wfm_ring_grid(r), n_slots, L.csdr_amd_wfm_ring_launches(r) - l0, blocks),
             "kernel_avg_ms": round(per_block_us * 1e-3, 5), "kernel_time": "device clock, completion to completion per block (no launch to time)",
             "block_latency_us_median": round(float(np.median(td - tf)), 2),
             "frac": round(algo / (per_block_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4), "fallback": False}
        st = np.zeros(4, np.float64)
        if L.csdr_amd_wfm_ring_stats(r, st.ctypes.data_as(C.c_void_p)) == 0:
            e["us_per_item_waiting_body_completion"] = [round(float(v), 2) for v in st[:3]]
        if verify:
            if os.path.join(ROOT, "tests") not in sys.path:
                sys.path.insert(0, os.path.join(ROOT, "tests"))
            import verify_configs as vc
            v = vc.verify_wfm_ring(ctx, r, S, T, taps)
            e["verify_ok"] = v["ok"]
            e["verify"] = {k: v[k] for k in ("blocks", "samples_compared", "frac_nonzero", "frac_over_1_lsb", "strict_rows", "strict_rows_max_abs_diff_lsb", "strict_rows_samples_compared",
                                             "rows_expected_len", "rows_got_len")}
        return e
    finally:
        L.csdr_amd_wfm_ring_destroy(r)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)        # ~0.5 s of timed work at ~1 ms per step
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--spinup-ms", type=float, default=60.0,
                    help="untimed steps for about this long BEFORE the W warm-up steps: after idle the GPU needs ~25 launches (25 ms) to ramp its clocks -- the first "
                         "launches of the chain kernel run up to 25 %% slower (profiles/r3_notes.md) -- and the metric is a stream's steady-state rate; 0 = off")
    ap.add_argument("--streams", type=int, default=1024)
    ap.add_argument("--block", type=int, default=2344 * 1024)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--verify", action="store_true", help="kept for old command lines: the full check is the default now")
    ap.add_argument("--strict-rows-only", action="store_true", help="check only the four planted FM rows (+-1 LSB, <= 1e-5), not the 16 noise rows")
    ap.add_argument("--no-verify", action="store_true", help="skip the default check: after the timed region, reset, one more pass over all streams with the same object / "
                         "buffers; four planted FM rows held to +-1 LSB and <= 1e-5 on every sample and 16 full audio rows of the noise batch spread over all stream "
                         "blocks under the statistical gate of tests/verify_configs.py, all against the CPU oracle on the same bytes")
    ap.add_argument("--no-other-configs", action="store_true", help="N = 1: skip the short legs of the other BASELINE configs and the short-block operating points")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False); there is no CPU fallback")
    # CSDR_BENCH_SHARED_GPU=1: dry run of the multi-rank code path on a box with ONE GPU (all ranks on device 0, gloo for the
    # barrier / max-over-ranks) -- a test aid for the launcher contract, never a measurement.
    shared = os.environ.get("CSDR_BENCH_SHARED_GPU") == "1"
    dev_index = 0 if shared else local_rank
    torch.cuda.set_device(dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if shared:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))

    import csdr_amd
    ctx = csdr_amd.Context(dev_index)            # own non-blocking HIP stream; timed with HIP events on that stream
    L = ctx.L
    S, T = args.streams, args.block
    assert T % 1024 == 0
    taps = ctx.firdes_lowpass_f(ctx.firdes_filter_len(0.05), 0.5 / 10, "HAMMING")      # csdr.c:1144-1158
    pitch = 2 * T + int(os.environ.get("CSDR_BENCH_PITCH_PAD", "0"))     # row pitch in bytes (experiments: channel/bank spreading)
    # synthetic input resident in HBM (torch is only the allocator / RNG here)
    g = torch.Generator(device="cuda"); g.manual_seed(42 + rank)
    x = torch.randint(0, 256, (S, pitch), dtype=torch.uint8, device="cuda", generator=g)
    strict_rows = []
    do_verify = rank == 0 and not args.no_verify
    if do_verify:
        # rows held to +-1 LSB on every sample: a real FM signal in a few rows spread over the batch, put there BEFORE the timed loop (the kernel's work does not
        # depend on the data; the rest of the batch stays i.i.d. noise, which only a statistical gate can check: tests/verify_configs.py)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from tests_helpers import wfm_signal_u8
        strict_rows = sorted({r for r in (5, S // 3 + 1, (2 * S) // 3 + 2, S - 2) if 0 <= r < S})
        for k, r in enumerate(strict_rows):
            x[r, :2 * T] = torch.from_numpy(wfm_signal_u8(7000 + k, T)).cuda()
    n_audio_max = (T // 50 + 64 + 63) // 64 * 64          # 128-byte aligned s16 rows (16-byte vector stores in the back end)
    out_s16 = torch.empty((S, n_audio_max), dtype=torch.int16, device="cuda")
    torch.cuda.synchronize()
    w = L.csdr_amd_wfm_create(ctx.h, S, -0.085, 10, taps.ctypes.data_as(C.c_void_p), taps.size, 5, 50e-6, 48000, T)
    if not w:
        raise SystemExit("wfm_create: " + ctx.err())

    def step():
        n = L.csdr_amd_wfm_process(w, x.data_ptr(), pitch, T, out_s16.data_ptr(), None, n_audio_max)
        if n < 0:
            raise SystemExit("wfm_process: " + ctx.err())
        return n

    # the first launches after idle (the GPU's clocks ramp for ~25 ms): reported beside the steady-state value, never as it
    cold_k = min(20, args.steps)
    ctx.sync(); tc0 = time.perf_counter()
    for _ in range(cold_k):
        step()
    ctx.sync(); cold_ms = (time.perf_counter() - tc0) / max(cold_k, 1) * 1e3
    spin_steps = bc.spinup(step, ctx.sync, args.spinup_ms)
    for _ in range(args.warmup):
        step()
    ctx.sync(); torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    L.csdr_amd_wfm_set_profiling(w, 1)
    t0 = time.perf_counter()
    ctx.timer_start()
    audio = 0
    for _ in range(args.steps):
        audio += step()
    ev_ms = ctx.timer_stop_ms()                  # HIP events on the kernels' own stream (includes the sync)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    if world > 1:
        dist.barrier()
        tt = torch.tensor([wall], device="cpu" if shared else "cuda", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        wall = float(tt.item())
    kms = C.c_double(0); kl = C.c_long(0)
    L.csdr_amd_wfm_kernel_time(w, C.byref(kms), C.byref(kl))
    kernel_name = L.csdr_amd_wfm_kernel_name(w).decode()

    if rank == 0:
        samples_per_step_gpu = S * T
        total_samples = samples_per_step_gpu * args.steps * world
        msps = total_samples / wall / 1e6
        k_avg_ms = kms.value / max(kl.value, 1)
        achieved_gbs = ALGO_BYTES_PER_SAMPLE * samples_per_step_gpu / (k_avg_ms * 1e-3) / 1e9
        res = {
            "metric": "complex MS/s in->out, WFM demod chain @2.4 MS/s x N streams",
            "value": round(msps, 1), "unit": "complex MS/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "spinup_steps_before_warmup": spin_steps,
            "ms_per_step": round(wall / args.steps * 1e3, 4), "cold_ms_per_step_first_%d_steps_after_idle" % cold_k: round(cold_ms, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs[1]: full WFM pipe u8 IQ -> s16 audio (convert_u8_f|shift_addition_cc -0.085|fir_decimate_cc 10 0.05 HAMMING|"
                                   "fmdemod_quadri_cf|fractional_decimator_ff 5|deemphasis_wfm_ff 48000 50e-6|convert_f_s16)",
                       "streams_per_gpu": S, "block_samples_per_stream": T, "stream_rate_sps": 2400000,
                       "realtime_streams_equivalent": round(msps / 2.4, 1), "parallelism": "streams sharded, no data-path collective",
                       "arithmetic": "dtype f32 = the reference's; the front end (convert_u8_f . shift . FIR, linear in the input bytes) is evaluated as three int8-digit "
                                     "v_mfma_i32_16x16x64_i8 products of the raw bytes with 23-bit fixed-point weights and exact int32 accumulation, recombined in f32 "
                                     "(1.3e-7 relative RMS against the float oracle: not narrower than the reference in effect); demodulator, de-emphasis, s16 conversion in f32",
                       "launches_per_step": "one (k_wfm_mfma_seq: history, partial tiles, state carry inside)", "fallback": bool(L.csdr_amd_wfm_fallback(w))},
            "roofline": {"bound": "hbm", "kernel": kernel_name, "achieved": round(achieved_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved_gbs / HBM_PEAK_GBS, 4), "traffic": None, "traffic_source": None,
                         "algorithmic_bytes_per_launch": ALGO_BYTES_PER_SAMPLE * samples_per_step_gpu,
                         "kernel_avg_ms": round(k_avg_ms, 4), "kernel_launches_timed": kl.value,
                         "frac_of_measured_copy_ceiling_6290": round(achieved_gbs / 6290.0, 4),
                         "hip_event_ms_per_step_all_kernels": round(ev_ms / args.steps, 4)},
            "audio_samples_per_step_per_stream": audio // max(args.steps, 1),
        }
        tr = pmc_traffic(kernel_name, S, T)
        if tr:
            res["roofline"]["traffic"] = tr[0]
            res["roofline"]["traffic_source"] = tr[1]
        if do_verify:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import verify_configs as vc
            L.csdr_amd_wfm_set_profiling(w, 0)
            out_f32 = torch.empty((S, n_audio_max), dtype=torch.float32, device="cuda")      # the float audio of the same pass: north_star's 1e-5 gate on the strict rows
            res["verify"] = vc.verify_wfm(ctx, w, x, out_s16, S, T, pitch, n_audio_max, taps, rows=[] if args.strict_rows_only else [r for r in vc.pick_rows(S) if r not in strict_rows],
                                          strict_rows=strict_rows, out_f32=out_f32)
            del out_f32
        if world == 1 and not args.no_other_configs:
            L.csdr_amd_wfm_destroy(w); w = None
            del x, out_s16; torch.cuda.empty_cache()
            res["operating_points"] = operating_points(ctx, taps, verify=do_verify)
            res["operating_points"].append(resident_point(ctx, taps, verify=do_verify))
            res["other_configs"] = other_configs()
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline()
        # every leg must have verified: a failed or crashed leg turns the run red (the line is still printed first, with the failure inside it)
        bad = []
        if do_verify:
            if not res["verify"]["ok"]:
                bad.append("headline verify: %s" % json.dumps(res["verify"]))
            for e in res.get("operating_points", []):
                if e.get("error") or e.get("verify_ok") is not True:
                    bad.append("operating point %s x %s: %s" % (e.get("streams"), e.get("block_samples_per_stream", e.get("block")), e.get("error") or "verify_ok=%r" % e.get("verify_ok")))
            for e in res.get("other_configs", []):
                if e.get("error") or e.get("verify_ok") is not True or e.get("rc") != 0:
                    bad.append("%s: %s" % (e.get("config"), e.get("error") or "verify_ok=%r rc=%r" % (e.get("verify_ok"), e.get("rc"))))
        res["all_legs_verified"] = (not bad) if do_verify else None
        print(json.dumps(res))
        if bad:
            raise SystemExit("bench.py: a leg did not verify against the oracle:\n  " + "\n  ".join(bad))
    if w:
        L.csdr_amd_wfm_destroy(w)
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
