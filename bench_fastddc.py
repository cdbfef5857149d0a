#!/usr/bin/env python3
"""bench_fastddc.py -- BASELINE.json configs[3]: fastddc_fwd_cc + fastddc_inv_cc, 256 output channels from one 61.44 MS/s
complexf input, the channels' outputs sharded across N GPUs.

One step = one batch of consecutive overlap-save blocks (input_size = 57344 samples each at D=256, tbw=0.001: fft 65536,
taps 8193, fft_inv 512) through the bank object.  Over N GPUs `--shard channels` (default; csdr_amd_fastddc_bank_create_sharded; BASELINE north_star's
partitioning) shards the channels: the forward transform is split by blocks, the transposed spectra are exchanged, every rank folds its channels -- one batch of
`--blocks` blocks per step ("scaling": "strong").  `--shard blocks` is the time-sliced schedule: every rank runs the whole pipeline on its run of `--blocks`
blocks of a `--blocks` x N batch, the decimated outputs are exchanged all-to-all so that rank r ends up with its slice of the channels (the batch grows with N:
"scaling": "weak (batch = blocks x N)").  Reports wideband INPUT MS/s (whole job) and aggregate output MS/s.
--emulate-world W times ONE rank's work of a W-rank bank on one GPU (no transport) beside a bytes-per-link model of the exchange.

    python bench_fastddc.py [--gpus N] [--steps K] [--warmup W] [--channels 256] [--blocks 64] [--no-cpu-baseline] [--verify]
N > 1:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench_fastddc.py --gpus N ...
Same line format and timing discipline as bench.py (barrier + synchronize, max over ranks).  Roofline of the dominant kernel, the alias fold
(fastddc.c:126-141 for all channels and blocks of a step): 8 flop per (bin, channel, block) against the fp32 matrix-core peak -- the fold is COMPUTE
bound (51 flop per byte of its compulsory traffic: the per-channel taps spectra once, the spectra once, the folded bins once; both figures are in
the line).  CPU baseline: the unmodified reference in process (oracle/cpu_bench.c mode fastddc: one forward transform per block, fastddc_inv_cc per
channel, channels spread over the host threads), FFTs by MKL when present.
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


FMT = {"cf32": ("", 8), "s16": ("_s16", 4), "u8": ("_u8", 2)}      # --input-format: entry-point suffix, bytes per complex sample


def make_input(torch, n_samples, fmt, dev, g):
    """the wideband stream: uniform(-1, 1) complexf, or the integer IQ pairs an SDR delivers (converted inside the forward transform)"""
    if fmt == "s16":
        return torch.randint(-32768, 32768, (n_samples, 2), dtype=torch.int16, device=dev, generator=g)
    if fmt == "u8":
        return torch.randint(0, 256, (n_samples, 2), dtype=torch.uint8, device=dev, generator=g)
    return (torch.rand((n_samples, 2), device=dev, generator=g) * 2 - 1).contiguous()


def verify(ctx, L, ddc, args, x, rates, first, count, nb):
    """Fresh forward / inverse objects, ONE call over the same `nb` blocks of the same input as a timed step (same kernels and tile shapes),
    16 of this rank's channels against the CPU oracle (tests/verify_configs.py)."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import verify_configs as vc
    import torch
    my_rates = np.ascontiguousarray(rates[first:first + count])
    # a context of its own (own non-blocking stream, as the test suite's): creating a second bank on the timed context -- which rides on torch's null stream --
    # faulted now and then inside the create's hipFFT pass over the taps (never seen on a stream of the library's own: profiles/r4_notes.md)
    import csdr_amd
    torch.cuda.synchronize()
    ctx = csdr_amd.Context(x.device.index or 0)
    bank = L.csdr_amd_fastddc_bank_create(ctx.h, args.tbw, args.decimation, my_rates.ctypes.data_as(C.c_void_p), count, 2, nb)
    inv = L.csdr_amd_fastddc_bank_inverse(bank)
    pitch = L.csdr_amd_fastddc_inv_max_output(inv, nb) + 8
    out = torch.zeros((count, pitch, 2), dtype=torch.float32, device=x.device)
    counts = np.zeros(count, np.int32)
    assert getattr(L, "csdr_amd_fastddc_bank_process" + FMT[args.input_format][0])(bank, x.data_ptr(), nb, out.data_ptr(), pitch, counts.ctypes.data_as(C.c_void_p)) >= 0, ctx.err()
    ctx.sync()
    if args.input_format == "cf32":
        xh = x.cpu().numpy().view(np.complex64).ravel()
    else:      # the oracle sees what the reference pipeline would hand fastddc_fwd_cc: the converter's output (csdr_amd_convert_* is bit exact against it, tests/)
        import oracle
        conv = oracle.port().convert_s16_f if args.input_format == "s16" else oracle.port().convert_u8_f
        xh = conv(x.cpu().numpy().ravel()).view(np.complex64)
    chans = vc.pick_rows(count)
    pspec, want = vc.fastddc_oracle_channels(xh, args.tbw, args.decimation, my_rates, chans)
    worst = 0.0
    ok = True
    for c in chans:
        got = out[c, :counts[c]].cpu().numpy().view(np.complex64).ravel()
        ok = ok and got.size == want[c].size
        worst = max(worst, vc.relrms(got[:want[c].size], want[c]))
    kname = L.csdr_amd_fastddc_inv_kernel_name(inv).decode()
    L.csdr_amd_fastddc_bank_destroy(bank)
    ctx.close()
    return {"channels": chans, "blocks": nb, "max_rel_rms": worst, "tolerance": 1e-5, "kernel": kname, "ok": bool(ok and worst < 1e-5)}


# xGMI on MI355X: 7 links per GPU, 153.6 GB/s per link counting BOTH directions (1075 GB/s aggregate in AMD's figures) = 76.8 GB/s per link and direction.
# The exchange model below prices every transfer at that per-direction rate (and shows the optimistic "153.6 per direction" reading beside it).
XGMI_LINK_GBS = 76.8


def emulate(args):
    """--emulate-world W: ONE rank's real per-batch work of a W-rank bank, timed alone on this box's GPU (csdr_amd_comm_create_null: the exchange calls
    return at once and move nothing, every kernel of the rank's schedule runs), for every rank in turn, beside the single-GPU bank on the same number of
    blocks and a bytes-per-link model of the exchange.  An EMULATION: it measures compute per rank, it does not measure a multi-GPU run."""
    import numpy as np
    import torch
    import csdr_amd
    W = args.emulate_world
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    stream = torch.cuda.current_stream()
    ctx = csdr_amd.Context(0, hip_stream=stream.cuda_stream)
    L = ctx.L
    ddc, err = ctx.fastddc_init(args.tbw, args.decimation, 0.0)
    assert err == 0
    nb = args.blocks                                                       # blocks per GLOBAL batch
    rates = (-0.5 + (np.arange(args.channels) + 0.5) / args.channels).astype(np.float32)
    g = torch.Generator(device=dev); g.manual_seed(4)
    sfx, es = FMT[args.input_format]
    x = make_input(torch, nb * ddc.input_size + ddc.overlap_length, args.input_format, dev, g)
    mode = csdr_amd.SHARD[args.shard]

    def time_steps(step, n):
        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n

    # single GPU, same blocks: calls of at most 64 blocks (the fold kernel's tile), i.e. what bench_fastddc.py times, repeated
    per1 = min(nb, 64)
    bank1 = L.csdr_amd_fastddc_bank_create(ctx.h, args.tbw, args.decimation, rates.ctypes.data_as(C.c_void_p), args.channels, 2, per1)
    pitch1 = L.csdr_amd_fastddc_bank_max_output(bank1, per1) + 8
    out1 = torch.empty((args.channels, pitch1, 2), dtype=torch.float32, device=dev)

    def step1():
        for b in range(0, nb, per1):
            if getattr(L, "csdr_amd_fastddc_bank_process" + sfx)(bank1, x.data_ptr() + es * b * ddc.input_size, min(per1, nb - b), out1.data_ptr(), pitch1, None) < 0:
                raise SystemExit(ctx.err())
    t1 = time_steps(step1, args.steps)
    L.csdr_amd_fastddc_bank_destroy(bank1); del out1

    ranks = range(W) if args.rank < 0 else [args.rank]
    t_rank = {}
    for r in ranks:
        comm = L.csdr_amd_comm_create_null(ctx.h, r, W)
        bank = L.csdr_amd_fastddc_bank_create_sharded_by(ctx.h, args.tbw, args.decimation, rates.ctypes.data_as(C.c_void_p), args.channels, 2, nb, comm, mode)
        if not bank:
            raise SystemExit("bank_create_sharded_by: " + ctx.err())
        f0 = C.c_int(); c0 = C.c_int(); L.csdr_amd_fastddc_bank_channel_slice(bank, C.byref(f0), C.byref(c0))
        pitch = L.csdr_amd_fastddc_bank_max_output(bank, nb) + 8
        out = torch.empty((c0.value, pitch, 2), dtype=torch.float32, device=dev)
        submit = getattr(L, ("csdr_amd_fastddc_bank_submit_local" if (args.local_input and mode == 1) else "csdr_amd_fastddc_bank_submit") + sfx)

        def step():
            if submit(bank, x.data_ptr(), nb) < 0 or L.csdr_amd_fastddc_bank_collect(bank, out.data_ptr(), pitch, None) < 0:
                raise SystemExit(ctx.err())
        if submit(bank, x.data_ptr(), nb) < 0:                            # one batch always staged, like the pipelined multi-GPU loop
            raise SystemExit(ctx.err())
        t_rank[r] = time_steps(step, args.steps)
        L.csdr_amd_fastddc_bank_collect(bank, out.data_ptr(), pitch, None)
        L.csdr_amd_fastddc_bank_finish(bank, None)
        ctx.sync(); torch.cuda.synchronize()
        L.csdr_amd_fastddc_bank_destroy(bank); L.csdr_amd_comm_destroy(comm); del out
    worst = max(t_rank.values())
    # ---- exchange model: bytes each GPU pushes through ONE of its links per batch (every peer sits behind its own link: full mesh)
    in_b = float(es) * nb * ddc.input_size; spec_b = 8.0 * nb * ddc.fft_size; out_b = 8.0 * nb * (ddc.post_input_size // ddc.post_decimation) * args.channels
    per_link = {
        "input_scatter_root": in_b / W,                                    # the root sends every peer its 1/W of the stream (only when the stream lives on one GPU)
        "output_all_to_all": out_b / W / W if mode == 1 else 0.0,          # every rank sends each peer 1/W of its 1/W of the outputs
        "spectra_all_gather": spec_b / W if mode == 0 else 0.0,            # every rank sends each peer its 1/W of the spectra
    }
    model = {}
    for name, gbs in (("at_76p8_GBps_per_direction", XGMI_LINK_GBS), ("at_153p6_GBps_per_direction_optimistic", 2 * XGMI_LINK_GBS)):
        t_root = (per_link["input_scatter_root"] + per_link["output_all_to_all"] + per_link["spectra_all_gather"]) / (gbs * 1e9)
        t_dist = (per_link["output_all_to_all"] + per_link["spectra_all_gather"]) / (gbs * 1e9)
        model[name] = {"t_exchange_ms_input_on_rank0": round(t_root * 1e3, 4), "t_exchange_ms_input_distributed_at_ingest": round(t_dist * 1e3, 4),
                       "predicted_scaling_input_on_rank0": round(t1 / max(worst, t_root), 2),
                       "predicted_scaling_input_distributed_at_ingest": round(t1 / max(worst, t_dist), 2)}
    res = {"metric": "fastddc 256-channel channelizer: ONE rank's per-batch work of a world-%d bank, emulated on one GPU" % W, "emulation": True,
           "note": "compute per rank measured alone on one MI355X with a null transport; the exchange is a bytes-per-link model, NOT a measurement; "
                   "exchange and compute overlap by construction (own streams, batch N+1's input / batch N's output under batch N's / N+1's kernels)",
           "world": W, "shard": args.shard, "input_format": args.input_format, "input_bytes_per_sample": es, "blocks_per_global_batch": nb, "channels": args.channels, "steps": args.steps,
           "t1_ms_single_gpu_same_blocks": round(t1 * 1e3, 4), "t_rank_ms": {str(r): round(v * 1e3, 4) for r, v in t_rank.items()},
           "t_rank_ms_worst": round(worst * 1e3, 4), "t_rank_us_per_64_blocks": round(worst * 1e6 * 64 / nb, 2),
           "compute_only_scaling": round(t1 / worst, 2), "bytes_per_link_per_batch": {k: int(v) for k, v in per_link.items()}, "exchange_model": model,
           "input_GSps_single_gpu": round(nb * ddc.input_size / t1 / 1e9, 2), "input_GSps_world_compute_only": round(nb * ddc.input_size / worst / 1e9, 2)}
    print(json.dumps(res))
    ctx.close()


def all_modes(ctx, L, comm, rank, world, args, barrier, allmax, allmin, dev=None, log=None):
    """First contact of the sharded bank with N > 1 ranks: on the communicator `comm` (RCCL; the tests drive the same function over the loopback
    transport with rank threads) run csdr_amd_comm_selftest, then BOTH shard modes x {cf32, s16, u8} in ONE invocation.  Per mode: a fresh sharded bank; its first
    batch is checked on every rank against an unsharded bank of the rank's channel slice on the same input (broadcast from rank 0 for that purpose; gate 2e-6
    relative RMS and equal counts -- the single-GPU bank is itself oracle-clean, tests/test_configs_gpu.py), then `steps` pipelined steps are timed (barrier, max over
    ranks).  Returns {"selftest": [...], "modes": [six entries]} on rank 0 (the other ranks: their own view, unused).
    barrier() / allmax(x) / allmin(x): the process group of the caller (torch.distributed in main(), a threading.Barrier in the tests)."""
    import numpy as np
    import torch
    import csdr_amd
    from csdr_amd import dist as cd
    dev = dev if dev is not None else torch.device("cuda", torch.cuda.current_device())
    ddc, err = ctx.fastddc_init(args.tbw, args.decimation, 0.0)
    assert err == 0
    rates = (-0.5 + (np.arange(args.channels) + 0.5) / args.channels).astype(np.float32)
    rep = C.create_string_buffer(2048)
    rc = L.csdr_amd_comm_selftest(comm, 1 << 20, rep, 2048)
    line = rep.value.decode() if rc == 0 else "FAILED: " + ctx.err()
    if log:
        log("comm selftest: " + line)
    st_ok = allmin(1.0 if rc == 0 else 0.0) > 0.5
    result = {"selftest_ok": bool(st_ok), "selftest_rank0": line, "modes": []}
    if not st_ok:
        return result
    first, count = cd.shard(args.channels, rank, world)
    my_rates = np.ascontiguousarray(rates[first:first + count])
    for shard in ("channels", "blocks"):
        for fmt in ("cf32", "s16", "u8"):
            sfx, es = FMT[fmt]
            nb = args.blocks * (world if shard == "blocks" else 1)
            g = torch.Generator(device=dev); g.manual_seed(4)
            x = make_input(torch, nb * ddc.input_size, fmt, dev, g)      # every rank draws the same stream (same seed); only rank 0's copy is the bank's input ...
            torch.cuda.synchronize()                                      # (torch's stream filled it; the library works on the context's own stream)
            if L.csdr_amd_comm_broadcast(comm, x.data_ptr(), x.numel() * x.element_size(), 0) < 0:      # ... and the others' copies are overwritten with it, for the check
                raise SystemExit("comm_broadcast: " + ctx.err())
            ctx.sync()
            entry = {"shard": shard, "input_format": fmt, "blocks_per_step": nb}
            bank = L.csdr_amd_fastddc_bank_create_sharded_by(ctx.h, args.tbw, args.decimation, rates.ctypes.data_as(C.c_void_p), args.channels, 2, nb, comm, csdr_amd.SHARD[shard])
            if not bank:
                raise SystemExit("fastddc_bank_create_sharded_by(%s): %s" % (shard, ctx.err()))
            pitch = L.csdr_amd_fastddc_bank_max_output(bank, nb) + 8
            out = torch.zeros((count, pitch, 2), dtype=torch.float32, device=dev)
            out1 = torch.zeros((count, pitch, 2), dtype=torch.float32, device=dev)
            torch.cuda.synchronize()
            submit = getattr(L, "csdr_amd_fastddc_bank_submit" + sfx)
            xp = x.data_ptr() if rank == 0 else None
            # ---- the first batch against the unsharded bank of this rank's slice
            counts = np.zeros(count, np.int32)
            if submit(bank, xp, nb) < 0 or L.csdr_amd_fastddc_bank_collect(bank, out.data_ptr(), pitch, None) < 0 or L.csdr_amd_fastddc_bank_finish(bank, counts.ctypes.data_as(C.c_void_p)) < 0:
                raise SystemExit("sharded step (%s, %s): %s" % (shard, fmt, ctx.err()))
            ctx.sync()
            ref = L.csdr_amd_fastddc_bank_create(ctx.h, args.tbw, args.decimation, my_rates.ctypes.data_as(C.c_void_p), count, 2, nb)
            if not ref:
                raise SystemExit("fastddc_bank_create (reference of the check): " + ctx.err())
            counts1 = np.zeros(count, np.int32)
            if getattr(L, "csdr_amd_fastddc_bank_process" + sfx)(ref, x.data_ptr(), nb, out1.data_ptr(), pitch, counts1.ctypes.data_as(C.c_void_p)) < 0:
                raise SystemExit("reference bank: " + ctx.err())
            ctx.sync()
            L.csdr_amd_fastddc_bank_destroy(ref)
            worst = 0.0; same = bool((counts == counts1).all() and counts.min() > 0)
            if same:
                a = out.cpu().numpy().view(np.complex64)[..., 0]; b = out1.cpu().numpy().view(np.complex64)[..., 0]
                for c in range(count):
                    n = int(counts[c]); den = float(np.sqrt((np.abs(b[c, :n]) ** 2).sum()))
                    worst = max(worst, float(np.sqrt((np.abs(a[c, :n] - b[c, :n]) ** 2).sum())) / den if den else 1.0)
            ok_here = same and worst < 2e-6
            entry["verify"] = {"against": "the unsharded bank of every rank's channel slice, first batch, all of the slice's channels", "tolerance": 2e-6,
                               "max_rel_rms_over_ranks": allmax(worst), "ok": bool(allmin(1.0 if ok_here else 0.0) > 0.5)}
            del out1
            # ---- timed: one batch always staged (submit(N + 1) before collect(N))
            def step():
                if submit(bank, xp, nb) < 0 or L.csdr_amd_fastddc_bank_collect(bank, out.data_ptr(), pitch, None) < 0:
                    raise SystemExit(ctx.err())
            if submit(bank, xp, nb) < 0:
                raise SystemExit(ctx.err())
            for _ in range(args.warmup):
                step()
            ctx.sync(); barrier()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step()
            L.csdr_amd_fastddc_bank_finish(bank, None)
            ctx.sync()
            wall = allmax(time.perf_counter() - t0)
            barrier()
            L.csdr_amd_fastddc_bank_collect(bank, out.data_ptr(), pitch, None)
            L.csdr_amd_fastddc_bank_finish(bank, None)
            ctx.sync()
            L.csdr_amd_fastddc_bank_destroy(bank)
            entry["ms_per_step"] = round(wall / args.steps * 1e3, 4)
            entry["value"] = round(nb * ddc.input_size * args.steps / wall / 1e6, 2); entry["unit"] = "complex MS/s (input)"
            entry["scaling"] = "weak (batch = blocks x N)" if shard == "blocks" else "strong"
            if log:
                log("%s / %s: %.4f ms per step, %.1f MS/s, verify %s (%.2e)" % (shard, fmt, entry["ms_per_step"], entry["value"], entry["verify"]["ok"], entry["verify"]["max_rel_rms_over_ranks"]))
            result["modes"].append(entry)
            del x, out
    return result


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--emulate-world", type=int, default=0, help="time ONE rank's work of a world-N bank on this GPU (null transport) instead of running the bench")
    ap.add_argument("--rank", type=int, default=-1, help="with --emulate-world: the rank to time (default: every rank in turn)")
    ap.add_argument("--shard", choices=["auto", "blocks", "channels"], default="auto",
                    help="auto (default) = what csdr_amd_fastddc_bank_create_sharded picks for the world size (csdr_amd_fastddc_bank_default_shard_mode: channels up to two "
                         "GPUs, blocks beyond); channels = BASELINE north_star's partitioning (channels sharded, spectra exchanged); blocks = the time-sliced schedule "
                         "(every rank runs the whole pipeline on its run of the batch's blocks, decimated outputs exchanged all-to-all)")
    ap.add_argument("--local-input", action="store_true", help="with --emulate-world --shard blocks: every rank is handed its own run (no input exchange)")
    ap.add_argument("--input-format", choices=["cf32", "s16", "u8"], default="cf32",
                    help="the wideband stream as complexf (default; SURVEY.md 8d) or as the s16 / u8 IQ pairs an SDR delivers: converted inside the forward transform "
                         "(bit-equal to convert_s16_f / convert_u8_f in front, README.md:66-87); a sharded bank scatters the raw integers")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--channels", type=int, default=256)
    ap.add_argument("--decimation", type=int, default=256)
    ap.add_argument("--tbw", type=float, default=0.001)
    ap.add_argument("--blocks", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--verify", action="store_true")
    ap.add_argument("--all-modes", action="store_true",
                    help="after the headline measurement: csdr_amd_comm_selftest, then both shard modes x {cf32, s16, u8} in this one invocation, each checked against the unsharded "
                         "bank (\"modes\" in the line).  Default when N > 1 (--single-mode switches it off)")
    ap.add_argument("--single-mode", action="store_true", help="N > 1: only the --shard / --input-format asked for")
    args = ap.parse_args()
    shard_auto = args.shard == "auto"
    if shard_auto:       # the library's own choice for this world size (the emulation of one rank: for the emulated world)
        import csdr_amd as _ca
        _w = args.emulate_world if args.emulate_world else int(os.environ.get("WORLD_SIZE", "1"))
        args.shard = "blocks" if _ca.lib().csdr_amd_fastddc_bank_default_shard_mode(max(_w, 1)) == _ca.SHARD["blocks"] else "channels"
    if args.emulate_world:
        return emulate(args)

    import numpy as np
    import torch
    import csdr_amd
    from csdr_amd import dist as cd

    if not torch.cuda.is_available():
        raise SystemExit("bench_fastddc.py needs an MI355X; there is no CPU fallback")
    # CSDR_BENCH_SHARED_GPU=1: dry run of the multi-rank launcher contract on a box with ONE GPU (all ranks on device 0, gloo; RCCL refuses two ranks on
    # one device, so the exchange is a gloo broadcast of the natural-order spectrum through the two-object API) -- a test aid, never a measurement.
    # CSDR_BENCH_SHARED_GPU=rccl: all ranks on device 0 as well, but the bank's exchange over the library's own RCCL communicator (torch.distributed over gloo only
    # carries the 128-byte id and the barrier): the N > 1 RCCL call pattern on a one-GPU box, if RCCL accepts several ranks on one device.
    shared_rccl = os.environ.get("CSDR_BENCH_SHARED_GPU") == "rccl"
    shared = os.environ.get("CSDR_BENCH_SHARED_GPU") == "1"
    rank, local_rank, world = cd.init("gloo" if (shared or shared_rccl) else None)
    dev_index = 0 if (shared or shared_rccl) else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    stream = torch.cuda.current_stream()
    ctx = csdr_amd.Context(dev_index, hip_stream=stream.cuda_stream)       # the context's stream = torch's current stream: ordering with torch is implicit
    L = ctx.L
    ddc, err = ctx.fastddc_init(args.tbw, args.decimation, 0.0)
    assert err == 0
    nb = args.blocks * (world if (world > 1 and not shared and args.shard == "blocks") else 1)      # time slices: every rank gets a run of --blocks blocks per batch
    # channel c sits at shift_rate = -0.5 + (c + 0.5)/C  
    rates = (-0.5 + (np.arange(args.channels) + 0.5) / args.channels).astype(np.float32)
    first, count = cd.shard(args.channels, rank, world)
    my_rates = np.ascontiguousarray(rates[first:first + count])
    bank = None; comm = None; inv = None; fwd = None
    if world == 1 or not shared:
        # the bank object: forward + inverse; over N GPUs the library's own RCCL communicator (the 128-byte id travels over torch.distributed) shards the
        # channels, splits the forward transform by blocks and all-gathers the transposed spectra (csdr_amd/csrc/comm.cpp)
        if world > 1:
            idt = torch.zeros(128, dtype=torch.uint8, device="cpu" if shared_rccl else dev)
            if rank == 0:
                idb = (C.c_char * 128)()
                if L.csdr_amd_comm_unique_id(idb) < 0:
                    raise SystemExit("comm_unique_id: " + ctx.err())
                idt.copy_(torch.frombuffer(bytearray(idb.raw), dtype=torch.uint8))
            torch.distributed.broadcast(idt, 0)
            idb = (C.c_char * 128).from_buffer_copy(bytes(idt.cpu().numpy().tobytes()))
            comm = L.csdr_amd_comm_create(ctx.h, idb, rank, world)
            if not comm:
                raise SystemExit("comm_create: " + ctx.err())
            bank = L.csdr_amd_fastddc_bank_create_sharded_by(ctx.h, args.tbw, args.decimation, rates.ctypes.data_as(C.c_void_p), args.channels, 2, nb, comm, csdr_amd.SHARD[args.shard])
        else:
            bank = L.csdr_amd_fastddc_bank_create(ctx.h, args.tbw, args.decimation, my_rates.ctypes.data_as(C.c_void_p), count, 2, nb)
        if not bank:
            raise SystemExit("fastddc_bank_create: " + ctx.err())
        inv = L.csdr_amd_fastddc_bank_inverse(bank)
        if world > 1:
            f0 = C.c_int(); c0 = C.c_int(); L.csdr_amd_fastddc_bank_channel_slice(bank, C.byref(f0), C.byref(c0))
            assert (f0.value, c0.value) == (first, count)
    else:
        inv = L.csdr_amd_fastddc_inv_create(ctx.h, args.tbw, args.decimation, my_rates.ctypes.data_as(C.c_void_p), count, 2, nb)
        if not inv:
            raise SystemExit("fastddc_inv_create: " + ctx.err())
    pitch = L.csdr_amd_fastddc_inv_max_output(inv, nb) + 8
    out = torch.empty((count, pitch, 2), dtype=torch.float32, device=dev)
    spectra = None if bank else torch.empty((nb, ddc.fft_size, 2), dtype=torch.float32, device=dev)
    x = None
    if rank == 0:
        if not bank:
            fwd = L.csdr_amd_fastddc_fwd_create(ctx.h, C.byref(ddc), nb)
        g = torch.Generator(device=dev); g.manual_seed(4)
        x = make_input(torch, nb * ddc.input_size, args.input_format, dev, g)
    xp = x.data_ptr() if x is not None else None
    torch.cuda.synchronize()

    # One step = one batch of `nb` blocks through the whole channelizer.  Over several GPUs the batches are software-pipelined the way a stream is
    # processed: batch N+1 is staged (chains, exchange, forward transform: side stream) while batch N is folded -- every step still does one submit and one
    # collect.  On one GPU there is nothing to hide the transforms under (the fold's workgroups hold the whole LDS of every CU; measured: 0.184 ms
    # pipelined vs 0.179 ms in order, profiles/r2f_*), so a step is one process() call.
    pipelined = bank is not None and world > 1
    if args.input_format != "cf32" and not bank:
        raise SystemExit("--input-format %s needs the bank object" % args.input_format)
    bank_submit = getattr(L, "csdr_amd_fastddc_bank_submit" + FMT[args.input_format][0])
    bank_process = getattr(L, "csdr_amd_fastddc_bank_process" + FMT[args.input_format][0])

    def step():
        if pipelined:
            if bank_submit(bank, xp, nb) < 0 or L.csdr_amd_fastddc_bank_collect(bank, out.data_ptr(), pitch, None) < 0:
                raise SystemExit(ctx.err())
            return
        if bank:
            if bank_process(bank, xp, nb, out.data_ptr(), pitch, None) < 0:
                raise SystemExit(ctx.err())
            return
        if rank == 0:
            rc = L.csdr_amd_fastddc_fwd_process(fwd, x.data_ptr(), spectra.data_ptr(), nb)
            if rc < 0:
                raise SystemExit(ctx.err())
        if shared:                                                       # gloo moves host tensors
            sp = spectra.cpu(); cd.broadcast_spectra(sp, 0); spectra.copy_(sp)
        else:
            cd.broadcast_spectra(spectra, 0)
        rc = L.csdr_amd_fastddc_inv_process(inv, spectra.data_ptr(), nb, out.data_ptr(), pitch, None)
        if rc < 0:
            raise SystemExit(ctx.err())

    if pipelined and bank_submit(bank, xp, nb) < 0:      # prime the pipeline: from here on one batch is always staged
        raise SystemExit(ctx.err())
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(); cd.barrier()
    L.csdr_amd_fastddc_inv_set_profiling(inv, 1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if bank:
        L.csdr_amd_fastddc_bank_finish(bank, None)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    kms = C.c_double(0); kl = C.c_long(0)
    L.csdr_amd_fastddc_inv_kernel_time(inv, C.byref(kms), C.byref(kl))
    kname = L.csdr_amd_fastddc_inv_kernel_name(inv).decode()
    stage_ms = {}
    if world == 1 and not pipelined:                                     # a short pass of its own (behind the timed region: every event pair is a marker packet between two kernels)
        L.csdr_amd_fastddc_inv_set_profiling(inv, 2)
        for _ in range(40):
            step()
        ctx.sync(); torch.cuda.synchronize()
    for stg in (1, 2):
        sm = C.c_double(0); sl = C.c_long(0)
        L.csdr_amd_fastddc_inv_stage_time(inv, stg, C.byref(sm), C.byref(sl))
        stage_ms[stg] = sm.value / sl.value if sl.value else 0.0
    cd.barrier()
    red_dev = dev if (world > 1 and not shared and not shared_rccl) else "cpu"
    wall = cd.max_over_ranks(wall, red_dev)
    modes = None
    if bank is not None and (args.all_modes or (world > 1 and not args.single_mode)):
        # the timed bank is drained and released first: the six banks of the sweep are made on the same communicator
        if pipelined:
            L.csdr_amd_fastddc_bank_collect(bank, out.data_ptr(), pitch, None)
        L.csdr_amd_fastddc_bank_finish(bank, None)
        ctx.sync(); torch.cuda.synchronize()
        L.csdr_amd_fastddc_bank_destroy(bank); bank = None; inv = None
        own_comm = comm
        if own_comm is None:      # one GPU: a one-rank RCCL communicator, so that the same code (selftest, csdr_amd_comm_dup, sharded create) runs here too
            idb = (C.c_char * 128)()
            if L.csdr_amd_comm_unique_id(idb) < 0:
                raise SystemExit("comm_unique_id: " + ctx.err())
            own_comm = L.csdr_amd_comm_create(ctx.h, idb, 0, 1)
            if not own_comm:
                raise SystemExit("comm_create: " + ctx.err())
        modes = all_modes(ctx, L, own_comm, rank, world, args, cd.barrier, lambda v: cd.max_over_ranks(v, red_dev), lambda v: -cd.max_over_ranks(-v, red_dev), dev=dev,
                          log=(lambda m: print("[bench_fastddc rank 0] " + m, file=sys.stderr)) if rank == 0 else None)
        if comm is None:
            L.csdr_amd_comm_destroy(own_comm)
    if rank == 0:
        in_samples = nb * ddc.input_size * args.steps
        h_bytes = args.channels * ddc.fft_size * 8                         # per-channel taps_fft, read once per CALL (not per block)
        res = {"metric": "fastddc 256-channel channelizer, wideband input MS/s", "value": round(in_samples / wall / 1e6, 2), "unit": "complex MS/s (input)",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(wall / args.steps * 1e3, 4),
               "higher_is_better": True, "scaling": "weak (batch = blocks x N)" if (world > 1 and not shared and args.shard == "blocks") else "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": "configs[3]: fastddc_fwd_cc + fastddc_inv_cc" + ("" if args.input_format == "cf32" else " fed %s IQ pairs (convert_%s_f fused into the forward transform)" % (args.input_format, args.input_format)),
                          "input_format": args.input_format, "channels": args.channels, "decimation": args.decimation,
                          "transition_bw": args.tbw, "fft_size": ddc.fft_size, "fft_inv_size": ddc.fft_inv_size, "blocks_per_step": nb,
                          "parallelism": ("time slices: every rank runs the whole pipeline on its run of the batch's blocks (input sent point to point from rank 0), decimated outputs exchanged "
                                          "all-to-all so that rank r delivers its slice of the channels (RCCL, from libcsdr_amd.so)") if args.shard == "blocks" else
                                         "channels sharded over ranks; forward transform split by blocks, input scattered point-to-point, transposed spectra all-gathered (RCCL, from libcsdr_amd.so)",
                          "pipelining": "batch N+1's input exchange and batch N's output exchange on their own streams beside the kernels" if world > 1 else "none (one process() per step)"},
               "aggregate_output_msps": round(in_samples / args.decimation * args.channels / wall / 1e6, 2),
               "realtime_factor_at_61p44_msps": round(in_samples / wall / 61.44e6, 3),
               "taps_fft_bytes_per_step": h_bytes}
        k_avg_ms = kms.value / max(kl.value, 1)
        flops = 8.0 * ddc.fft_size * count * nb                       # complex MAC per (bin, channel, block) of this rank's channel slice
        hbm_min = 8.0 * ddc.fft_size * count + 8.0 * ddc.fft_size * nb + 8.0 * ddc.fft_inv_size * count * nb      # taps spectra + spectra in, folded bins out
        if k_avg_ms > 0:
            tf = flops / (k_avg_ms * 1e-3) / 1e12
            res["roofline"] = {"bound": "mfma", "kernel": kname, "achieved": round(tf, 2), "peak": bc.FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / bc.FP32_PEAK_TFLOPS, 4),
                               "traffic": None, "traffic_source": None, "algorithmic_flops_per_launch": flops, "algorithmic_bytes_per_launch": hbm_min, "kernel_avg_ms": round(k_avg_ms, 4),
                               "kernel_launches_timed": kl.value,
                               "hbm_bound_alternative": {"algorithmic_bytes_per_launch": hbm_min, "achieved_GBps": round(hbm_min / (k_avg_ms * 1e-3) / 1e9, 1),
                                                         "frac_of_8TBps": round(hbm_min / (k_avg_ms * 1e-3) / 1e9 / bc.HBM_PEAK_GBS, 4),
                                                         "note": "51 flop per compulsory byte > the fp32 ridge of 19.7: the fp32 matrix-core peak is the roofline, not HBM"}}
            tr = bc.pmc_traffic(kname, {"channels": args.channels, "blocks_per_step": nb})
            if tr:
                res["roofline"]["traffic"], res["roofline"]["traffic_source"] = tr
            if kname.startswith("k_ddc_gemm3"):
                # ALGORITHMIC flops (4 real multiply-adds per complex one, fastddc.c:126-141) over the kernel's time; the kernel itself issues 3 (Gauss), i.e. the
                # matrix pipe is busy for 3/4 of this fraction
                res["roofline"]["executed_flops_per_launch"] = 0.75 * flops
                res["roofline"]["matrix_pipe_frac"] = round(0.75 * tf / bc.FP32_PEAK_TFLOPS, 4)
            # the two kernels around the fold, each with its own HBM roofline (HIP events around every launch, csdr_amd_fastddc_inv_stage_time): pass 1 of the forward
            # transform reads the step's new samples and writes the 512 x 128 intermediate; the inverse transforms read the folded bins and write the decimated channels
            es = {"cf32": 8, "s16": 4, "u8": 2}.get(args.input_format, 8)
            others = []
            for stg, nm, byts in ((1, "k_ddc_fwd512", float(es) * ddc.input_size * nb + 8.0 * ddc.fft_size * nb),
                                  (2, "k_ddc_ifft256d_post" if ddc.post_decimation == 2 else "k_ddc_ifft512_post",
                                   8.0 * ddc.fft_inv_size * count * nb + 8.0 * (ddc.post_input_size // ddc.post_decimation) * count * nb)):
                if stage_ms.get(stg, 0) > 0:
                    others.append({"kernel": nm, "bound": "hbm", "kernel_avg_ms": round(stage_ms[stg], 4), "algorithmic_bytes_per_launch": byts,
                                   "achieved_GBps": round(byts / (stage_ms[stg] * 1e-3) / 1e9, 1), "frac": round(byts / (stage_ms[stg] * 1e-3) / 1e9 / bc.HBM_PEAK_GBS, 4)})
            if others:
                res["roofline"]["other_kernels"] = others
        else:
            res["roofline"] = {"bound": "mfma", "kernel": kname, "achieved": None, "peak": bc.FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": None, "traffic": None,
                               "note": "general kernels (geometry outside the matrix-core path): no per-kernel timing"}
        if args.verify:
            # with the timed objects gone: two banks alive at once + integer input faulted sporadically inside the second bank's first call (open: profiles/r4_notes.md)
            if world == 1:
                ctx.sync(); torch.cuda.synchronize()
                if bank:
                    L.csdr_amd_fastddc_bank_destroy(bank); bank = None; inv = None      # (the bank owned its inverse half)
                elif inv:
                    L.csdr_amd_fastddc_inv_destroy(inv); inv = None
            res["verify"] = verify(ctx, L, ddc, args, x, rates, first, count, nb)
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = bc.cpu_baseline("fastddc", (args.channels, args.tbw), unit="complex MS/s (input)", single_amount=1, probe_amount=4, target_wall_s=10.0,
                                                  fast_fft=True, describe="config 4 in process: fastddc_fwd_cc framing + FFT once per block, fastddc_inv_cc for %d channels "
                                                  "spread over the threads (every thread transforms the block itself)" % args.channels)
        if modes is not None:
            res["comm_selftest_ok"] = modes["selftest_ok"]; res["comm_selftest_rank0"] = modes["selftest_rank0"]; res["modes"] = modes["modes"]
            # the headline of an N-GPU line = the best VERIFIED mode of the same ingest format (all six stay in "modes"); the timed run above is the library's default schedule
            good = [m for m in modes["modes"] if m.get("verify", {}).get("ok") and m.get("input_format") == args.input_format and m.get("value")]
            res["config"]["default_schedule"] = {"shard": args.shard, "chosen_by": "csdr_amd_fastddc_bank_default_shard_mode(world)" if shard_auto else "--shard", "value": res["value"], "ms_per_step": res["ms_per_step"]}
            if good:
                best = max(good, key=lambda m: m["value"])
                res["config"]["headline_mode"] = {"shard": best["shard"], "input_format": best["input_format"], "source": "modes (best verified)"}
                if best["value"] > res["value"]:
                    res["value"] = best["value"]; res["ms_per_step"] = best["ms_per_step"]
        print(json.dumps(res))
        if modes is not None and (not modes["selftest_ok"] or not all(m["verify"]["ok"] for m in modes["modes"])):
            raise SystemExit("bench_fastddc.py: the communicator self test or a mode's check against the unsharded bank failed: %s" % json.dumps(modes))
        if args.verify and not res["verify"]["ok"]:
            raise SystemExit("bench_fastddc.py --verify failed: %s" % json.dumps(res["verify"]))
    if bank:
        if pipelined:
            L.csdr_amd_fastddc_bank_collect(bank, out.data_ptr(), pitch, None)   # drain the staged batch
        L.csdr_amd_fastddc_bank_finish(bank, None)
        ctx.sync(); torch.cuda.synchronize()
        L.csdr_amd_fastddc_bank_destroy(bank)
    elif inv:
        L.csdr_amd_fastddc_inv_destroy(inv)
    if comm:
        L.csdr_amd_comm_destroy(comm)
    if fwd:
        L.csdr_amd_fastddc_fwd_destroy(fwd)
    ctx.close()
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
