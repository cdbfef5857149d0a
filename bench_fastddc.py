#!/usr/bin/env python3
"""bench_fastddc.py -- BASELINE.json configs[3]: fastddc_fwd_cc + fastddc_inv_cc, 256 output channels from one 61.44 MS/s
complexf input, channels sharded across N GPUs with an RCCL broadcast of the forward spectrum.

One step = `--blocks` consecutive overlap-save blocks (input_size = 57344 samples each at D=256, tbw=0.001: fft 65536,
taps 8193, fft_inv 512): rank 0 frames + FFTs the new input once, the [blocks, 65536] spectrum is broadcast over xGMI,
every rank folds/IFFTs/post-shifts its slice of the channels.  Reports wideband INPUT MS/s (whole job) and aggregate output MS/s.

    python bench_fastddc.py [--gpus N] [--steps K] [--warmup W] [--channels 256] [--blocks 16]
N > 1:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench_fastddc.py --gpus N ...
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--channels", type=int, default=256)
    ap.add_argument("--decimation", type=int, default=256)
    ap.add_argument("--tbw", type=float, default=0.001)
    ap.add_argument("--blocks", type=int, default=16)
    args = ap.parse_args()

    import numpy as np
    import torch
    import csdr_amd
    from csdr_amd import dist as cd

    if not torch.cuda.is_available():
        raise SystemExit("bench_fastddc.py needs an MI355X; there is no CPU fallback")
    rank, local_rank, world = cd.init()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    stream = torch.cuda.current_stream()
    ctx = csdr_amd.Context(local_rank, hip_stream=stream.cuda_stream)      # same stream as torch/RCCL: ordering is implicit
    L = ctx.L
    ddc, err = ctx.fastddc_init(args.tbw, args.decimation, 0.0)
    assert err == 0
    nb = args.blocks
    # channel c sits at shift_rate = -0.5 + (c + 0.5)/C  
    rates = (-0.5 + (np.arange(args.channels) + 0.5) / args.channels).astype(np.float32)
    first, count = cd.shard(args.channels, rank, world)
    my_rates = np.ascontiguousarray(rates[first:first + count])
    inv = L.csdr_amd_fastddc_inv_create(ctx.h, args.tbw, args.decimation, my_rates.ctypes.data_as(C.c_void_p), count, 2, nb)
    if not inv:
        raise SystemExit("fastddc_inv_create: " + ctx.err())
    pitch = L.csdr_amd_fastddc_inv_max_output(inv, nb) + 8
    out = torch.empty((count, pitch, 2), dtype=torch.float32, device=dev)
    spectra = torch.empty((nb, ddc.fft_size, 2), dtype=torch.float32, device=dev)
    fwd = None
    if rank == 0:
        fwd = L.csdr_amd_fastddc_fwd_create(ctx.h, C.byref(ddc), nb)
        g = torch.Generator(device=dev); g.manual_seed(4)
        x = (torch.rand((nb * ddc.input_size, 2), device=dev, generator=g) * 2 - 1).contiguous()
    torch.cuda.synchronize()

    def step():
        if rank == 0:
            rc = L.csdr_amd_fastddc_fwd_process(fwd, x.data_ptr(), spectra.data_ptr(), nb)
            if rc < 0:
                raise SystemExit(ctx.err())
        cd.broadcast_spectra(spectra, 0)                                   # the one exchange step (RCCL over xGMI)
        rc = L.csdr_amd_fastddc_inv_process(inv, spectra.data_ptr(), nb, out.data_ptr(), pitch, None)
        if rc < 0:
            raise SystemExit(ctx.err())

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(); cd.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    cd.barrier()
    wall = cd.max_over_ranks(wall, dev if world > 1 else "cpu")
    if rank == 0:
        in_samples = nb * ddc.input_size * args.steps
        h_bytes = args.channels * ddc.fft_size * 8                         # per-channel taps_fft, read once per CALL (not per block)
        res = {"metric": "fastddc 256-channel channelizer, wideband input MS/s", "value": round(in_samples / wall / 1e6, 2), "unit": "complex MS/s (input)",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(wall / args.steps * 1e3, 4),
               "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": "configs[3]: fastddc_fwd_cc + fastddc_inv_cc", "channels": args.channels, "decimation": args.decimation,
                          "transition_bw": args.tbw, "fft_size": ddc.fft_size, "fft_inv_size": ddc.fft_inv_size, "blocks_per_step": nb,
                          "parallelism": "channels sharded over ranks, spectrum broadcast (RCCL)"},
               "aggregate_output_msps": round(in_samples / args.decimation * args.channels / wall / 1e6, 2),
               "realtime_factor_at_61p44_msps": round(in_samples / wall / 61.44e6, 3),
               "taps_fft_bytes_per_step": h_bytes}
        print(json.dumps(res))
    L.csdr_amd_fastddc_inv_destroy(inv)
    if fwd:
        L.csdr_amd_fastddc_fwd_destroy(fwd)
    ctx.close()
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
