"""csdr_amd.dist -- the (small) multi-GPU layer: one process per GPU, torch.distributed over RCCL ("nccl" backend
on ROCm) on the GPU box, gloo in the CPU tests.

Where the hot path shards:
  * WFM / NFM chains, converters, FIRs: streams are independent -> block-distribute streams over ranks, NO data-path
    collective (replicas).  Only the barrier and the max-over-ranks of the timing use the communicator.
  * fastddc: the one path with an exchange.  The C library (csdr_amd/csrc/comm.cpp, fftpath.hip) offers two schedules: channel slices (the default of
    csdr_amd_fastddc_bank_create_sharded, north_star's wording: forward transform split by blocks, spectra all-gathered, every rank folds its own channels:
    bank_exchange below is its CPU model) and time slices (CSDR_AMD_SHARD_BLOCKS: every rank runs the whole pipeline on its run of a batch's blocks, the
    decimated outputs are exchanged all-to-all: bank_time_sliced below; the one that scales beyond two GPUs).  fastddc_sharded is the round-1 root broadcast.
"""
import os
import torch
import torch.distributed as dist


def init(backend=None):
    """Initialise from the torchrun environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*). Returns (rank, local_rank, world)."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend, **kw)
    return rank, local_rank, world


def shard(n_items, rank, world):
    """Block distribution of n_items (streams, channels) over ranks: returns (first, count); counts differ by at most 1."""
    base, extra = divmod(n_items, world)
    count = base + (1 if rank < extra else 0)
    first = rank * base + min(rank, extra)
    return first, count


def barrier():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def max_over_ranks(value, device="cpu"):
    """The bench contract: the step time is the MAX over ranks."""
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device="cpu"):
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def broadcast_spectra(spectra, src=0):
    """fastddc exchange step: spectra is a real view [n_blocks, fft_size, 2] float32 tensor (complexf bins), filled on `src`."""
    if dist.is_available() and dist.is_initialized():
        dist.broadcast(spectra, src=src)
    return spectra


def fastddc_sharded(x_or_none, n_samples, ddc_fft_size, ddc_input_size, shift_rates, forward_fn, inverse_fn, rank, world, device="cpu"):
    """One fastddc step over `world` ranks.
    forward_fn(x) -> float32 tensor [n_blocks, fft_size, 2]   (runs on rank 0 only)
    inverse_fn(spectra, rates_slice) -> list of per-channel outputs for this rank's channels
    Returns (first_channel, outputs_of_this_rank)."""
    n_blocks = n_samples // ddc_input_size
    if rank == 0:
        spectra = forward_fn(x_or_none).to(device)
    else:
        spectra = torch.empty((n_blocks, ddc_fft_size, 2), dtype=torch.float32, device=device)
    broadcast_spectra(spectra, 0)
    first, count = shard(len(shift_rates), rank, world)
    return first, inverse_fn(spectra, shift_rates[first:first + count])


def bank_block_ranges(n_blocks, max_blocks, world, input_size, overlap):
    """The block split of a sharded fastddc bank (csdr_amd/csrc/fastddc_mfma.hip, ddc_mfma_submit): rank g transforms blocks [b0, b1) and needs
    the stream samples [s0, s1) = its windows (overlap in front included; s0 < 0 on rank 0 = the kept tail of the previous call).
    Returns (nbl, [(b0, b1, s0, s1) per rank]); nbl = blocks per chunk of the all-gathered transposed spectra."""
    nbl = (max_blocks + world - 1) // world if world > 1 else max_blocks
    out = []
    for g in range(world):
        b0 = min(g * nbl, n_blocks); b1 = min((g + 1) * nbl, n_blocks)
        out.append((b0, b1, b0 * input_size - overlap, b1 * input_size))
    return nbl, out


def bank_exchange(x_or_none, n_blocks, max_blocks, input_size, overlap, fft_size, forward_window_fn, rank, world):
    """The exchange of one batch as the C library schedules it, on torch.distributed (the CPU model of comm.cpp used by tests/test_dist_cpu.py):
    scatter of each rank's window samples from rank 0 (point to point), local forward transforms, all-gather of the per-rank chunks.
    forward_window_fn(samples [n_loc * input_size + overlap] complex64 numpy) -> [n_loc, fft_size] complex64 spectra of those windows.
    Returns the gathered spectra in CHUNK order: [world, nbl, fft_size] complex64 (rows beyond a rank's blocks are zero)."""
    import numpy as np
    nbl, ranges = bank_block_ranges(n_blocks, max_blocks, world, input_size, overlap)
    b0, b1, s0, s1 = ranges[rank]
    if rank == 0:
        xin = np.concatenate([np.zeros(overlap, np.complex64), np.asarray(x_or_none, np.complex64)])      # first call: the kept tail is zero
        reqs = []
        for g in range(1, world):
            g0, g1, t0, t1 = ranges[g]
            if g1 > g0:
                reqs.append(dist.isend(torch.from_numpy(xin[overlap + t0:overlap + t1].view(np.float32).copy()), g))
        mine = xin[overlap + s0:overlap + s1]
        for r in reqs:
            r.wait()
    elif b1 > b0:
        buf = torch.empty(2 * (s1 - s0), dtype=torch.float32)
        dist.recv(buf, 0)
        mine = buf.numpy().view(np.complex64)
    else:
        mine = np.zeros(0, np.complex64)
    chunk = np.zeros((nbl, fft_size), np.complex64)
    if b1 > b0:
        chunk[:b1 - b0] = forward_window_fn(mine)
    gathered = [torch.empty(2 * nbl * fft_size, dtype=torch.float32) for _ in range(world)]
    if world > 1:
        dist.all_gather(gathered, torch.from_numpy(chunk.view(np.float32).reshape(-1).copy()))
    else:
        gathered[0] = torch.from_numpy(chunk.view(np.float32).reshape(-1).copy())
    return nbl, np.stack([t.numpy().view(np.complex64).reshape(nbl, fft_size) for t in gathered])


def bank_time_sliced(x_or_none, n_blocks, max_blocks, input_size, overlap, n_channels, run_fn, rank, world):
    """The time-sliced bank's batch (csdr_amd/csrc/fftpath.hip: bank_submit_blocks / bank_collect_blocks) on torch.distributed -- the CPU model used by
    tests/test_dist_cpu.py: rank 0 sends every rank the samples of its run of blocks (overlap in front, point to point); every rank runs the whole
    channelizer on its run for ALL channels, starting each channel's shift state where the blocks in front of the run leave it (data independent: walked
    locally); the per-channel outputs are exchanged all-to-all so that rank r ends up with its block-distributed slice of the channels, runs in rank order.
    run_fn(samples [overlap + n_loc * input_size] complex64, first_block, n_blocks) -> list of n_channels complex64 arrays (this run's output per channel).
    Returns (first_channel, [stitched output per channel of this rank's slice])."""
    import numpy as np
    nbl, ranges = bank_block_ranges(n_blocks, max_blocks, world, input_size, overlap)
    b0, b1, s0, s1 = ranges[rank]
    if rank == 0:
        xin = np.concatenate([np.zeros(overlap, np.complex64), np.asarray(x_or_none, np.complex64)])
        reqs = []
        for g in range(1, world):
            g0, g1, t0, t1 = ranges[g]
            if g1 > g0:
                reqs.append(dist.isend(torch.from_numpy(xin[overlap + t0:overlap + t1].view(np.float32).copy()), g))
        mine = xin[overlap + s0:overlap + s1]
        for r in reqs:
            r.wait()
    elif b1 > b0:
        buf = torch.empty(2 * (s1 - s0), dtype=torch.float32)
        dist.recv(buf, 0)
        mine = buf.numpy().view(np.complex64)
    else:
        mine = np.zeros(overlap, np.complex64)
    outs = run_fn(mine, b0, n_blocks)                                 # every channel, this run
    # all-to-all of the outputs: lengths first (the C library knows them from the chain; the model just sends them), then the samples
    first, count = shard(n_channels, rank, world)
    pieces = {rank: [outs[first + c] for c in range(count)]}
    for shift in range(1, world):
        to = (rank + shift) % world; frm = (rank - shift) % world
        tf, tc = shard(n_channels, to, world)
        payload = np.concatenate([np.asarray(outs[tf + c], np.complex64) for c in range(tc)]) if tc else np.zeros(0, np.complex64)
        lens = torch.tensor([len(outs[tf + c]) for c in range(tc)], dtype=torch.int64)
        req1 = dist.isend(lens, to); got_lens = torch.empty(count, dtype=torch.int64); dist.recv(got_lens, frm); req1.wait()
        req2 = dist.isend(torch.from_numpy(payload.view(np.float32).copy()), to)
        got = torch.empty(2 * int(got_lens.sum()), dtype=torch.float32); dist.recv(got, frm); req2.wait()
        g = got.numpy().view(np.complex64); at = 0; lst = []
        for c in range(count):
            lst.append(g[at:at + int(got_lens[c])].copy()); at += int(got_lens[c])
        pieces[frm] = lst
    return first, [np.concatenate([pieces[g][c] for g in range(world)]) for c in range(count)]
