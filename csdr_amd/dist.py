"""csdr_amd.dist -- the (small) multi-GPU layer: one process per GPU, torch.distributed over RCCL ("nccl" backend
on ROCm) on the GPU box, gloo in the CPU tests.

Where the hot path shards:
  * WFM / NFM chains, converters, FIRs: streams are independent -> block-distribute streams over ranks, NO data-path
    collective (replicas).  Only the barrier and the max-over-ranks of the timing use the communicator.
  * fastddc: ONE exchange step -- the forward overlap-save FFT is computed once (rank 0) and the fft_size spectrum
    block is broadcast; rank r then runs its slice of the channels (own taps_fft slab), outputs stay per rank.
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
