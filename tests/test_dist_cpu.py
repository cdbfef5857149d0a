"""world_size-2 gloo tests (CPU) of the multi-GPU layer: stream/channel sharding, the max-over-ranks timing reduction of
the bench contract, and the fastddc spectrum broadcast + channel sharding checked against the unsharded oracle."""
import os
import sys
import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from csdr_amd import dist as cd
    import oracle
    r, lr, w = cd.init("gloo")
    assert (r, w) == (rank, world)
    # 1. timing reduction of the bench contract
    mx = cd.max_over_ranks(1.0 + rank)
    sm = cd.sum_over_ranks(10.0)
    # 2. fastddc: forward FFT on rank 0, broadcast, channels sharded
    port_o = oracle.port()
    D, tbw = 16, 0.05
    rates = [-0.1, 0.2, 0.33, -0.4, 0.05]
    ddc, _ = port_o.fastddc_init(tbw, D, 0.0)
    rng = np.random.default_rng(4)
    n = ddc.input_size * 6
    x = (rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n)).astype(np.complex64)

    def fwd(xx):
        return torch.from_numpy(port_o.fastddc_fwd_cc(xx, ddc).view(np.float32).reshape(-1, ddc.fft_size, 2).copy())

    def inv(spec, rr):
        s = spec.numpy().reshape(-1, ddc.fft_size * 2).view(np.complex64)
        outs = []
        for rate in rr:
            d, _ = port_o.fastddc_init(tbw, D, rate)
            outs.append(port_o.fastddc_inv_cc(s, d, port_o.fastddc_taps_fft(d, rate, D)))
        return outs

    first, outs = cd.fastddc_sharded(x if rank == 0 else None, n, ddc.fft_size, ddc.input_size, rates, fwd, inv, rank, world)
    cd.barrier()
    q.put((rank, mx, sm, first, [o.tolist() for o in outs]))


def test_gloo_world2_sharding_broadcast_and_timing():
    import oracle
    from csdr_amd import dist as cd
    assert [cd.shard(1024, r, 8) for r in range(8)] == [(128 * r, 128) for r in range(8)]
    assert [cd.shard(5, r, 2) for r in range(2)] == [(0, 3), (3, 2)]
    assert sum(cd.shard(4097, r, 8)[1] for r in range(8)) == 4097
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(abs(r[1] - 2.0) < 1e-12 for r in res)             # max over ranks of (1, 2)
    assert all(abs(r[2] - 20.0) < 1e-12 for r in res)
    # unsharded reference
    po = oracle.port()
    D, tbw = 16, 0.05
    rates = [-0.1, 0.2, 0.33, -0.4, 0.05]
    ddc, _ = po.fastddc_init(tbw, D, 0.0)
    rng = np.random.default_rng(4)
    n = ddc.input_size * 6
    x = (rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n)).astype(np.complex64)
    spec = po.fastddc_fwd_cc(x, ddc)
    got = {}
    for rank, _, _, first, outs in res:
        for k, o in enumerate(outs):
            got[first + k] = np.array(o, dtype=np.complex64)
    assert sorted(got) == list(range(len(rates)))
    for c, rate in enumerate(rates):
        d, _ = po.fastddc_init(tbw, D, rate)
        ref = po.fastddc_inv_cc(spec, d, po.fastddc_taps_fft(d, rate, D))
        assert np.array_equal(got[c], ref)


def _bank_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from csdr_amd import dist as cd
    import oracle
    cd.init("gloo")
    po = oracle.port()
    D, tbw = 16, 0.05
    rates = [-0.1, 0.2, 0.33, -0.4, 0.05]
    ddc, _ = po.fastddc_init(tbw, D, 0.0)
    n_blocks, max_blocks = 7, 8                                     # 3 ranks x 3 blocks per chunk: the last rank holds one block, n_blocks < max_blocks
    rng = np.random.default_rng(14)
    x = (rng.uniform(-1, 1, ddc.input_size * n_blocks) + 1j * rng.uniform(-1, 1, ddc.input_size * n_blocks)).astype(np.complex64)

    def fwd_windows(samples):                                       # samples = overlap ++ n_loc * input_size: one overlap-save window per block (csdr.c:2292-2296)
        n_loc = (samples.size - ddc.overlap_length) // ddc.input_size
        return np.stack([po.fft_c2c(samples[b * ddc.input_size:b * ddc.input_size + ddc.fft_size], True) for b in range(n_loc)])

    nbl, chunks = cd.bank_exchange(x if rank == 0 else None, n_blocks, max_blocks, ddc.input_size, ddc.overlap_length, ddc.fft_size, fwd_windows, rank, world)
    # every rank now holds all spectra in chunk order: block b = chunks[b // nbl][b % nbl]; fold this rank's channel slice
    spec = np.stack([chunks[b // nbl][b % nbl] for b in range(n_blocks)])
    first, count = cd.shard(len(rates), rank, world)
    outs = []
    for rate in rates[first:first + count]:
        d, _ = po.fastddc_init(tbw, D, rate)
        outs.append(po.fastddc_inv_cc(spec, d, po.fastddc_taps_fft(d, rate, D)).tolist())
    cd.barrier()
    q.put((rank, first, nbl, outs))


def test_gloo_world3_bank_schedule():
    """The sharded bank's batch schedule (comm.cpp / fastddc_mfma.hip ddc_mfma_submit) on gloo with three ranks: the root scatters every rank the samples of
    its blocks' windows point to point, the ranks transform their blocks, the chunks are all-gathered, every rank folds its channels -- bit-identical
    to the unsharded oracle (same arithmetic, different owner)."""
    import oracle
    from csdr_amd import dist as cd
    nbl, rg = cd.bank_block_ranges(64, 64, 8, 57344, 8192)
    assert nbl == 8 and rg[0] == (0, 8, -8192, 8 * 57344) and rg[7] == (56, 64, 56 * 57344 - 8192, 64 * 57344)
    nbl, rg = cd.bank_block_ranges(5, 64, 8, 57344, 8192)            # a short batch: later ranks idle
    assert [r[1] - r[0] for r in rg] == [5, 0, 0, 0, 0, 0, 0, 0]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_bank_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in procs])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    po = oracle.port()
    D, tbw = 16, 0.05
    rates = [-0.1, 0.2, 0.33, -0.4, 0.05]
    ddc, _ = po.fastddc_init(tbw, D, 0.0)
    rng = np.random.default_rng(14)
    x = (rng.uniform(-1, 1, ddc.input_size * 7) + 1j * rng.uniform(-1, 1, ddc.input_size * 7)).astype(np.complex64)
    spec = po.fastddc_fwd_cc(x, ddc)
    got = {}
    for rank, first, nbl, outs in res:
        assert nbl == 3
        for k, o in enumerate(outs):
            got[first + k] = np.array(o, dtype=np.complex64)
    assert sorted(got) == list(range(len(rates)))
    for c, rate in enumerate(rates):
        d, _ = po.fastddc_init(tbw, D, rate)
        assert np.array_equal(got[c], po.fastddc_inv_cc(spec, d, po.fastddc_taps_fft(d, rate, D)))


def _sliced_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from csdr_amd import dist as cd
    import oracle
    cd.init("gloo")
    po = oracle.port()
    D, tbw = 16, 0.05
    rates = [-0.1, 0.2, 0.33, -0.4, 0.05]
    ddc, _ = po.fastddc_init(tbw, D, 0.0)
    n_blocks, max_blocks = 7, 8                                     # runs of 4 blocks: [0, 4) and [4, 7)
    rng = np.random.default_rng(15)
    x = (rng.uniform(-1, 1, ddc.input_size * n_blocks) + 1j * rng.uniform(-1, 1, ddc.input_size * n_blocks)).astype(np.complex64)
    zero = np.zeros((1, ddc.fft_size), np.complex64)

    def run(samples, first_block, n_glob):
        n_loc = (samples.size - ddc.overlap_length) // ddc.input_size
        spec = np.stack([po.fft_c2c(samples[b * ddc.input_size:b * ddc.input_size + ddc.fft_size], True) for b in range(n_loc)]) if n_loc else np.zeros((0, ddc.fft_size), np.complex64)
        outs = []
        for rate in rates:
            d, _ = po.fastddc_init(tbw, D, rate)
            tf = po.fastddc_taps_fft(d, rate, D)
            st = (0, 0.0)
            for _ in range(first_block):                             # the shift state in front of the run: data independent, so walked on zeros
                _, st = po.fastddc_inv_cc(zero, d, tf, status=st)
            y, _ = po.fastddc_inv_cc(spec, d, tf, status=st)
            outs.append(y)
        return outs

    first, outs = cd.bank_time_sliced(x if rank == 0 else None, n_blocks, max_blocks, ddc.input_size, ddc.overlap_length, len(rates), run, rank, world)
    cd.barrier()
    q.put((rank, first, [o.tolist() for o in outs]))


@pytest.mark.parametrize("world", [2, 3])
def test_gloo_time_sliced_bank_schedule(world):
    """The time-sliced bank (fftpath.hip: bank_submit_blocks / bank_collect_blocks; the default of csdr_amd_fastddc_bank_create_sharded) as a gloo model: runs of
    blocks per rank, every rank all channels with the shift state walked locally over the blocks in front of its run, outputs exchanged all-to-all and
    stitched in rank order -- bit-identical to the unsharded oracle."""
    import oracle
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + os.getpid() % 2000 + world
    procs = [ctx.Process(target=_sliced_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in procs])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    po = oracle.port()
    D, tbw = 16, 0.05
    rates = [-0.1, 0.2, 0.33, -0.4, 0.05]
    ddc, _ = po.fastddc_init(tbw, D, 0.0)
    rng = np.random.default_rng(15)
    x = (rng.uniform(-1, 1, ddc.input_size * 7) + 1j * rng.uniform(-1, 1, ddc.input_size * 7)).astype(np.complex64)
    spec = po.fastddc_fwd_cc(x, ddc)
    got = {}
    for rank, first, outs in res:
        for k, o in enumerate(outs):
            got[first + k] = np.array(o, dtype=np.complex64)
    assert sorted(got) == list(range(len(rates)))
    for c, rate in enumerate(rates):
        d, _ = po.fastddc_init(tbw, D, rate)
        assert np.array_equal(got[c], po.fastddc_inv_cc(spec, d, po.fastddc_taps_fft(d, rate, D)))
