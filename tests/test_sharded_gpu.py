"""The multi-rank fastddc bank (ddcd_old.cpp:238-252, 474-492: one forward transform feeding many inverse halves) RUN at world 2 / 4 / 8
on the one GPU a box has: `world` rank threads joined by the library's loopback communicator (comm.cpp: every exchange a stream-ordered device copy), each
driving csdr_amd_fastddc_bank_create_sharded_by + submit / collect / finish exactly as one process per GPU would over RCCL.  Both ways of dividing the work
are covered: time slices (the default: blocks dealt to the ranks, outputs exchanged all-to-all) and channel slices (spectra all-gathered).  BASELINE config
4's geometry, all 256 channels; batches of 64, 5, 1 and 64 blocks (runs of ceil(64 / world) blocks: with 5 and 1 most ranks have nothing to transform and
must still carry their channels' states and take part in the exchange), a retune between batches.  Every channel against the single-GPU bank (2e-6: the
same kernels, another summation context for the second forward pass), a spread of channels against the CPU oracle (1e-5)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import verify_configs as vc  # noqa: E402

pytestmark = pytest.mark.gpu
c64 = np.complex64
f32 = np.float32
TBW, D, NCH = 0.001, 256, 256
SCHEDULE = [64, 5, 1, 64]
RETUNES = {2: [(37, 0.123), (200, -0.3711)]}           # before the third batch
ORACLE_CHANNELS = [0, 31, 32, 100, 127, 128, 223, 224, 255]


@pytest.fixture(scope="module")
def gpu():
    import torch  # noqa: F401
    import csdr_amd
    ctx = csdr_amd.Context(0)
    assert ctx.arch().startswith("gfx950")
    yield ctx
    ctx.close()


@pytest.fixture(scope="module")
def case(gpu, port):
    """input, the single-GPU bank's outputs and the oracle's (computed once for all worlds / modes)"""
    ddc, err = gpu.fastddc_init(TBW, D, 0.0)
    assert err == 0 and ddc.fft_size == 65536 and ddc.input_size >= ddc.overlap_length
    nb = sum(SCHEDULE)
    rng = np.random.default_rng(83)
    x = (rng.uniform(-1, 1, ddc.input_size * nb) + 1j * rng.uniform(-1, 1, ddc.input_size * nb)).astype(c64)
    rates = vc.c4_rates(NCH)
    single = gpu.fastddc_bank(x, TBW, D, rates, schedule=SCHEDULE, retunes=RETUNES)
    _, want = vc.fastddc_oracle_channels(x, TBW, D, rates, ORACLE_CHANNELS)
    for c in ORACLE_CHANNELS:                             # the reference point itself is oracle-clean
        assert single[c].size == want[c].size and vc.relrms(single[c], want[c]) < 1e-5
    return x, rates, single, want


def _check(outs, single, want):
    assert len(outs) == NCH and all(o is not None for o in outs)
    worst = 0.0
    for c in range(NCH):
        assert outs[c].size == single[c].size, "channel %d: %d samples, single GPU %d" % (c, outs[c].size, single[c].size)
        worst = max(worst, vc.relrms(outs[c], single[c]))
    assert worst < 2e-6, "worst channel vs the single-GPU bank: %g" % worst
    for c in ORACLE_CHANNELS:
        assert vc.relrms(outs[c], want[c]) < 1e-5, "channel %d vs oracle" % c


@pytest.mark.parametrize("world", [2, 4, 8])
@pytest.mark.parametrize("mode", ["blocks", "channels"])
def test_sharded_bank_loopback(gpu, case, mode, world):
    import csdr_amd
    x, rates, single, want = case
    outs = csdr_amd.sharded_bank_loopback(world, x, TBW, D, rates, SCHEDULE, mode=mode, retunes=RETUNES, pipelined=True)
    _check(outs, single, want)


def test_retune_while_a_batch_is_staged(gpu, case):
    """A retune that arrives between submit(k) and collect(k) leaves batch k alone and applies from batch k + 1 on (csdr.c:2329-2376: the new rate between two reads).
    The time-sliced bank computes a batch's chain tables at its collect and therefore holds such a retune back: same streams as the single-GPU bank retuned between
    the two batches.  Every other bank has part of the staged batch's tables fixed already and refuses the call until the batch is collected."""
    import ctypes as C
    import csdr_amd
    x, rates, single, want = case
    outs = csdr_amd.sharded_bank_loopback(2, x, TBW, D, rates, SCHEDULE, mode="blocks", retunes=RETUNES, pipelined=True, retune_while_staged=True)
    _check(outs, single, want)
    L = gpu.L
    ddc, _ = gpu.fastddc_init(TBW, D, 0.0)
    r4 = np.ascontiguousarray(rates[:4], np.float32)
    bank = L.csdr_amd_fastddc_bank_create(gpu.h, TBW, D, r4.ctypes.data_as(C.c_void_p), 4, 1, 2)
    assert bank
    try:
        di = gpu.upload(x[:2 * ddc.input_size])
        assert L.csdr_amd_fastddc_bank_set_rate(bank, 1, 0.111) == 0                      # nothing staged: applied
        assert L.csdr_amd_fastddc_bank_submit(bank, di.ptr, 2) == 0
        assert L.csdr_amd_fastddc_bank_set_rate(bank, 1, 0.222) < 0 and "staged" in gpu.err()
        pitch = L.csdr_amd_fastddc_bank_max_output(bank, 2) + 8
        do = gpu.alloc(8 * 4 * pitch)
        assert L.csdr_amd_fastddc_bank_collect(bank, do.ptr, pitch, None) == 0
        assert L.csdr_amd_fastddc_bank_set_rate(bank, 1, 0.222) == 0
        gpu.sync()
    finally:
        L.csdr_amd_fastddc_bank_destroy(bank)


def test_a_held_back_retune_does_not_override_a_newer_one(gpu, case):
    """time-sliced bank, retune A of a channel issued while batch k is staged (held back for collect(k + 1)), then -- after collect(k), nothing
    staged -- retune B of the same channel, applied at once.  The queue must not replay A on top of B at collect(k + 1): the streams are those of the single-GPU bank
    retuned to B between the two batches."""
    import csdr_amd
    x, rates, single, want = case
    outs = csdr_amd.sharded_bank_loopback(2, x, TBW, D, rates, SCHEDULE, mode="blocks", retunes=RETUNES, pipelined=False, retune_while_staged=True, superseded_retune=True)
    _check(outs, single, want)


def test_sharded_bank_loopback_unpipelined_and_odd_world(gpu, case):
    """three ranks (256 channels do not divide: slices of 86 / 85 / 85; runs of 22 blocks), every batch submitted and collected in turn"""
    import csdr_amd
    x, rates, single, want = case
    for mode in ("blocks", "channels"):
        outs = csdr_amd.sharded_bank_loopback(3, x, TBW, D, rates, SCHEDULE, mode=mode, retunes=RETUNES, pipelined=False)
        _check(outs, single, want)


def test_time_sliced_bank_local_ingest(gpu, case):
    """csdr_amd_fastddc_bank_submit_local: every rank is handed its own run of each batch (overlap in front) -- no input exchange at all"""
    import csdr_amd
    x, rates, single, want = case
    outs = csdr_amd.sharded_bank_loopback(4, x, TBW, D, rates, SCHEDULE, mode="blocks", retunes=RETUNES, pipelined=True, local_input=True)
    _check(outs, single, want)


@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_comm_selftest_over_the_loopback_transport(gpu, world):
    """csdr_amd_comm_selftest (VERDICT r4 next #5a: the first run on a new transport must diagnose itself): ring, all-gather, broadcast and the two-communicator /
    two-stream ring of rank-stamped buffers, every byte checked, one report line per rank.  Here over the loopback transport at world 1 / 2 / 4 / 8 (and csdr_amd_comm_dup
    with it); over RCCL the same function is the first thing bench_fastddc.py --gpus N calls."""
    import threading
    import csdr_amd
    L = gpu.L
    grp = L.csdr_amd_loopback_create(world)
    assert grp
    res = [None] * world

    def rank_main(r):
        ctx = csdr_amd.Context(0)
        comm = L.csdr_amd_comm_create_loopback(ctx.h, grp, r)
        rep = C.create_string_buffer(1024)
        rc = L.csdr_amd_comm_selftest(comm, 1 << 18, rep, 1024)
        res[r] = (rc, rep.value.decode(), ctx.err() if rc else "")
        L.csdr_amd_comm_destroy(comm); ctx.close()

    ts = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    [t.start() for t in ts]; [t.join() for t in ts]
    L.csdr_amd_loopback_destroy(grp)
    for r in range(world):
        rc, line, err = res[r]
        assert rc == 0, (r, line, err)
        assert line.startswith("rank %d/%d" % (r, world)) and "MISMATCH" not in line and "broadcast ok" in line and "all_gather" in line, line


@pytest.mark.parametrize("world", [2, 3])
def test_bench_all_modes_sweep_over_loopback_ranks(gpu, world):
    """bench_fastddc.py's N > 1 sweep (all_modes: the communicator self test, then both shard modes x {cf32, s16, u8}, each checked against the unsharded bank and
    timed) -- the code the first multi-GPU run executes -- driven here by `world` rank threads over the loopback transport on the one GPU of the box."""
    import argparse
    import threading
    import csdr_amd
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench_fastddc as bf
    L = gpu.L
    grp = L.csdr_amd_loopback_create(world)
    assert grp
    args = argparse.Namespace(tbw=TBW, decimation=D, channels=NCH, blocks=6, steps=2, warmup=1)
    bar = threading.Barrier(world)
    slots = [0.0] * world
    res = [None] * world; errors = []

    def make_red(r, fn):
        def red(v):
            slots[r] = float(v); bar.wait(); out = fn(slots); bar.wait(); return out
        return red

    def rank_main(r):
        ctx = None
        try:
            ctx = csdr_amd.Context(0)
            comm = L.csdr_amd_comm_create_loopback(ctx.h, grp, r)
            res[r] = bf.all_modes(ctx, L, comm, r, world, args, bar.wait, make_red(r, max), make_red(r, min))
            L.csdr_amd_comm_destroy(comm)
        except BaseException as e:  # noqa: BLE001
            errors.append((r, repr(e))); L.csdr_amd_loopback_abort(grp); bar.abort()
        finally:
            if ctx is not None:
                ctx.close()

    ts = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    [t.start() for t in ts]; [t.join() for t in ts]
    L.csdr_amd_loopback_destroy(grp)
    assert not errors, errors
    r0 = res[0]
    assert r0["selftest_ok"] and "MISMATCH" not in r0["selftest_rank0"]
    assert [(m["shard"], m["input_format"]) for m in r0["modes"]] == [(a, b) for a in ("channels", "blocks") for b in ("cf32", "s16", "u8")]
    for m in r0["modes"]:
        assert m["verify"]["ok"] and m["verify"]["max_rel_rms_over_ranks"] < 2e-6 and m["value"] > 0, m
        assert m["blocks_per_step"] == (6 * world if m["shard"] == "blocks" else 6)


def test_comm_selftest_on_a_one_rank_rccl_communicator(gpu):
    """the same over RCCL itself with the one rank a box has (ncclCommInitRank twice: the parent and csdr_amd_comm_dup's)"""
    L = gpu.L
    uid = C.create_string_buffer(128)
    assert L.csdr_amd_comm_unique_id(uid) == 0, gpu.err()
    comm = L.csdr_amd_comm_create(gpu.h, uid, 0, 1)
    assert comm, gpu.err()
    try:
        rep = C.create_string_buffer(1024)
        assert L.csdr_amd_comm_selftest(comm, 1 << 16, rep, 1024) == 0, gpu.err()
        assert "rccl" in rep.value.decode() and "MISMATCH" not in rep.value.decode()
    finally:
        L.csdr_amd_comm_destroy(comm)


def test_loopback_broadcast_and_failure(gpu):
    """the transport itself: a broadcast over four rank threads; a rank that never arrives fails the others instead of hanging them"""
    import threading
    import csdr_amd
    L = gpu.L
    W = 4
    grp = L.csdr_amd_loopback_create(W)
    assert grp
    got = [None] * W

    def rank_main(r):
        ctx = csdr_amd.Context(0)
        comm = L.csdr_amd_comm_create_loopback(ctx.h, grp, r)
        buf = ctx.upload(np.full(1024, 7 if r == 2 else r, np.int32))
        rc = L.csdr_amd_comm_broadcast(comm, buf.ptr, 4096, 2)
        got[r] = (rc, ctx.download(buf, np.int32, 1024))
        L.csdr_amd_comm_destroy(comm); ctx.close()

    ts = [threading.Thread(target=rank_main, args=(r,)) for r in range(W)]
    [t.start() for t in ts]; [t.join() for t in ts]
    for r in range(W):
        assert got[r][0] == 0 and (got[r][1] == 7).all()
    L.csdr_amd_loopback_abort(grp)                        # from now on every rendezvous fails at once
    ctx = csdr_amd.Context(0)
    comm = L.csdr_amd_comm_create_loopback(ctx.h, grp, 0)
    buf = ctx.alloc(4096)
    assert L.csdr_amd_comm_broadcast(comm, buf.ptr, 4096, 0) < 0 and "did not arrive" in ctx.err()
    L.csdr_amd_comm_destroy(comm); ctx.close()
    L.csdr_amd_loopback_destroy(grp)


def test_loopback_local_error_does_not_stall_the_group(gpu):
    """a rank whose group fails locally (two sends to one peer) clears what it published and fails the others' rendezvous AT ONCE -- not after the 60 s of a
    rank that never arrives; the group stays broken for later calls"""
    import threading
    import time
    import ctypes as C
    import csdr_amd
    L = gpu.L
    L.csdr_amd_debug_comm_exchange.restype = C.c_int
    L.csdr_amd_debug_comm_exchange.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int]
    grp = L.csdr_amd_loopback_create(2)
    got = [None, None]

    def rank_main(r):
        ctx = csdr_amd.Context(0)
        comm = L.csdr_amd_comm_create_loopback(ctx.h, grp, r)
        a = ctx.upload(np.full(256, r + 1, np.float32)); b = ctx.alloc(1024)
        t0 = time.perf_counter()
        rc1 = L.csdr_amd_debug_comm_exchange(comm, a.ptr, b.ptr, 256, 1 - r, 1)          # a good group first
        v = ctx.download(b, np.float32, 256).copy()
        rc2 = L.csdr_amd_debug_comm_exchange(comm, a.ptr, b.ptr, 256, 1 - r, 2 if r == 0 else 1)
        e2 = ctx.err()
        rc3 = L.csdr_amd_debug_comm_exchange(comm, a.ptr, b.ptr, 256, 1 - r, 1)
        got[r] = (rc1, v, rc2, e2, rc3, time.perf_counter() - t0)
        L.csdr_amd_comm_destroy(comm); ctx.close()

    ts = [threading.Thread(target=rank_main, args=(r,)) for r in range(2)]
    [t.start() for t in ts]; [t.join() for t in ts]
    for r in range(2):
        rc1, v, rc2, e2, rc3, dt = got[r]
        assert rc1 == 0 and (v == 2 - r).all()
        assert rc2 < 0 and rc3 < 0 and dt < 10.0, got[r]
    assert "two sends" in got[0][3] and "did not arrive" in got[1][3]
    L.csdr_amd_loopback_destroy(grp)


def test_null_transport_times_one_rank(gpu):
    """csdr_amd_comm_create_null: rank 3 of 8 with no peers -- the calls of a world-8 schedule go through (what bench_fastddc.py --emulate-world times)"""
    L = gpu.L
    ddc, _ = gpu.fastddc_init(TBW, D, 0.0)
    rates = vc.c4_rates(NCH)
    for mode in (0, 1):
        comm = L.csdr_amd_comm_create_null(gpu.h, 3, 8)
        assert comm and (L.csdr_amd_comm_rank(comm), L.csdr_amd_comm_world(comm)) == (3, 8)
        bank = L.csdr_amd_fastddc_bank_create_sharded_by(gpu.h, TBW, D, rates.ctypes.data_as(C.c_void_p), NCH, 2, 64, comm, mode)
        assert bank, gpu.err()
        f0 = C.c_int(); c0 = C.c_int(); L.csdr_amd_fastddc_bank_channel_slice(bank, C.byref(f0), C.byref(c0))
        assert (f0.value, c0.value) == (96, 32) and L.csdr_amd_fastddc_bank_shard_mode(bank) == mode
        di = gpu.alloc(8 * 64 * ddc.input_size)
        pitch = L.csdr_amd_fastddc_bank_max_output(bank, 64) + 8
        do = gpu.alloc(8 * 32 * pitch)
        for _ in range(3):
            assert L.csdr_amd_fastddc_bank_process(bank, di.ptr, 64, do.ptr, pitch, None) == 0, gpu.err()
        gpu.sync()
        L.csdr_amd_fastddc_bank_destroy(bank); L.csdr_amd_comm_destroy(comm)


def test_time_sliced_bank_at_the_emulated_size(gpu, port):
    """bench_fastddc.py --emulate-world 8's configuration run for real: 8 rank threads, batches of 512 blocks (64 per rank: every rank's pipeline at the shape the
    single-GPU bench times, the chain walked over 448 foreign blocks, the 3.7-MB-per-peer output exchange), two batches so that state and overlap cross a batch
    boundary.  All 256 channels against the single-GPU bank fed the same stream in calls of 64 blocks, five against the oracle."""
    import csdr_amd
    ddc, _ = gpu.fastddc_init(TBW, D, 0.0)
    nb = 1024
    rng = np.random.default_rng(512)
    x = (rng.uniform(-1, 1, ddc.input_size * nb) + 1j * rng.uniform(-1, 1, ddc.input_size * nb)).astype(c64)
    rates = vc.c4_rates(NCH)
    outs = csdr_amd.sharded_bank_loopback(8, x, TBW, D, rates, [512, 512], mode="blocks", pipelined=True)
    single = gpu.fastddc_bank(x, TBW, D, rates, blocks_per_call=64)
    worst = 0.0
    for c in range(NCH):
        assert outs[c].size == single[c].size, "channel %d" % c
        worst = max(worst, vc.relrms(outs[c], single[c]))
    assert worst < 2e-6, worst
    check = [0, 77, 128, 200, 255]
    _, want = vc.fastddc_oracle_channels(x, TBW, D, rates, check)
    for c in check:
        assert outs[c].size == want[c].size and vc.relrms(outs[c], want[c]) < 1e-5, "channel %d vs oracle" % c


@pytest.mark.parametrize("fmt", ["s16", "u8"])
def test_bank_integer_ingest_is_bit_equal_to_the_converter_in_front(gpu, fmt):
    """csdr_amd_fastddc_bank_process_s16 / _u8: the wideband stream as integer IQ pairs, converted inside the forward transform (README.md:66-87: the reference
    runs convert_s16_f / convert_u8_f in front of fastddc_fwd_cc).  Against the device converter followed by the complexf entry point, in calls of 24 / 3 / 37
    blocks (the carried overlap is complexf either way): the same bits on all 256 channels."""
    ddc, _ = gpu.fastddc_init(TBW, D, 0.0)
    nb = 64
    rng = np.random.default_rng(91)
    if fmt == "s16":
        raw = rng.integers(-32768, 32768, 2 * nb * ddc.input_size, dtype=np.int16)
        xf = gpu.convert_s16_f(raw).view(c64)
    else:
        raw = rng.integers(0, 256, 2 * nb * ddc.input_size, dtype=np.uint8)
        xf = gpu.convert_u8_f(raw).view(c64)
    rates = vc.c4_rates(NCH)
    a = gpu.fastddc_bank(xf, TBW, D, rates, schedule=[24, 3, 37])
    b = gpu.fastddc_bank(raw, TBW, D, rates, schedule=[24, 3, 37])
    for c in range(NCH):
        assert a[c].size == b[c].size and np.array_equal(a[c].view(np.uint32), b[c].view(np.uint32)), "channel %d" % c


@pytest.mark.parametrize("mode,fmt", [("channels", "s16"), ("blocks", "s16"), ("channels", "u8"), ("blocks", "u8")])
def test_sharded_bank_scatters_raw_integers(gpu, mode, fmt):
    """A sharded bank fed integer samples on rank 0: the RAW integers are what crosses the links (4 or 2 bytes per sample instead of 8: the root's egress is what
    bounds the scaling of a single-ingest bank), every rank converts its run inside its forward transform.  World 4 over the loopback communicator, batches of
    40 / 5 / 19 blocks, every channel against the one-GPU bank on the same integers (2e-6: another summation context of the second forward pass)."""
    import csdr_amd
    ddc, _ = gpu.fastddc_init(TBW, D, 0.0)
    sched = [40, 5, 19]
    rng = np.random.default_rng(92)
    n = 2 * sum(sched) * ddc.input_size
    raw = rng.integers(-32768, 32768, n, dtype=np.int16) if fmt == "s16" else rng.integers(0, 256, n, dtype=np.uint8)
    rates = vc.c4_rates(NCH)
    single = gpu.fastddc_bank(raw, TBW, D, rates, schedule=sched)
    outs = csdr_amd.sharded_bank_loopback(4, raw, TBW, D, rates, sched, mode=mode, pipelined=True)
    worst = 0.0
    for c in range(NCH):
        assert outs[c].size == single[c].size, "channel %d" % c
        worst = max(worst, vc.relrms(outs[c], single[c]))
    assert worst < 2e-6, worst
    if mode == "blocks" and fmt == "s16":               # every rank handed its own run (s16: the zeros in front of the stream are exact)
        outs = csdr_amd.sharded_bank_loopback(4, raw, TBW, D, rates, sched, mode=mode, pipelined=True, local_input=True)
        assert max(vc.relrms(outs[c], single[c]) for c in range(NCH)) < 2e-6
