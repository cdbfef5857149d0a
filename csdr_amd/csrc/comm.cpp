// comm.cpp -- the multi-GPU exchange of libcsdr_amd.so: one process per GPU, RCCL over xGMI.
//
// The only part of the hot path with a real exchange step is the fastddc bank (csdr_amd_fastddc_bank_create_sharded): the wideband input lives on
// rank 0, every rank owns a slice of the channels.  What the reference does there is one `csdr fastddc_fwd_cc | nmux` feeding one `csdr fastddc_inv_cc`
// process per client over TCP (ddcd_old.cpp:238-252, 474-492).  Here, per batch of blocks (fastddc_mfma.hip, ddc_mfma_submit):
//   1. scatter: rank 0 sends rank g the samples of ITS windows (blocks [g nbl, (g+1) nbl) plus the overlap in front): world-1 point-to-point
//      transfers in one group, each over its own xGMI link (the mesh is point-to-point: a root broadcast would push the whole input through
//      every one of the root's links: 7 x the bytes per link);
//   2. every rank transforms its blocks (1 / world of the forward work) straight into the fold's layout;
//   3. all-gather of the transposed spectra over the full mesh (every link carries 1 / world of the spectrum, in both directions);
//   4. every rank folds / inverse-transforms its channel slice.
// Per link and batch that is (1 + 1) / world of the data instead of 1: the exchange ceiling rises from one link's rate to ~world/2 x it, and
// it runs on a side stream under the previous batch's fold (submit / collect).
//
// RCCL is loaded on demand (dlopen of librccl.so.1: the copy torch already mapped when the caller is a torch process, the system one otherwise), so
// the single-GPU library and the CLI do not depend on it.
#include "fastddc.hpp"
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <vector>

using namespace csdr_amd;

namespace {

typedef int ncclResult_t;
typedef void *ncclComm_t;
struct ncclUniqueId { char internal[128]; };
enum { ncclFloat32 = 7 };                                             // rccl.h ncclDataType_t: ncclFloat = 7

struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl; std::mutex g_rccl_mu;

int load_rccl()
{
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.lib) return 0;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *n : names) { h = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (h) break; }
    if (!h) return fail_msg(-6, "cannot load RCCL (librccl.so.1): %s", dlerror());
#define SYM(field, name) do { *(void **)(&g_rccl.field) = dlsym(h, name); if (!g_rccl.field) { dlclose(h); return fail_msg(-6, "RCCL symbol %s missing", name); } } while (0)
    SYM(GetUniqueId, "ncclGetUniqueId"); SYM(CommInitRank, "ncclCommInitRank"); SYM(CommDestroy, "ncclCommDestroy");
    SYM(GroupStart, "ncclGroupStart"); SYM(GroupEnd, "ncclGroupEnd"); SYM(Send, "ncclSend"); SYM(Recv, "ncclRecv");
    SYM(AllGather, "ncclAllGather"); SYM(Broadcast, "ncclBroadcast"); SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    g_rccl.lib = h;
    return 0;
}

int nccl_fail(ncclResult_t r, const char *what) { return fail_msg(-6, "RCCL error %d (%s) in %s", (int)r, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?", what); }
#define CSDR_NCCL(expr) do { ncclResult_t r__ = (expr); if (r__ != 0) return nccl_fail(r__, #expr); } while (0)

} // namespace

// ------------------------------------------------------------------ loopback transport: the ranks of a communicator inside ONE process
// world "ranks" = world host threads (one context / stream each, on any devices -- in the tests all on device 0), the exchange as device-to-device copies
// ordered by events.  It exists so that the multi-rank branches of the channelizer (scatter, local transforms, all-gather, all-to-all: ddc_mfma_submit,
// fftpath.hip) run for real on the ONE GPU a box has; it implements the same function table as the RCCL transport and nothing in the data path knows
// which one it talks to.  Semantics: every rank calls group_end / all_gather / broadcast the same number of times in the same order (each is a host
// rendezvous of the rank threads followed by stream-ordered copies); a receive pulls from the sender's buffer on the RECEIVER's stream after the sender's
// "ready" event, and the sender's stream then waits for the receiver's "done" event, so a buffer is never reused under a copy.
struct csdr_amd_loopback {
    int world;
    std::mutex mu; std::condition_variable cv; int arrived = 0; unsigned long gen = 0; bool broken = false;
    struct Msg { const void *buf; size_t n; bool valid, taken; };
    std::vector<Msg> box;                     // [src * world + dst]: the sends published by the current group
    std::vector<hipEvent_t> ready, done;      // [src * world + dst]
    std::vector<void *> ag_buf; std::vector<hipEvent_t> ag_ready, ag_done;      // collectives: per rank
    int attached = 0;
    // host rendezvous of the rank threads; fails (instead of hanging the box) when a rank never arrives
    int barrier()
    {
        std::unique_lock<std::mutex> lk(mu);
        if (broken) return -6;
        const unsigned long g = gen;
        if (++arrived == world) { arrived = 0; gen++; cv.notify_all(); return 0; }
        if (!cv.wait_for(lk, std::chrono::seconds(60), [&] { return gen != g || broken; }) || broken) { broken = true; cv.notify_all(); return -6; }
        return 0;
    }
};

struct csdr_amd_comm {
    csdr_amd_ctx *ctx; ncclComm_t comm; DdcComm ddc;
    csdr_amd_loopback *loop = nullptr; bool null_transport = false;
    struct Op { bool send; void *buf; size_t n; int peer; hipStream_t st; };
    std::vector<Op> ops; bool in_group = false;
};

namespace {
ncclComm_t nc(const DdcComm *c) { return ((csdr_amd_comm *)c->impl)->comm; }
int c_group_start(const DdcComm *) { CSDR_NCCL(g_rccl.GroupStart()); return 0; }
int c_group_end(const DdcComm *) { CSDR_NCCL(g_rccl.GroupEnd()); return 0; }
int c_send(const DdcComm *c, const void *buf, size_t n, int peer, hipStream_t st) { CSDR_NCCL(g_rccl.Send(buf, n, ncclFloat32, peer, nc(c), st)); return 0; }
int c_recv(const DdcComm *c, void *buf, size_t n, int peer, hipStream_t st) { CSDR_NCCL(g_rccl.Recv(buf, n, ncclFloat32, peer, nc(c), st)); return 0; }
int c_all_gather(const DdcComm *c, void *all, size_t n, hipStream_t st)
{
    if (c->world == 1) return 0;                                      // in place: nothing moves
    CSDR_NCCL(g_rccl.AllGather((const char *)all + (size_t)c->rank * n * sizeof(float), all, n, ncclFloat32, nc(c), st));
    return 0;
}
} // namespace

namespace {
// ---- loopback implementation of the DdcComm table
csdr_amd_comm *lc(const DdcComm *c) { return (csdr_amd_comm *)c->impl; }
int l_barrier(csdr_amd_loopback *g) { const int rc = g->barrier(); return rc ? fail_msg(-6, "loopback communicator: a rank did not arrive within 60 s, or gave up after an error of its own (all ranks must make the same exchange calls)") : 0; }
int l_group_start(const DdcComm *c) { csdr_amd_comm *m = lc(c); m->ops.clear(); m->in_group = true; return 0; }
int l_send(const DdcComm *c, const void *buf, size_t n, int peer, hipStream_t st)
{
    csdr_amd_comm *m = lc(c);
    if (!m->in_group || peer < 0 || peer >= c->world || peer == c->rank) return fail_msg(-3, "loopback send: outside a group or bad peer %d", peer);
    m->ops.push_back({true, const_cast<void *>(buf), n, peer, st}); return 0;
}
int l_recv(const DdcComm *c, void *buf, size_t n, int peer, hipStream_t st)
{
    csdr_amd_comm *m = lc(c);
    if (!m->in_group || peer < 0 || peer >= c->world || peer == c->rank) return fail_msg(-3, "loopback recv: outside a group or bad peer %d", peer);
    m->ops.push_back({false, buf, n, peer, st}); return 0;
}
int l_group_end(const DdcComm *c)
{
    // A local error (two sends to one peer, a HIP call that fails) must not leave the other rank threads waiting 60 s at a rendezvous this rank never reaches, nor box
    // entries marked valid for the next group: this rank's entries are cleared, the group is marked broken with a wake-up (what csdr_amd_loopback_abort does), and
    // the first error is the one reported.
    csdr_amd_comm *m = lc(c); csdr_amd_loopback *g = m->loop;
    const int me = c->rank, W = c->world;
    m->in_group = false;
    int err = 0;
    auto hip = [&](hipError_t e, const char *what) { if (e != hipSuccess && !err) err = ::csdr_amd::fail(e, what, __FILE__, __LINE__); return e == hipSuccess; };
    auto give_up = [&]() {
        for (const auto &op : m->ops) if (op.send) g->box[me * W + op.peer].valid = false;
        m->ops.clear();
        { std::lock_guard<std::mutex> lk(g->mu); g->broken = true; }
        g->cv.notify_all();
        return err;
    };
    for (const auto &op : m->ops) if (op.send && !err) {              // publish: what, and when it is ready
        const int id = me * W + op.peer;
        if (g->box[id].valid) { err = fail_msg(-3, "loopback: two sends to rank %d in one group", op.peer); break; }
        if (!hip(hipEventRecord(g->ready[id], op.st), "hipEventRecord(ready)")) break;
        g->box[id] = {op.buf, op.n, true, false};
    }
    if (err) return give_up();
    int rc = l_barrier(g); if (rc) { err = rc; return give_up(); }
    int bad = 0;
    for (const auto &op : m->ops) if (!op.send && !err) {             // pull on the receiver's stream
        const int id = op.peer * W + me;
        csdr_amd_loopback::Msg &msg = g->box[id];
        if (!msg.valid || msg.n != op.n) { bad = 1; continue; }
        if (!hip(hipStreamWaitEvent(op.st, g->ready[id], 0), "hipStreamWaitEvent(ready)") ||
            !hip(hipMemcpyAsync(op.buf, msg.buf, op.n * sizeof(float), hipMemcpyDeviceToDevice, op.st), "hipMemcpyAsync(loopback)") ||
            !hip(hipEventRecord(g->done[id], op.st), "hipEventRecord(done)")) break;
        msg.taken = true;
    }
    if (err) return give_up();
    rc = l_barrier(g); if (rc) { err = rc; return give_up(); }
    for (const auto &op : m->ops) if (op.send) {                      // the source may be reused only after the copy
        const int id = me * W + op.peer;
        if (!g->box[id].taken) bad = 1; else hip(hipStreamWaitEvent(op.st, g->done[id], 0), "hipStreamWaitEvent(done)");
        g->box[id].valid = false;
    }
    m->ops.clear();
    if (err) return err;
    if (bad) return fail_msg(-3, "loopback: a send / recv of rank %d had no matching partner (or the sizes differ)", me);
    return 0;
}
// collective copy: every rank's piece (root < 0: rank s owns [s n, (s + 1) n) of its buffer; root >= 0: the root owns [0, n)) reaches every other rank
int l_collect(csdr_amd_comm *m, void *buf, size_t n_bytes, int root, hipStream_t st)
{
    csdr_amd_loopback *g = m->loop; const int me = m->ddc.rank, W = m->ddc.world;
    g->ag_buf[me] = buf;
    CSDR_HIP(hipEventRecord(g->ag_ready[me], st));
    int rc = l_barrier(g); if (rc) return rc;
    for (int s = 0; s < W; s++) {
        if (s == me || (root >= 0 && s != root)) continue;
        const size_t off = root >= 0 ? 0 : (size_t)s * n_bytes;
        CSDR_HIP(hipStreamWaitEvent(st, g->ag_ready[s], 0));
        CSDR_HIP(hipMemcpyAsync((char *)buf + off, (const char *)g->ag_buf[s] + off, n_bytes, hipMemcpyDeviceToDevice, st));
    }
    CSDR_HIP(hipEventRecord(g->ag_done[me], st));
    rc = l_barrier(g); if (rc) return rc;
    for (int s = 0; s < W; s++) if (s != me && (root < 0 || me == root)) CSDR_HIP(hipStreamWaitEvent(st, g->ag_done[s], 0));      // my piece has been read everywhere
    return 0;                                                         // (the next collective re-records the events only behind its own first rendezvous: every wait above is queued by then)
}
int l_all_gather(const DdcComm *c, void *all, size_t n, hipStream_t st) { return c->world == 1 ? 0 : l_collect(lc(c), all, n * sizeof(float), -1, st); }
// ---- no transport at all: ONE rank of a world-N schedule timed alone (bench_fastddc.py --emulate-world); nothing moves, results are meaningless
int n_ok(const DdcComm *) { return 0; }
int n_send(const DdcComm *, const void *, size_t, int, hipStream_t) { return 0; }
int n_recv(const DdcComm *, void *, size_t, int, hipStream_t) { return 0; }
int n_all_gather(const DdcComm *, void *, size_t, hipStream_t) { return 0; }
} // namespace

namespace csdr_amd { const DdcComm *csdr_amd_comm_ddc(csdr_amd_comm *c) { return c ? &c->ddc : nullptr; } }

extern "C" {

int csdr_amd_comm_unique_id(char id128[128])
{
    int rc = load_rccl(); if (rc) return rc;
    ncclUniqueId id; CSDR_NCCL(g_rccl.GetUniqueId(&id));
    memcpy(id128, id.internal, 128);
    return 0;
}

csdr_amd_comm *csdr_amd_comm_create(csdr_amd_ctx *ctx, const char id128[128], int rank, int world)
{
    if (world < 1 || rank < 0 || rank >= world) { fail_msg(-3, "comm: bad rank %d of %d", rank, world); return nullptr; }
    if (load_rccl()) return nullptr;
    if (hipSetDevice(ctx->device) != hipSuccess) { fail_msg(-1, "hipSetDevice(%d) failed", ctx->device); return nullptr; }
    csdr_amd_comm *c = new csdr_amd_comm();
    c->ctx = ctx; c->comm = nullptr;
    ncclUniqueId id; memcpy(id.internal, id128, 128);
    const ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, id, rank);
    if (r != 0) { nccl_fail(r, "ncclCommInitRank"); delete c; return nullptr; }
    c->ddc.rank = rank; c->ddc.world = world; c->ddc.impl = c;
    c->ddc.group_start = c_group_start; c->ddc.group_end = c_group_end; c->ddc.send = c_send; c->ddc.recv = c_recv; c->ddc.all_gather = c_all_gather;
    return c;
}

csdr_amd_loopback *csdr_amd_loopback_create(int world)
{
    if (world < 1 || world > 64) { fail_msg(-3, "loopback: bad world size %d", world); return nullptr; }
    csdr_amd_loopback *g = new csdr_amd_loopback();
    g->world = world;
    g->box.assign((size_t)world * world, {nullptr, 0, false, false});
    g->ready.assign((size_t)world * world, nullptr); g->done.assign((size_t)world * world, nullptr);
    g->ag_buf.assign(world, nullptr); g->ag_ready.assign(world, nullptr); g->ag_done.assign(world, nullptr);
    bool ok = true;
    for (auto *v : {&g->ready, &g->done, &g->ag_ready, &g->ag_done}) for (auto &e : *v) ok = ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
    if (!ok) { fail_msg(-1, "loopback: hipEventCreate failed"); csdr_amd_loopback_destroy(g); return nullptr; }
    return g;
}

void csdr_amd_loopback_destroy(csdr_amd_loopback *g)
{
    if (!g) return;
    for (auto *v : {&g->ready, &g->done, &g->ag_ready, &g->ag_done}) for (auto &e : *v) if (e) (void)hipEventDestroy(e);
    delete g;
}

/* a rank thread died: fail the others' rendezvous now instead of after the 60 s limit */
void csdr_amd_loopback_abort(csdr_amd_loopback *g)
{
    if (!g) return;
    std::lock_guard<std::mutex> lk(g->mu);
    g->broken = true; g->cv.notify_all();
}

csdr_amd_comm *csdr_amd_comm_create_loopback(csdr_amd_ctx *ctx, csdr_amd_loopback *g, int rank)
{
    if (!g || rank < 0 || rank >= g->world) { fail_msg(-3, "loopback comm: bad rank %d", rank); return nullptr; }
    csdr_amd_comm *c = new csdr_amd_comm();
    c->ctx = ctx; c->comm = nullptr; c->loop = g;
    c->ddc.rank = rank; c->ddc.world = g->world; c->ddc.impl = c;
    c->ddc.group_start = l_group_start; c->ddc.group_end = l_group_end; c->ddc.send = l_send; c->ddc.recv = l_recv; c->ddc.all_gather = l_all_gather;
    return c;
}

csdr_amd_comm *csdr_amd_comm_create_null(csdr_amd_ctx *ctx, int rank, int world)
{
    if (world < 1 || rank < 0 || rank >= world) { fail_msg(-3, "comm: bad rank %d of %d", rank, world); return nullptr; }
    csdr_amd_comm *c = new csdr_amd_comm();
    c->ctx = ctx; c->comm = nullptr; c->null_transport = true;
    c->ddc.rank = rank; c->ddc.world = world; c->ddc.impl = c;
    c->ddc.group_start = n_ok; c->ddc.group_end = n_ok; c->ddc.send = n_send; c->ddc.recv = n_recv; c->ddc.all_gather = n_all_gather;
    return c;
}

void csdr_amd_comm_destroy(csdr_amd_comm *c)
{
    if (!c) return;
    (void)hipStreamSynchronize(c->ctx->stream);
    if (c->comm) (void)g_rccl.CommDestroy(c->comm);
    delete c;
}

int csdr_amd_comm_rank(const csdr_amd_comm *c) { return c->ddc.rank; }
int csdr_amd_comm_world(const csdr_amd_comm *c) { return c->ddc.world; }

/* broadcast of a device buffer from `root` on the context's stream (bench / test plumbing: e.g. checking a sharded run against rank 0's data) */
int csdr_amd_comm_broadcast(csdr_amd_comm *c, void *dev_buf, size_t bytes, int root)
{
    if (c->ddc.world == 1 || c->null_transport) return 0;
    if (c->loop) return l_collect(c, dev_buf, bytes, root, c->ctx->stream);
    CSDR_NCCL(g_rccl.Broadcast(dev_buf, dev_buf, bytes, 1 /* ncclUint8 */, root, c->comm, c->ctx->stream));
    return 0;
}

// test hook (tests/test_sharded_gpu.py): one group with `n_sends` sends of n floats to `peer` and one receive from it -- n_sends = 2 is the local error whose
// handling l_group_end documents
int csdr_amd_debug_comm_exchange(csdr_amd_comm *c, const float *send_buf, float *recv_buf, size_t n, int peer, int n_sends)
{
    const DdcComm *d = &c->ddc;
    int rc = d->group_start(d); if (rc) return rc;
    for (int k = 0; k < n_sends; k++) { rc = d->send(d, send_buf, n, peer, c->ctx->stream); if (rc) return rc; }
    rc = d->recv(d, recv_buf, n, peer, c->ctx->stream); if (rc) return rc;
    return d->group_end(d);
}

} // extern "C"
