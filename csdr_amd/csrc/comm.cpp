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
#include <rccl/rccl.h>      // types and enum values only (the entry points are dlsym'ed): a change of RCCL's ABI is a compile error here, not a crash at the first N > 1 run
#include <dlfcn.h>
#include <errno.h>
#include <poll.h>
#include <sys/socket.h>
#include <sys/un.h>
#include <time.h>
#include <unistd.h>
#include <map>
#include <stdio.h>
#include <string.h>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <string>
#include <vector>
#include <stdlib.h>

using namespace csdr_amd;

namespace {

// every pointer has the type rccl.h declares for that entry point (decltype): argument lists, ncclDataType_t, ncclUniqueId and the enum values ncclFloat32 / ncclUint8
// come from the header the library was built against
static_assert(sizeof(ncclUniqueId) == 128, "csdr_amd_comm_unique_id hands out 128 bytes (include/csdr_amd.h)");
struct Rccl {
    void *lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
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
    // csdr_amd_comm_dup: a rank's dup joins the lowest-numbered child group it holds no communicator of (created by whoever comes first; every rank makes its dup /
    // destroy calls in the same order, so all ranks pick the same child); a destroyed dup gives its place back, so create / destroy cycles reuse the children
    std::vector<csdr_amd_loopback *> children; std::vector<std::vector<char>> child_in_use;      // [rank][child]
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
    csdr_amd_loopback *dup_parent = nullptr; int dup_index = -1;      // a loopback dup: the parent group and the child's index in it
    // ---- the inter-PROCESS transport (csdr_amd_comm_create_ipc): a unix socket per peer, peers' buffers mapped through HIP IPC
    bool ipc = false; std::string ipc_prefix; int ipc_dups = 0; int listen_fd = -1; std::vector<int> peer_fd;
    std::map<std::string, void *> ipc_open;                               // opened handles (a handle opens once per process)
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
// ---- ranks as PROCESSES on one box (csdr_amd_comm_create_ipc): the same table over unix sockets + HIP IPC.  RCCL refuses two ranks on one device
// (profiles/r4_rccl_two_ranks_one_gpu.txt), the loopback transport's ranks are threads of one process -- neither crosses the process boundary that `csdr fastddc_bank_cc`
// started once per GPU crosses (CSDR_AMD_RANK / _WORLD / _COMM_FILE, csdr_cli.cpp).  A test transport: every group is host synchronous.  Protocol of one group: a sender
// waits for its stream, then tells the peer {IPC handle of the allocation, offset, floats}; the receiver maps the handle (once), copies on ITS stream, waits, answers
// with one byte; the sender returns when every peer has answered (its buffer is free again).  All sends go out before any receive is served: the messages are small,
// the sockets buffer them, no order of ranks can deadlock.
struct IpcMsg { char handle[HIP_IPC_HANDLE_SIZE]; unsigned long long offset, n_floats; };
int i_fail(const char *what) { return fail_msg(-6, "ipc communicator: %s (%s)", what, strerror(errno)); }
int i_write_all(int fd, const void *p, size_t n) { const char *c = (const char *)p; while (n) { const ssize_t r = send(fd, c, n, MSG_NOSIGNAL); if (r < 0) { if (errno == EINTR) continue; return -1; } c += r; n -= (size_t)r; } return 0; }
int i_read_all(int fd, void *p, size_t n)
{
    char *c = (char *)p;
    while (n) {
        struct pollfd pf = {fd, POLLIN, 0};
        const int pr = poll(&pf, 1, 60000);                            // a peer that died or never makes the matching call: fail instead of hanging the box
        if (pr == 0) { errno = ETIMEDOUT; return -1; }
        if (pr < 0) { if (errno == EINTR) continue; return -1; }
        const ssize_t r = recv(fd, c, n, 0);
        if (r == 0) { errno = ECONNRESET; return -1; }
        if (r < 0) { if (errno == EINTR) continue; return -1; }
        c += r; n -= (size_t)r;
    }
    return 0;
}
int i_send_one(csdr_amd_comm *m, const void *buf, size_t n, int peer, hipStream_t st)
{
    CSDR_HIP(hipStreamSynchronize(st));                                // the data is there before the peer is told
    void *base = nullptr; size_t size = 0;
    CSDR_HIP(hipMemGetAddressRange((hipDeviceptr_t *)&base, &size, (hipDeviceptr_t)buf));
    IpcMsg msg; memset(&msg, 0, sizeof msg);
    hipIpcMemHandle_t h; CSDR_HIP(hipIpcGetMemHandle(&h, base));
    memcpy(msg.handle, &h, sizeof h); msg.offset = (unsigned long long)((const char *)buf - (const char *)base); msg.n_floats = n;
    return i_write_all(m->peer_fd[peer], &msg, sizeof msg) ? i_fail("send to a peer") : 0;
}
int i_recv_one(csdr_amd_comm *m, void *buf, size_t n, int peer, hipStream_t st)
{
    IpcMsg msg;
    if (i_read_all(m->peer_fd[peer], &msg, sizeof msg)) return i_fail("no message from a peer within 60 s");
    if (msg.n_floats != n) return fail_msg(-3, "ipc communicator: rank %d sends %llu floats, rank %d expects %zu", peer, msg.n_floats, m->ddc.rank, n);
    const std::string key(msg.handle, sizeof msg.handle);
    void *base = nullptr;
    auto it = m->ipc_open.find(key);
    if (it != m->ipc_open.end()) base = it->second;
    else {
        hipIpcMemHandle_t h; memcpy(&h, msg.handle, sizeof h);
        CSDR_HIP(hipIpcOpenMemHandle(&base, h, hipIpcMemLazyEnablePeerAccess));
        m->ipc_open[key] = base;
    }
    CSDR_HIP(hipMemcpyAsync(buf, (const char *)base + msg.offset, n * sizeof(float), hipMemcpyDeviceToDevice, st));
    CSDR_HIP(hipStreamSynchronize(st));
    const char ack = 1;
    return i_write_all(m->peer_fd[peer], &ack, 1) ? i_fail("acknowledge to a peer") : 0;
}
int i_wait_ack(csdr_amd_comm *m, int peer) { char a = 0; return (i_read_all(m->peer_fd[peer], &a, 1) || a != 1) ? i_fail("no acknowledgement from a peer") : 0; }
int i_group_start(const DdcComm *c) { csdr_amd_comm *m = lc(c); m->ops.clear(); m->in_group = true; return 0; }
int i_send(const DdcComm *c, const void *buf, size_t n, int peer, hipStream_t st)
{
    csdr_amd_comm *m = lc(c);
    if (!m->in_group || peer < 0 || peer >= c->world || peer == c->rank) return fail_msg(-3, "ipc send: outside a group or bad peer %d", peer);
    m->ops.push_back({true, const_cast<void *>(buf), n, peer, st}); return 0;
}
int i_recv(const DdcComm *c, void *buf, size_t n, int peer, hipStream_t st)
{
    csdr_amd_comm *m = lc(c);
    if (!m->in_group || peer < 0 || peer >= c->world || peer == c->rank) return fail_msg(-3, "ipc recv: outside a group or bad peer %d", peer);
    m->ops.push_back({false, buf, n, peer, st}); return 0;
}
int i_group_end(const DdcComm *c)
{
    csdr_amd_comm *m = lc(c);
    m->in_group = false;
    int rc = 0;
    // Acknowledgements and messages share a socket per peer: a rank both sends to and receives from the same peer (a ring of two), so the order on each socket is
    // fixed: my message, then -- read side -- the peer's message, my acknowledgement, then the peer's acknowledgement.
    for (const auto &op : m->ops) if (op.send && !rc) rc = i_send_one(m, op.buf, op.n, op.peer, op.st);
    for (const auto &op : m->ops) if (!op.send && !rc) rc = i_recv_one(m, op.buf, op.n, op.peer, op.st);
    for (const auto &op : m->ops) if (op.send && !rc) rc = i_wait_ack(m, op.peer);
    m->ops.clear();
    return rc;
}
// every rank's piece [rank n, (rank + 1) n) of `all` to every other rank (root < 0), or the root's [0, n) to everybody
int i_collect(csdr_amd_comm *m, void *buf, size_t n_bytes, int root, hipStream_t st)
{
    const int me = m->ddc.rank, W = m->ddc.world;
    if (n_bytes % 4) return fail_msg(-3, "ipc communicator: collective of %zu bytes (whole floats only)", n_bytes);
    const size_t n = n_bytes / 4;
    int rc = 0;
    for (int p = 0; p < W && !rc; p++) if (p != me && (root < 0 || me == root)) rc = i_send_one(m, (const char *)buf + (root < 0 ? (size_t)me * n_bytes : 0), n, p, st);
    for (int p = 0; p < W && !rc; p++) if (p != me && (root < 0 || p == root)) rc = i_recv_one(m, (char *)buf + (root < 0 ? (size_t)p * n_bytes : 0), n, p, st);
    for (int p = 0; p < W && !rc; p++) if (p != me && (root < 0 || me == root)) rc = i_wait_ack(m, p);
    return rc;
}
int i_all_gather(const DdcComm *c, void *all, size_t n, hipStream_t st) { return c->world == 1 ? 0 : i_collect(lc(c), all, n * sizeof(float), -1, st); }
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
    g->child_in_use.assign(world, std::vector<char>());
    bool ok = true;
    for (auto *v : {&g->ready, &g->done, &g->ag_ready, &g->ag_done}) for (auto &e : *v) ok = ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
    if (!ok) { fail_msg(-1, "loopback: hipEventCreate failed"); csdr_amd_loopback_destroy(g); return nullptr; }
    return g;
}

void csdr_amd_loopback_destroy(csdr_amd_loopback *g)
{
    if (!g) return;
    for (auto *v : {&g->ready, &g->done, &g->ag_ready, &g->ag_done}) for (auto &e : *v) if (e) (void)hipEventDestroy(e);
    for (csdr_amd_loopback *ch : g->children) csdr_amd_loopback_destroy(ch);
    delete g;
}

/* a rank thread died: fail the others' rendezvous now instead of after the 60 s limit */
void csdr_amd_loopback_abort(csdr_amd_loopback *g)
{
    if (!g) return;
    std::lock_guard<std::mutex> lk(g->mu);
    g->broken = true; g->cv.notify_all();
    for (csdr_amd_loopback *ch : g->children) csdr_amd_loopback_abort(ch);
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

/* Ranks as processes on one box, any devices (in the tests: all on device 0).  path_prefix: a filesystem path all ranks see; rank r listens on "<prefix>.<r>" and
 * connects to every lower rank (60 s for the peers to appear).  Collective. */
csdr_amd_comm *csdr_amd_comm_create_ipc(csdr_amd_ctx *ctx, const char *path_prefix, int rank, int world)
{
    if (!path_prefix || world < 1 || world > 64 || rank < 0 || rank >= world) { fail_msg(-3, "ipc comm: bad rank %d of %d", rank, world); return nullptr; }
    csdr_amd_comm *c = new csdr_amd_comm();
    c->ctx = ctx; c->comm = nullptr; c->ipc = true; c->ipc_prefix = path_prefix;
    c->ddc.rank = rank; c->ddc.world = world; c->ddc.impl = c;
    c->ddc.group_start = i_group_start; c->ddc.group_end = i_group_end; c->ddc.send = i_send; c->ddc.recv = i_recv; c->ddc.all_gather = i_all_gather;
    c->peer_fd.assign(world, -1);
    auto addr_of = [&](int r, sockaddr_un &a) { memset(&a, 0, sizeof a); a.sun_family = AF_UNIX; snprintf(a.sun_path, sizeof a.sun_path, "%s.%d", path_prefix, r); };
    auto bail = [&](const char *what) { i_fail(what); csdr_amd_comm_destroy(c); return (csdr_amd_comm *)nullptr; };
    sockaddr_un me_a; addr_of(rank, me_a);
    if (strlen(path_prefix) + 8 >= sizeof me_a.sun_path) { fail_msg(-3, "ipc comm: path too long"); csdr_amd_comm_destroy(c); return nullptr; }
    c->listen_fd = socket(AF_UNIX, SOCK_STREAM, 0);
    (void)unlink(me_a.sun_path);
    if (c->listen_fd < 0 || bind(c->listen_fd, (sockaddr *)&me_a, sizeof me_a) || listen(c->listen_fd, world)) return bail("listen");
    for (int p = 0; p < rank; p++) {                                   // connect to the lower ranks ...
        sockaddr_un a; addr_of(p, a);
        int fd = -1;
        for (int tries = 0; tries < 6000; tries++) {
            fd = socket(AF_UNIX, SOCK_STREAM, 0);
            if (fd >= 0 && connect(fd, (sockaddr *)&a, sizeof a) == 0) break;
            if (fd >= 0) close(fd);
            fd = -1; usleep(10000);
        }
        if (fd < 0) return bail("a lower rank did not appear within 60 s");
        const int r32 = rank;
        if (i_write_all(fd, &r32, sizeof r32)) { close(fd); return bail("hello"); }
        c->peer_fd[p] = fd;
    }
    for (int k = rank + 1; k < world; k++) {                           // ... and accept the higher ones (in whatever order they come)
        struct pollfd pf = {c->listen_fd, POLLIN, 0};
        if (poll(&pf, 1, 60000) <= 0) { errno = ETIMEDOUT; return bail("a higher rank did not connect within 60 s"); }
        const int fd = accept(c->listen_fd, nullptr, nullptr);
        int r32 = -1;
        if (fd < 0 || i_read_all(fd, &r32, sizeof r32) || r32 <= rank || r32 >= world || c->peer_fd[r32] >= 0) { if (fd >= 0) close(fd); return bail("accept"); }
        c->peer_fd[r32] = fd;
    }
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
    if (c->ipc) {
        for (auto &kv : c->ipc_open) (void)hipIpcCloseMemHandle(kv.second);
        for (int fd : c->peer_fd) if (fd >= 0) close(fd);
        if (c->listen_fd >= 0) { close(c->listen_fd); char path[160]; snprintf(path, sizeof path, "%s.%d", c->ipc_prefix.c_str(), c->ddc.rank); (void)unlink(path); }
    }
    if (c->dup_parent && c->dup_index >= 0) {                         // the child group's place is free again for this rank
        std::lock_guard<std::mutex> lk(c->dup_parent->mu);
        std::vector<char> &use = c->dup_parent->child_in_use[c->ddc.rank];
        if (c->dup_index < (int)use.size()) use[c->dup_index] = 0;
    }
    delete c;
}

/* A second communicator over the same ranks (collective: every rank calls it, in the same order).  The time-sliced bank issues its input exchange and its output
 * exchange from two side streams; on ONE ncclComm the order in which the two groups reach the communicator could differ between ranks (ADVICE r3 / VERDICT r4 weak
 * #10 ii) -- each exchange now has a communicator of its own.  RCCL: rank 0 draws a new unique id and broadcasts it over the parent; loopback: a child group shared
 * by the rank threads; null transport: another null. */
csdr_amd_comm *csdr_amd_comm_dup(csdr_amd_comm *c)
{
    if (!c) { fail_msg(-3, "comm_dup: null communicator"); return nullptr; }
    const int rank = c->ddc.rank, world = c->ddc.world;
    if (c->null_transport) return csdr_amd_comm_create_null(c->ctx, rank, world);
    if (c->ipc) { const std::string pre = c->ipc_prefix + ".dup" + std::to_string(c->ipc_dups++); return csdr_amd_comm_create_ipc(c->ctx, pre.c_str(), rank, world); }      // (every rank counts its dups alike)
    if (c->loop) {
        csdr_amd_loopback *g = c->loop, *child = nullptr;
        int k = 0;
        {
            std::lock_guard<std::mutex> lk(g->mu);
            std::vector<char> &use = g->child_in_use[rank];
            while (k < (int)use.size() && use[k]) k++;
            while ((int)g->children.size() <= k) {
                csdr_amd_loopback *n = csdr_amd_loopback_create(world);
                if (!n) return nullptr;                                   // (nothing has been marked yet)
                g->children.push_back(n);
            }
            child = g->children[k];
        }
        csdr_amd_comm *d = csdr_amd_comm_create_loopback(c->ctx, child, rank);
        if (!d) return nullptr;
        {
            std::lock_guard<std::mutex> lk(g->mu);
            std::vector<char> &use = g->child_in_use[rank];
            if ((int)use.size() <= k) use.resize(k + 1, 0);
            use[k] = 1;
        }
        d->dup_parent = g; d->dup_index = k;
        return d;
    }
    if (hipSetDevice(c->ctx->device) != hipSuccess) { fail_msg(-1, "hipSetDevice(%d) failed", c->ctx->device); return nullptr; }
    // rank 0 ALWAYS takes part in the broadcast: when it cannot draw an id it sends zeros, which every rank (itself included) reads as the failure -- nobody is left
    // waiting in a collective the root never entered
    char id[128]; memset(id, 0, sizeof id);
    if (rank == 0 && csdr_amd_comm_unique_id(id)) memset(id, 0, sizeof id);
    if (world > 1) {
        void *d = nullptr;
        if (hipMalloc(&d, 128) != hipSuccess || hipMemcpy(d, id, 128, hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(d); fail_msg(-2, "comm_dup: staging buffer"); return nullptr; }
        const int rc = csdr_amd_comm_broadcast(c, d, 128, 0);
        const bool ok = rc == 0 && hipStreamSynchronize(c->ctx->stream) == hipSuccess && hipMemcpy(id, d, 128, hipMemcpyDeviceToHost) == hipSuccess;
        (void)hipFree(d);
        if (!ok) { if (!rc) fail_msg(-6, "comm_dup: the new unique id did not arrive"); return nullptr; }
    }
    { bool zero = true; for (int i = 0; i < 128; i++) zero = zero && id[i] == 0; if (zero) { if (rank != 0) fail_msg(-6, "comm_dup: rank 0 could not draw a unique id"); return nullptr; } }
    return csdr_amd_comm_create(c->ctx, id, rank, world);
}

/* First contact with a new transport / a new box: every exchange primitive the bank uses, once, on rank-stamped data with a checksum -- a send/recv ring
 * (rank r -> r + 1), an all-gather, a broadcast from the last rank, and two rings in flight at once on two communicators and two streams (the
 * time-sliced bank's pattern) -- timed, reported per rank.  report (may be null): one line, e.g.
 *   "rank 1/8 dev 1: ring 4.2 MB ok 0.31 ms 13.5 GB/s | all_gather 8 x 4.2 MB ok 0.9 ms | broadcast ok | second comm ring ok"
 * Returns 0 when every byte arrived as stamped, -6 with the first mismatch otherwise.  Collective: every rank calls it. */
int csdr_amd_comm_selftest(csdr_amd_comm *c, size_t n_floats, char *report, size_t report_cap)
{
    if (!c) return fail_msg(-3, "comm_selftest: null communicator");
    const DdcComm *d = &c->ddc;
    const int W = d->world, me = d->rank;
    if (n_floats < 16) n_floats = 16;
    n_floats &= ~(size_t)3;
    hipStream_t st = c->ctx->stream;
    CSDR_HIP(hipSetDevice(c->ctx->device));
    auto stamp = [](int rank, size_t i, int salt) { return (float)(int)(((unsigned)(rank + 1) * 2654435761u + (unsigned)i * 40503u + (unsigned)salt * 97u) >> 9); };      // < 2^23: exact in float
    std::vector<float> h(n_floats * (size_t)W), back(n_floats * (size_t)W);
    float *d_send = nullptr, *d_recv = nullptr, *d_all = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = 0; std::string line; char buf[256];
    snprintf(buf, sizeof buf, "rank %d/%d dev %d (%s):", me, W, c->ctx->device, c->loop ? "loopback" : c->null_transport ? "null" : c->ipc ? "ipc" : "rccl"); line = buf;
    auto cleanup = [&]() { (void)hipFree(d_send); (void)hipFree(d_recv); (void)hipFree(d_all); if (e0) (void)hipEventDestroy(e0); if (e1) (void)hipEventDestroy(e1); };
#define ST_HIP(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) { cleanup(); return ::csdr_amd::fail(e__, #expr, __FILE__, __LINE__); } } while (0)
    ST_HIP(hipMalloc((void **)&d_send, sizeof(float) * n_floats)); ST_HIP(hipMalloc((void **)&d_recv, sizeof(float) * n_floats)); ST_HIP(hipMalloc((void **)&d_all, sizeof(float) * n_floats * W));
    ST_HIP(hipEventCreate(&e0)); ST_HIP(hipEventCreate(&e1));
    const double mb = n_floats * 4 / 1e6;
    // the ring on `cm` / stream `s`: send to me + 1, receive from me - 1, check
    auto ring = [&](const DdcComm *cm, hipStream_t s, int salt, const char *name) -> int {
        for (size_t i = 0; i < n_floats; i++) h[i] = stamp(me, i, salt);
        ST_HIP(hipMemcpyAsync(d_send, h.data(), sizeof(float) * n_floats, hipMemcpyHostToDevice, s));
        ST_HIP(hipMemsetAsync(d_recv, 0xff, sizeof(float) * n_floats, s));
        float ms = 0.f; size_t badi = 0; bool bad = false;
        if (W > 1) {
            ST_HIP(hipEventRecord(e0, s));
            int r = cm->group_start(cm); if (!r) r = cm->send(cm, d_send, n_floats, (me + 1) % W, s); if (!r) r = cm->recv(cm, d_recv, n_floats, (me + W - 1) % W, s); if (!r) r = cm->group_end(cm);
            if (r) { cleanup(); return r; }
            ST_HIP(hipEventRecord(e1, s));
            ST_HIP(hipMemcpyAsync(back.data(), d_recv, sizeof(float) * n_floats, hipMemcpyDeviceToHost, s));
            ST_HIP(hipStreamSynchronize(s));
            ST_HIP(hipEventElapsedTime(&ms, e0, e1));
            if (!c->null_transport) for (size_t i = 0; i < n_floats && !bad; i++) if (back[i] != stamp((me + W - 1) % W, i, salt)) { bad = true; badi = i; }
        } else ST_HIP(hipStreamSynchronize(s));
        if (bad) { snprintf(buf, sizeof buf, " %s MISMATCH at float %zu (got %g, rank %d stamps %g)", name, badi, (double)back[badi], (me + W - 1) % W, (double)stamp((me + W - 1) % W, badi, salt)); line += buf; return 1; }
        snprintf(buf, sizeof buf, " %s %.1f MB ok %.3f ms%s", name, mb, (double)ms, W > 1 ? "" : " (one rank: nothing moves)"); line += buf;
        if (W > 1 && ms > 0) { snprintf(buf, sizeof buf, " %.1f GB/s |", mb / ms); line += buf; } else line += " |";
        return 0;
    };
    int bad_total = 0;
    { const int r = ring(d, st, 1, "ring"); if (r < 0) return r; bad_total += r; }
    // all-gather in place
    {
        for (size_t i = 0; i < n_floats; i++) h[i] = stamp(me, i, 2);
        ST_HIP(hipMemsetAsync(d_all, 0xff, sizeof(float) * n_floats * W, st));
        ST_HIP(hipMemcpyAsync(d_all + (size_t)me * n_floats, h.data(), sizeof(float) * n_floats, hipMemcpyHostToDevice, st));
        ST_HIP(hipEventRecord(e0, st));
        rc = d->all_gather(d, d_all, n_floats, st); if (rc) { cleanup(); return rc; }
        ST_HIP(hipEventRecord(e1, st));
        ST_HIP(hipMemcpyAsync(back.data(), d_all, sizeof(float) * n_floats * W, hipMemcpyDeviceToHost, st));
        ST_HIP(hipStreamSynchronize(st));
        float ms = 0.f; ST_HIP(hipEventElapsedTime(&ms, e0, e1));
        bool bad = false; int br = 0; size_t bi = 0;
        if (!c->null_transport) for (int r = 0; r < W && !bad; r++) for (size_t i = 0; i < n_floats; i++) if (back[(size_t)r * n_floats + i] != stamp(r, i, 2)) { bad = true; br = r; bi = i; break; }
        if (bad) { snprintf(buf, sizeof buf, " all_gather MISMATCH in rank %d's piece at float %zu |", br, bi); bad_total++; }
        else snprintf(buf, sizeof buf, " all_gather %d x %.1f MB ok %.3f ms |", W, mb, (double)ms);
        line += buf;
    }
    // broadcast from the last rank
    {
        const int root = W - 1;
        for (size_t i = 0; i < n_floats; i++) h[i] = stamp(me, i, 3);
        ST_HIP(hipMemcpyAsync(d_send, h.data(), sizeof(float) * n_floats, hipMemcpyHostToDevice, st));
        rc = csdr_amd_comm_broadcast(c, d_send, sizeof(float) * n_floats, root); if (rc) { cleanup(); return rc; }
        ST_HIP(hipMemcpyAsync(back.data(), d_send, sizeof(float) * n_floats, hipMemcpyDeviceToHost, st));
        ST_HIP(hipStreamSynchronize(st));
        bool bad = false;
        if (!c->null_transport) for (size_t i = 0; i < n_floats && !bad; i++) bad = back[i] != stamp(root, i, 3);
        line += bad ? " broadcast MISMATCH |" : " broadcast ok |"; bad_total += bad ? 1 : 0;
    }
    // a second communicator, its ring on a side stream while the first communicator's stream carries a ring of its own (the time-sliced bank's two exchanges)
    {
        csdr_amd_comm *c2 = csdr_amd_comm_dup(c);
        if (!c2) { cleanup(); return -6; }
        hipStream_t side = nullptr;
        float *d_s2 = nullptr, *d_r2 = nullptr;
        bool ok = hipStreamCreateWithFlags(&side, hipStreamNonBlocking) == hipSuccess && hipMalloc((void **)&d_s2, sizeof(float) * n_floats) == hipSuccess && hipMalloc((void **)&d_r2, sizeof(float) * n_floats) == hipSuccess;
        std::vector<float> h2(n_floats), back2(n_floats);
        int r = 0;
        if (ok) {
            for (size_t i = 0; i < n_floats; i++) { h[i] = stamp(me, i, 4); h2[i] = stamp(me, i, 5); }
            ok = hipMemcpy(d_send, h.data(), sizeof(float) * n_floats, hipMemcpyHostToDevice) == hipSuccess && hipMemcpy(d_s2, h2.data(), sizeof(float) * n_floats, hipMemcpyHostToDevice) == hipSuccess &&
                 hipMemset(d_recv, 0xff, sizeof(float) * n_floats) == hipSuccess && hipMemset(d_r2, 0xff, sizeof(float) * n_floats) == hipSuccess;
        }
        if (ok && W > 1) {
            const DdcComm *d2 = &c2->ddc;
            // the bank's pattern: every rank issues the input exchange's group (communicator 1, its stream) and then the output exchange's (communicator 2, the other
            // stream) without waiting in between -- the same host order on every rank, two communicators, so neither RCCL's per-communicator ordering nor the
            // loopback's host rendezvous can see the two groups cross
            for (int pass = 0; pass < 2 && !r; pass++) {
                const bool first_comm = pass == 0;
                const DdcComm *cm = first_comm ? d : d2; hipStream_t s = first_comm ? st : side;
                float *sb = first_comm ? d_send : d_s2, *rb = first_comm ? d_recv : d_r2;
                r = cm->group_start(cm); if (!r) r = cm->send(cm, sb, n_floats, (me + 1) % W, s); if (!r) r = cm->recv(cm, rb, n_floats, (me + W - 1) % W, s); if (!r) r = cm->group_end(cm);
            }
            if (!r) ok = hipStreamSynchronize(st) == hipSuccess && hipStreamSynchronize(side) == hipSuccess &&
                         hipMemcpy(back.data(), d_recv, sizeof(float) * n_floats, hipMemcpyDeviceToHost) == hipSuccess && hipMemcpy(back2.data(), d_r2, sizeof(float) * n_floats, hipMemcpyDeviceToHost) == hipSuccess;
            bool bad = false;
            if (!r && ok && !c->null_transport) for (size_t i = 0; i < n_floats && !bad; i++) bad = back[i] != stamp((me + W - 1) % W, i, 4) || back2[i] != stamp((me + W - 1) % W, i, 5);
            line += bad ? " two communicators, two streams: MISMATCH" : " two communicators on two streams, both groups in flight: ok"; bad_total += bad ? 1 : 0;
        } else if (ok) line += " second communicator: ok (one rank)";
        (void)hipFree(d_s2); (void)hipFree(d_r2); if (side) { (void)hipStreamSynchronize(side); (void)hipStreamDestroy(side); }
        csdr_amd_comm_destroy(c2);
        if (r) { cleanup(); return r; }
        if (!ok) { cleanup(); return fail_msg(-1, "comm_selftest: HIP call failed in the two-communicator pass"); }
    }
#undef ST_HIP
    cleanup();
    if (report && report_cap) { snprintf(report, report_cap, "%s", line.c_str()); }
    if (getenv("CSDR_AMD_COMM_VERBOSE")) fprintf(stderr, "csdr_amd comm selftest: %s\n", line.c_str());
    return bad_total ? fail_msg(-6, "comm_selftest: %s", line.c_str()) : 0;
}

int csdr_amd_comm_rank(const csdr_amd_comm *c) { return c->ddc.rank; }
int csdr_amd_comm_world(const csdr_amd_comm *c) { return c->ddc.world; }

/* broadcast of a device buffer from `root` on the context's stream (bench / test plumbing: e.g. checking a sharded run against rank 0's data) */
int csdr_amd_comm_broadcast(csdr_amd_comm *c, void *dev_buf, size_t bytes, int root)
{
    if (c->ddc.world == 1 || c->null_transport) return 0;
    if (c->loop) return l_collect(c, dev_buf, bytes, root, c->ctx->stream);
    if (c->ipc) return i_collect(c, dev_buf, (bytes + 3) & ~(size_t)3, root, c->ctx->stream);      // (whole floats: the callers' buffers are padded)
    CSDR_NCCL(g_rccl.Broadcast(dev_buf, dev_buf, bytes, ncclUint8, root, c->comm, c->ctx->stream));
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
