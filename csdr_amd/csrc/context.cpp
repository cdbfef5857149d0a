// context.cpp -- context, memory, error reporting for libcsdr_amd.so
#include "common.hpp"
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <map>
#include <mutex>

namespace csdr_amd {
static thread_local char g_err[512] = "";

int fail(hipError_t e, const char *what, const char *file, int line)
{
    snprintf(g_err, sizeof(g_err), "HIP error %d (%s) at %s:%d: %s", (int)e, hipGetErrorString(e), file, line, what);
    return -(int)e - 1000;
}
int fail_msg(int code, const char *fmt, ...)
{
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
    return code;
}
static std::mutex g_attr_mu;
static std::map<std::pair<int, const void *>, size_t> g_attr_done;
static int g_cu_count[64];

int lds_attr_once(const void *kernel, size_t lds_bytes)
{
    int dev = 0; (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(g_attr_mu);
    auto key = std::make_pair(dev, kernel);
    auto it = g_attr_done.find(key);
    if (it != g_attr_done.end() && it->second >= lds_bytes) return 0;
    CSDR_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    g_attr_done[key] = lds_bytes;
    return 0;
}

int current_device_cu_count()
{
    int dev = 0; (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(g_attr_mu);
    if (dev >= 0 && dev < 64 && g_cu_count[dev]) return g_cu_count[dev];
    hipDeviceProp_t pr; int n = (hipGetDeviceProperties(&pr, dev) == hipSuccess) ? pr.multiProcessorCount : 256;
    if (n < 1) n = 256;
    if (dev >= 0 && dev < 64) g_cu_count[dev] = n;
    return n;
}
} // namespace csdr_amd
using namespace csdr_amd;

void *csdr_amd_ctx::get_scratch(int slot, size_t bytes)
{
    if (bytes <= scratch_bytes[slot] && scratch[slot]) return scratch[slot];
    if (scratch[slot]) { (void)hipStreamSynchronize(stream); (void)hipFree(scratch[slot]); scratch[slot] = nullptr; scratch_bytes[slot] = 0; }
    size_t want = bytes + bytes / 4 + 4096;
    void *p = nullptr;
    if (hipMalloc(&p, want) != hipSuccess) { fail_msg(-2, "scratch allocation of %zu bytes failed", want); return nullptr; }
    scratch[slot] = p; scratch_bytes[slot] = want;
    return p;
}

void *csdr_amd_ctx::pinned_acquire(size_t bytes)
{
    if (pinned_in_flight) { (void)hipEventSynchronize(pinned_ev); pinned_in_flight = false; }
    if (bytes > pinned_bytes) {
        if (pinned) (void)hipHostFree(pinned);
        pinned = nullptr; pinned_bytes = 0;
        const size_t want = bytes + bytes / 2 + 4096;
        if (hipHostMalloc(&pinned, want, hipHostMallocDefault) != hipSuccess) { fail_msg(-2, "pinned allocation of %zu bytes failed", want); return nullptr; }
        pinned_bytes = want;
    }
    return pinned;
}

int csdr_amd_ctx::pinned_upload(void *dst_dev, size_t bytes)
{
    CSDR_HIP(hipMemcpyAsync(dst_dev, pinned, bytes, hipMemcpyHostToDevice, stream));
    CSDR_HIP(hipEventRecord(pinned_ev, stream));
    pinned_in_flight = true;
    return 0;
}

extern "C" {

const char *csdr_amd_last_error(void) { return g_err; }

int csdr_amd_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

csdr_amd_ctx *csdr_amd_ctx_create(int device, void *hip_stream)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) { fail_msg(-1, "no HIP device available (hipGetDeviceCount: %s) -- libcsdr_amd has no CPU fallback", hipGetErrorString(e)); return nullptr; }
    if (device < 0 || device >= n) { fail_msg(-1, "device %d out of range (%d devices)", device, n); return nullptr; }
    if (hipSetDevice(device) != hipSuccess) { fail_msg(-1, "hipSetDevice(%d) failed", device); return nullptr; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) { fail_msg(-1, "hipGetDeviceProperties failed"); return nullptr; }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) { fail_msg(-1, "device %d is %s; libcsdr_amd is built for gfx950 (MI355X) only", device, prop.gcnArchName); return nullptr; }
    csdr_amd_ctx *c = new csdr_amd_ctx();
    c->device = device; c->arch = prop.gcnArchName;
    for (int i = 0; i < SCRATCH_SLOTS; i++) { c->scratch[i] = nullptr; c->scratch_bytes[i] = 0; }
    if (hip_stream) { c->stream = (hipStream_t)hip_stream; c->own_stream = false; }
    else {
        if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { fail_msg(-1, "hipStreamCreate failed"); delete c; return nullptr; }
        c->own_stream = true;
    }
    if (hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess || hipEventCreateWithFlags(&c->pinned_ev, hipEventDisableTiming) != hipSuccess) { fail_msg(-1, "hipEventCreate failed"); delete c; return nullptr; }
    c->pinned = nullptr; c->pinned_bytes = 0; c->pinned_in_flight = false;
    c->shift_ahead = nullptr; c->shift_ahead_free = nullptr;
    return c;
}

void csdr_amd_ctx_destroy(csdr_amd_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    drop_fft_plans(c->stream);
    if (c->shift_ahead && c->shift_ahead_free) c->shift_ahead_free(c->shift_ahead);
    for (int i = 0; i < SCRATCH_SLOTS; i++) if (c->scratch[i]) (void)hipFree(c->scratch[i]);
    (void)hipEventDestroy(c->ev0); (void)hipEventDestroy(c->ev1); (void)hipEventDestroy(c->pinned_ev);
    if (c->pinned) (void)hipHostFree(c->pinned);
    if (c->own_stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int csdr_amd_ctx_sync(csdr_amd_ctx *c) { CSDR_HIP(hipStreamSynchronize(c->stream)); return 0; }
void *csdr_amd_ctx_stream(csdr_amd_ctx *c) { return (void *)c->stream; }
const char *csdr_amd_device_arch(csdr_amd_ctx *c) { return c->arch.c_str(); }

void *csdr_amd_malloc(csdr_amd_ctx *c, size_t bytes)
{
    void *p = nullptr;
    (void)hipSetDevice(c->device);
    hipError_t e = hipMalloc(&p, bytes ? bytes : 16);
    if (e != hipSuccess) { fail(e, "hipMalloc", __FILE__, __LINE__); return nullptr; }
    return p;
}
void csdr_amd_free(csdr_amd_ctx *c, void *p) { if (p) { (void)hipSetDevice(c->device); (void)hipFree(p); } }

int csdr_amd_h2d(csdr_amd_ctx *c, void *dst, const void *src, size_t bytes)
{
    CSDR_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    CSDR_HIP(hipStreamSynchronize(c->stream));
    return 0;
}
int csdr_amd_d2h(csdr_amd_ctx *c, void *dst, const void *src, size_t bytes)
{
    CSDR_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    CSDR_HIP(hipStreamSynchronize(c->stream));
    return 0;
}
int csdr_amd_d2d(csdr_amd_ctx *c, void *dst, const void *src, size_t bytes)
{
    if (!bytes) return 0;
    CSDR_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, c->stream));
    return 0;
}

int csdr_amd_memset(csdr_amd_ctx *c, void *dst, int value, size_t bytes)
{
    CSDR_HIP(hipMemsetAsync(dst, value, bytes, c->stream));
    return 0;
}
int csdr_amd_timer_start(csdr_amd_ctx *c) { CSDR_HIP(hipEventRecord(c->ev0, c->stream)); return 0; }
int csdr_amd_timer_stop_ms(csdr_amd_ctx *c, float *ms)
{
    CSDR_HIP(hipEventRecord(c->ev1, c->stream));
    CSDR_HIP(hipEventSynchronize(c->ev1));
    CSDR_HIP(hipEventElapsedTime(ms, c->ev0, c->ev1));
    return 0;
}

} // extern "C"
