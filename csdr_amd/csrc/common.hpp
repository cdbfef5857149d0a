// common.hpp -- shared host-side plumbing for libcsdr_amd.so (MI355X / gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <string>
#include <vector>
#include "../../include/csdr_amd.h"

namespace csdr_amd {

typedef csdr_complexf cf32;

int fail(hipError_t e, const char *what, const char *file, int line);
int fail_msg(int code, const char *fmt, ...);

#define CSDR_HIP(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) return ::csdr_amd::fail(e__, #expr, __FILE__, __LINE__); } while (0)
#define CSDR_LAUNCH_CHECK() CSDR_HIP(hipGetLastError())

// float constant PI exactly as the reference defines it (libcsdr.h:65): a float, not a double
static const float PI_F = (float)3.14159265358979323846;

static inline unsigned cdiv(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }

enum { SCRATCH_SLOTS = 8 };

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (current device, kernel): the attribute is per device, contexts may live on several
// devices and be created from several threads (thread safe).  Returns 0 or a negative error.
int lds_attr_once(const void *kernel, size_t lds_bytes);
// multiProcessorCount of the CURRENT device (cached per device)
int current_device_cu_count();
void drop_fft_plans(hipStream_t st);      // fftpath.hip

// fastagc_ff (audio.hip) with an optional convert_f_s16 output written in the same pass; `out` may be null
int fastagc_ff_s16(struct ::csdr_amd_ctx *c, const float *in, float *out, int16_t *out_s16, int n_streams, int n_blocks, int block,
                   size_t in_pitch, size_t out_pitch, size_t s16_pitch, float reference, float *state_io, bool have_peaks = false);
float *fastagc_peaks_buffer(struct ::csdr_amd_ctx *c, int n_streams, int n_blocks);      // [n_streams][n_blocks + 2]; entries 2.. = peak |x| of the call's new blocks

} // namespace csdr_amd

struct csdr_amd_ctx {
    int device;
    hipStream_t stream;
    bool own_stream;
    std::string arch;
    void *scratch[csdr_amd::SCRATCH_SLOTS];
    size_t scratch_bytes[csdr_amd::SCRATCH_SLOTS];
    hipEvent_t ev0, ev1;
    // returns a device buffer of at least `bytes` that stays valid until the next request on the same slot
    void *get_scratch(int slot, size_t bytes);
    // pinned host staging for small host-computed tables (phase sequences, plans): acquire() waits until the
    // previous upload from the buffer has completed, upload() queues the async copy on the context's stream
    void *pinned; size_t pinned_bytes; hipEvent_t pinned_ev; bool pinned_in_flight;
    void *pinned_acquire(size_t bytes);
    int pinned_upload(void *dst_dev, size_t bytes);
    // shift_math_cc / shift_table_cc: the per-sample phase scan of the NEXT call, running on a helper thread while this call's kernels run (shift.hip: ShiftAhead)
    void *shift_ahead; void (*shift_ahead_free)(void *);
};

// ---- LDS-DMA row-step shared by the ring kernels (wfm_mfma.hip, ddc_mfma.hip).  Device code only.
#ifdef __HIPCC__
namespace csdr_amd {
// One row-step of a fetching wave as one piece of code: R rows, 1 KiB each (lane l: bytes 16 l .. 16 l + 15 of the run at sbase + vo[r]), LDS destinations RP bytes
// apart from la0 on.  M0 (the LDS destination) is saved once and stepped by s_add_u32, no branch between the pieces, `nt`: every byte is read once.  (Hand-written:
// the builtin form makes the compiler serialise the DMA with the LDS reads of other ring positions; vmcnt is counted by the callers.)
template <int R, int RP>
__device__ __forceinline__ void dma_rows(const uint32_t (&vo)[R], const uint8_t *sbase, uint32_t la0)
{
    uint32_t keep;
    static_assert(R == 2 || R == 4 || R == 8, "rows per fetching wave");
#define DMA_NEXT(k) "s_add_u32 m0, m0, %[rp]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v" #k "], %[sb] nt\n\t"
    if constexpr (R == 2)
        asm volatile("s_mov_b32 %[keep], m0\n\ts_mov_b32 m0, %[la]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v0], %[sb] nt\n\t" DMA_NEXT(1) "s_mov_b32 m0, %[keep]"
                     : [keep] "=&s"(keep) : [v0] "v"(vo[0]), [v1] "v"(vo[1]), [sb] "s"(sbase), [la] "s"(la0), [rp] "n"(RP) : "memory", "scc");
    else if constexpr (R == 4)
        asm volatile("s_mov_b32 %[keep], m0\n\ts_mov_b32 m0, %[la]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v0], %[sb] nt\n\t" DMA_NEXT(1) DMA_NEXT(2) DMA_NEXT(3) "s_mov_b32 m0, %[keep]"
                     : [keep] "=&s"(keep) : [v0] "v"(vo[0]), [v1] "v"(vo[1]), [v2] "v"(vo[2]), [v3] "v"(vo[3]), [sb] "s"(sbase), [la] "s"(la0), [rp] "n"(RP) : "memory", "scc");
    else
        asm volatile("s_mov_b32 %[keep], m0\n\ts_mov_b32 m0, %[la]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v0], %[sb] nt\n\t"
                     DMA_NEXT(1) DMA_NEXT(2) DMA_NEXT(3) DMA_NEXT(4) DMA_NEXT(5) DMA_NEXT(6) DMA_NEXT(7) "s_mov_b32 m0, %[keep]"
                     : [keep] "=&s"(keep) : [v0] "v"(vo[0]), [v1] "v"(vo[1]), [v2] "v"(vo[2]), [v3] "v"(vo[3]), [v4] "v"(vo[4]), [v5] "v"(vo[5]), [v6] "v"(vo[6]), [v7] "v"(vo[7]),
                       [sb] "s"(sbase), [la] "s"(la0), [rp] "n"(RP) : "memory", "scc");
#undef DMA_NEXT
}

} // namespace csdr_amd
#endif
