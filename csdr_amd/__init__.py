"""csdr_amd -- Python harness over libcsdr_amd.so (the MI355X-native libcsdr hot path).

The product is the C-ABI shared library (include/csdr_amd.h, include/libcsdr_amd_compat.h) built from
csdr_amd/csrc/*.hip for gfx950.  This module only loads it with ctypes and offers numpy-in/numpy-out
conveniences for the tests and bench.py; it contains no DSP and NO fallback: if the library or the GPU is
missing, everything raises.
"""
import ctypes as C
import os
import sys
import subprocess
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CSDR_AMD_LIB") or os.path.join(HERE, "libcsdr_amd.so")     # override: A/B builds on one GPU box
ROOT = os.path.dirname(HERE)

c64 = np.complex64
f32 = np.float32

SHIFT = {"addition": 0, "math": 1, "table": 2, "unroll": 3, "addfast": 4}
WINDOWS = {"BOXCAR": 0, "BLACKMAN": 1, "HAMMING": 2}


class CsdrAmdError(RuntimeError):
    pass


def build(verbose=False):
    """Compile every HIP source for gfx950 into csdr_amd/libcsdr_amd.so (hipcc cross-compiles without a GPU)."""
    r = subprocess.run(["make", "-C", os.path.join(HERE, "csrc"), "-j8"], capture_output=True, text=True)
    if r.returncode != 0:
        raise CsdrAmdError("libcsdr_amd build failed:\n" + r.stdout[-4000:] + r.stderr[-4000:])
    if verbose:
        print(r.stdout)
    return LIB_PATH


class FastDDC(C.Structure):       # csdr_fastddc_t == fastddc_t (fastddc.h:5-24)
    _fields_ = [(n, C.c_int) for n in ("pre_decimation", "post_decimation", "taps_length", "taps_min_length",
                                       "overlap_length", "fft_size", "fft_inv_size", "input_size", "post_input_size")] + \
               [("pre_shift", C.c_float), ("startbin", C.c_int), ("v", C.c_int), ("offsetbin", C.c_int),
                ("post_shift", C.c_float), ("output_scrape", C.c_int), ("scrap", C.c_int),
                ("sindelta", C.c_float), ("cosdelta", C.c_float), ("rate", C.c_float)]

    def as_dict(self):
        d = {n: getattr(self, n) for n, _ in self._fields_ if n not in ("sindelta", "cosdelta", "rate")}
        d["dsadata"] = (self.sindelta, self.cosdelta, self.rate)
        return d


_lib = None


def lib():
    """The loaded shared library (raises if it has not been built)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CsdrAmdError("%s is missing -- run `python -c 'import __graft_entry__ as g; g.build()'` first" % LIB_PATH)
    # One HIP runtime per process: libcsdr_amd.so links the system libamdhip64, torch ships its own copy.  Loaded torch-first, the library
    # resolves to the copy torch brought (both share devices and streams); loaded the other way round, torch's runtime finds the devices taken
    # ("no ROCm-capable device").  So if torch is going to be used in this process at all, it has to come first.
    if "torch" not in sys.modules and not os.environ.get("CSDR_AMD_NO_TORCH_PRELOAD"):
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    L = C.CDLL(LIB_PATH)
    vp, sz, i, fl = C.c_void_p, C.c_size_t, C.c_int, C.c_float
    L.csdr_amd_ctx_create.restype = vp; L.csdr_amd_ctx_create.argtypes = [i, vp]
    L.csdr_amd_ctx_destroy.argtypes = [vp]
    L.csdr_amd_ctx_sync.argtypes = [vp]
    L.csdr_amd_ctx_stream.restype = vp; L.csdr_amd_ctx_stream.argtypes = [vp]
    L.csdr_amd_last_error.restype = C.c_char_p
    L.csdr_amd_device_arch.restype = C.c_char_p; L.csdr_amd_device_arch.argtypes = [vp]
    L.csdr_amd_malloc.restype = vp; L.csdr_amd_malloc.argtypes = [vp, sz]
    L.csdr_amd_free.argtypes = [vp, vp]
    L.csdr_amd_h2d.argtypes = [vp, vp, vp, sz]
    L.csdr_amd_d2h.argtypes = [vp, vp, vp, sz]
    L.csdr_amd_memset.argtypes = [vp, vp, i, sz]
    sh = C.c_short
    L.csdr_amd_d2d.argtypes = [vp, vp, vp, sz]
    L.csdr_amd_amdemod_cf.argtypes = [vp, vp, vp, sz]
    L.csdr_amd_amdemod_estimator_cf.argtypes = [vp, vp, vp, sz, fl, fl]
    L.csdr_amd_realpart_cf.argtypes = [vp, vp, vp, sz]
    L.csdr_amd_logpower_cf.argtypes = [vp, vp, vp, sz, fl]
    L.csdr_amd_fmdemod_atan_cf.argtypes = [vp, vp, vp, i, sz, sz, sz, vp]
    L.csdr_amd_dcblock_ff.argtypes = [vp, vp, vp, i, sz, sz, sz, fl, vp]
    L.csdr_amd_fastdcblock_ff.argtypes = [vp, vp, vp, i, i, i, sz, sz, vp]
    L.csdr_amd_agc_ff.argtypes = [vp, vp, vp, i, sz, i, sz, sz, fl, fl, fl, fl, sh, sh, fl, vp]
    L.csdr_amd_precalculate_window.argtypes = [vp, i, i]; L.csdr_amd_precalculate_window.restype = None
    L.csdr_amd_fftcc_create.restype = vp; L.csdr_amd_fftcc_create.argtypes = [vp, i, i, i, i]
    L.csdr_amd_fftcc_destroy.argtypes = [vp]; L.csdr_amd_fftcc_destroy.restype = None
    L.csdr_amd_fftcc_process.argtypes = [vp, vp, sz, vp, C.POINTER(sz)]
    L.csdr_amd_encode_ima_adpcm_i16_u8.argtypes = [vp, vp, vp, i, sz, sz, sz, vp]
    L.csdr_amd_decode_ima_adpcm_u8_i16.argtypes = [vp, vp, vp, i, sz, sz, sz, vp]
    L.csdr_amd_compress_fft_adpcm_f_u8.argtypes = [vp, vp, vp, i, i]
    L.csdr_amd_timer_start.argtypes = [vp]
    L.csdr_amd_timer_stop_ms.argtypes = [vp, C.POINTER(fl)]
    L.csdr_amd_firdes_filter_len.argtypes = [fl]
    L.csdr_amd_firdes_lowpass_f.argtypes = [vp, i, fl, i]
    L.csdr_amd_firdes_bandpass_c.argtypes = [vp, i, fl, fl, i]
    L.csdr_amd_nfm_deemph_taps.argtypes = [i, C.POINTER(vp)]
    L.csdr_amd_shift_addition_init.argtypes = [fl, vp]
    for nm in ("u8_f", "s8_f", "s16_f", "f_u8", "f_s8", "f_s16"):
        getattr(L, "csdr_amd_convert_" + nm).argtypes = [vp, vp, vp, sz]
    L.csdr_amd_convert_f_s24.argtypes = [vp, vp, vp, sz, i]
    L.csdr_amd_convert_s24_f.argtypes = [vp, vp, vp, sz, i]
    L.csdr_amd_rotator_generate.argtypes = [vp, i, fl, C.POINTER(fl), vp, sz, i, i]
    L.csdr_amd_mix_cc.argtypes = [vp, vp, vp, vp, i, sz, sz, sz]
    L.csdr_amd_mix_fc.argtypes = [vp, vp, vp, vp, i, sz, sz, sz]
    L.csdr_amd_shift_cc.argtypes = [vp, i, fl, C.POINTER(fl), vp, vp, i, sz, sz, sz, i, i]
    L.csdr_amd_decimating_shift_addition_cc.argtypes = [vp, vp, vp, i, i, sz, sz, vp, i, vp]
    L.csdr_amd_fir_decimate_cc.argtypes = [vp, vp, vp, i, i, sz, sz, i, vp, i]
    L.csdr_amd_fir_last_kernel.restype = C.c_char_p; L.csdr_amd_fir_last_kernel.argtypes = []
    L.csdr_amd_fir_ff.argtypes = [vp, vp, vp, i, i, sz, sz, vp, i]
    L.csdr_amd_fmdemod_quadri_cf.argtypes = [vp, vp, vp, i, sz, sz, sz, vp]
    L.csdr_amd_limit_ff.argtypes = [vp, vp, vp, sz, fl]
    L.csdr_amd_gain_ff.argtypes = [vp, vp, vp, sz, fl]
    L.csdr_amd_deemphasis_wfm_ff.argtypes = [vp, vp, vp, i, sz, sz, sz, fl, i, vp]
    L.csdr_amd_fastagc_ff.argtypes = [vp, vp, vp, i, i, i, sz, sz, fl, vp]
    L.csdr_amd_fracdec_create.restype = vp; L.csdr_amd_fracdec_create.argtypes = [fl, i, vp, i]
    L.csdr_amd_fracdec_destroy.argtypes = [vp]
    L.csdr_amd_fractional_decimator_ff.argtypes = [vp, vp, vp, vp, i, i, sz, sz, C.POINTER(i)]
    L.csdr_amd_fracdec_set_cli_bufsize.restype = None; L.csdr_amd_fracdec_set_cli_bufsize.argtypes = [vp, i]
    L.csdr_amd_fftfilt_create.restype = vp; L.csdr_amd_fftfilt_create.argtypes = [vp, i, vp, i, i, i]
    L.csdr_amd_fftfilt_destroy.argtypes = [vp]
    L.csdr_amd_fftfilt_input_size.argtypes = [vp]
    L.csdr_amd_fftfilt_reset.argtypes = [vp]
    L.csdr_amd_fftfilt_kernel_name.restype = C.c_char_p; L.csdr_amd_fftfilt_kernel_name.argtypes = [vp]
    L.csdr_amd_fftfilt_window.argtypes = [vp]
    L.csdr_amd_fftfilt_process.argtypes = [vp, vp, vp, i, sz, sz]
    L.csdr_amd_fft_c2c.argtypes = [vp, vp, vp, i, i]
    L.csdr_amd_fastddc_init.argtypes = [vp, fl, i, fl]
    L.csdr_amd_fastddc_fwd_create.restype = vp; L.csdr_amd_fastddc_fwd_create.argtypes = [vp, vp, i]
    L.csdr_amd_fastddc_fwd_destroy.argtypes = [vp]
    L.csdr_amd_fastddc_fwd_process.argtypes = [vp, vp, vp, i]
    L.csdr_amd_fastddc_inv_create.restype = vp; L.csdr_amd_fastddc_inv_create.argtypes = [vp, fl, i, vp, i, i, i]
    L.csdr_amd_fastddc_inv_destroy.argtypes = [vp]
    L.csdr_amd_fastddc_inv_geometry.argtypes = [vp, i, vp]
    L.csdr_amd_fastddc_inv_max_output.argtypes = [vp, i]
    L.csdr_amd_fastddc_inv_process.argtypes = [vp, vp, i, vp, sz, vp]
    L.csdr_amd_fastddc_bank_create.restype = vp; L.csdr_amd_fastddc_bank_create.argtypes = [vp, fl, i, vp, i, i, i]
    L.csdr_amd_fastddc_bank_destroy.argtypes = [vp]; L.csdr_amd_fastddc_bank_destroy.restype = None
    L.csdr_amd_fastddc_bank_set_rate.argtypes = [vp, i, fl]
    L.csdr_amd_fastddc_bank_input_size.argtypes = [vp]
    L.csdr_amd_fastddc_bank_max_output.argtypes = [vp, i]
    L.csdr_amd_fastddc_bank_process.argtypes = [vp, vp, i, vp, sz, vp]
    L.csdr_amd_fastddc_bank_inverse.restype = vp; L.csdr_amd_fastddc_bank_inverse.argtypes = [vp]
    L.csdr_amd_fastddc_bank_submit.argtypes = [vp, vp, i]
    L.csdr_amd_fastddc_bank_collect.argtypes = [vp, vp, sz, vp]
    L.csdr_amd_comm_unique_id.argtypes = [vp]
    L.csdr_amd_comm_create.restype = vp; L.csdr_amd_comm_create.argtypes = [vp, vp, i, i]
    L.csdr_amd_comm_destroy.argtypes = [vp]; L.csdr_amd_comm_destroy.restype = None
    L.csdr_amd_comm_rank.argtypes = [vp]; L.csdr_amd_comm_world.argtypes = [vp]
    L.csdr_amd_comm_broadcast.argtypes = [vp, vp, sz, i]
    L.csdr_amd_comm_dup.restype = vp; L.csdr_amd_comm_dup.argtypes = [vp]
    L.csdr_amd_comm_selftest.argtypes = [vp, sz, C.c_char_p, sz]
    L.csdr_amd_fastddc_bank_create_sharded.restype = vp; L.csdr_amd_fastddc_bank_create_sharded.argtypes = [vp, fl, i, vp, i, i, i, vp]
    L.csdr_amd_fastddc_bank_channel_slice.argtypes = [vp, C.POINTER(i), C.POINTER(i)]
    L.csdr_amd_fastddc_bank_create_sharded_by.restype = vp; L.csdr_amd_fastddc_bank_create_sharded_by.argtypes = [vp, fl, i, vp, i, i, i, vp, i]
    L.csdr_amd_fastddc_bank_shard_mode.argtypes = [vp]
    L.csdr_amd_fastddc_bank_default_shard_mode.argtypes = [i]
    L.csdr_amd_fastddc_bank_local_blocks.argtypes = [vp, i, C.POINTER(i), C.POINTER(i)]
    L.csdr_amd_fastddc_bank_overlap.argtypes = [vp]
    L.csdr_amd_fastddc_bank_submit_local.argtypes = [vp, vp, i]
    if hasattr(L, "csdr_amd_fastddc_bank_process_s16"):
        for nm in ("s16", "u8"):
            getattr(L, "csdr_amd_fastddc_bank_process_" + nm).argtypes = [vp, vp, i, vp, sz, vp]
            getattr(L, "csdr_amd_fastddc_bank_submit_" + nm).argtypes = [vp, vp, i]
            getattr(L, "csdr_amd_fastddc_bank_submit_local_" + nm).argtypes = [vp, vp, i]
    L.csdr_amd_fastddc_bank_finish.argtypes = [vp, vp]
    L.csdr_amd_fastddc_bank_set_rate_global.argtypes = [vp, i, fl]
    L.csdr_amd_loopback_create.restype = vp; L.csdr_amd_loopback_create.argtypes = [i]
    L.csdr_amd_loopback_destroy.argtypes = [vp]; L.csdr_amd_loopback_destroy.restype = None
    L.csdr_amd_loopback_abort.argtypes = [vp]; L.csdr_amd_loopback_abort.restype = None
    L.csdr_amd_comm_create_loopback.restype = vp; L.csdr_amd_comm_create_loopback.argtypes = [vp, vp, i]
    L.csdr_amd_comm_create_null.restype = vp; L.csdr_amd_comm_create_null.argtypes = [vp, i, i]
    L.csdr_amd_comm_create_ipc.restype = vp; L.csdr_amd_comm_create_ipc.argtypes = [vp, C.c_char_p, i, i]
    L.csdr_amd_fastddc_inv_kernel_name.restype = C.c_char_p; L.csdr_amd_fastddc_inv_kernel_name.argtypes = [vp]
    L.csdr_amd_fastddc_inv_set_profiling.argtypes = [vp, i]
    L.csdr_amd_fastddc_inv_kernel_time.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_long)]
    L.csdr_amd_fastddc_inv_stage_time.argtypes = [vp, i, C.POINTER(C.c_double), C.POINTER(C.c_long)]
    L.csdr_amd_wfm_create.restype = vp; L.csdr_amd_wfm_create.argtypes = [vp, i, fl, i, vp, i, i, fl, i, sz]
    L.csdr_amd_wfm_destroy.argtypes = [vp]
    L.csdr_amd_wfm_reset.argtypes = [vp]
    L.csdr_amd_wfm_process.restype = C.c_long; L.csdr_amd_wfm_process.argtypes = [vp, vp, sz, sz, vp, vp, sz]
    L.csdr_amd_wfm_kernel_name.restype = C.c_char_p; L.csdr_amd_wfm_kernel_name.argtypes = [vp]
    ll = C.c_longlong; db = C.c_double
    L.csdr_amd_wfm_ring_create.restype = vp; L.csdr_amd_wfm_ring_create.argtypes = [vp, i, fl, i, vp, i, i, fl, i, sz, i]
    L.csdr_amd_wfm_ring_destroy.argtypes = [vp]; L.csdr_amd_wfm_ring_destroy.restype = None
    L.csdr_amd_wfm_ring_reset.argtypes = [vp]
    L.csdr_amd_wfm_ring_acquire.argtypes = [vp, ll, db]
    L.csdr_amd_wfm_ring_input.restype = vp; L.csdr_amd_wfm_ring_input.argtypes = [vp, ll, C.POINTER(sz)]
    L.csdr_amd_wfm_ring_submit.restype = ll; L.csdr_amd_wfm_ring_submit.argtypes = [vp]
    L.csdr_amd_wfm_ring_wait.restype = C.c_long; L.csdr_amd_wfm_ring_wait.argtypes = [vp, ll, db]
    L.csdr_amd_wfm_ring_output.restype = vp; L.csdr_amd_wfm_ring_output.argtypes = [vp, ll, C.POINTER(sz)]
    L.csdr_amd_wfm_ring_set_rate.argtypes = [vp, fl]
    L.csdr_amd_wfm_ring_get_rate.restype = fl; L.csdr_amd_wfm_ring_get_rate.argtypes = [vp]
    L.csdr_amd_wfm_ring_set_timeouts.argtypes = [vp, db, db]
    L.csdr_amd_wfm_ring_resident.argtypes = [vp]
    L.csdr_amd_wfm_ring_stop.argtypes = [vp]
    L.csdr_amd_wfm_ring_slots.argtypes = [vp]
    L.csdr_amd_wfm_ring_grid.argtypes = [vp]
    L.csdr_amd_wfm_ring_launches.restype = C.c_long; L.csdr_amd_wfm_ring_launches.argtypes = [vp]
    L.csdr_amd_wfm_ring_submitted.restype = ll; L.csdr_amd_wfm_ring_submitted.argtypes = [vp]
    L.csdr_amd_wfm_ring_block_times.argtypes = [vp, ll, C.POINTER(db), C.POINTER(db)]
    L.csdr_amd_wfm_ring_replay.argtypes = [vp, C.c_long, vp, vp]
    L.csdr_amd_wfm_ring_stats.argtypes = [vp, vp]
    L.csdr_amd_wfm_set_profiling.argtypes = [vp, i]
    L.csdr_amd_wfm_kernel_time.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_long)]
    L.csdr_amd_ddc_create.restype = vp; L.csdr_amd_ddc_create.argtypes = [vp, i, fl, i, vp, i, sz]
    if hasattr(L, "csdr_amd_ddc_create_rates"):          # (absent from older builds selected with CSDR_AMD_LIB for A/B runs)
        L.csdr_amd_ddc_create_rates.restype = vp; L.csdr_amd_ddc_create_rates.argtypes = [vp, i, vp, i, vp, i, sz]
        L.csdr_amd_ddc_set_rate.argtypes = [vp, i, fl]
        L.csdr_amd_ddc_get_rate.restype = fl; L.csdr_amd_ddc_get_rate.argtypes = [vp, i]
        L.csdr_amd_ddc_fallback.argtypes = [vp]
        L.csdr_amd_wfm_fallback.argtypes = [vp]
        L.csdr_amd_wfm_create_rates.restype = vp; L.csdr_amd_wfm_create_rates.argtypes = [vp, i, vp, i, vp, i, i, fl, i, sz]
        L.csdr_amd_wfm_set_rate.argtypes = [vp, i, fl]
        L.csdr_amd_wfm_get_rate.restype = fl; L.csdr_amd_wfm_get_rate.argtypes = [vp, i]
        L.csdr_amd_nfm_create_rates.restype = vp; L.csdr_amd_nfm_create_rates.argtypes = [vp, i, vp, i, vp, i, i, i, fl, fl, sz]
        L.csdr_amd_nfm_set_rate.argtypes = [vp, i, fl]
    L.csdr_amd_ddc_destroy.argtypes = [vp]
    L.csdr_amd_ddc_reset.argtypes = [vp]
    L.csdr_amd_ddc_process.restype = C.c_long; L.csdr_amd_ddc_process.argtypes = [vp, vp, sz, sz, vp, sz]
    L.csdr_amd_ddc_kernel_name.restype = C.c_char_p; L.csdr_amd_ddc_kernel_name.argtypes = [vp]
    L.csdr_amd_ddc_set_profiling.argtypes = [vp, i]
    L.csdr_amd_ddc_kernel_time.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_long)]
    L.csdr_amd_nfm_create.restype = vp; L.csdr_amd_nfm_create.argtypes = [vp, i, fl, i, vp, i, i, i, fl, fl, sz]
    L.csdr_amd_nfm_destroy.argtypes = [vp]
    L.csdr_amd_nfm_reset.argtypes = [vp]
    L.csdr_amd_nfm_process.restype = C.c_long; L.csdr_amd_nfm_process.argtypes = [vp, vp, sz, sz, vp, vp, sz]
    L.csdr_amd_nfm_front_end.restype = vp; L.csdr_amd_nfm_front_end.argtypes = [vp]
    _lib = L
    return L


def _hp(a):
    return a.ctypes.data_as(C.c_void_p)


class DevBuf:
    """A device allocation owned by a Context."""

    def __init__(self, ctx, nbytes):
        self.ctx, self.nbytes = ctx, int(nbytes)
        self.ptr = lib().csdr_amd_malloc(ctx.h, max(self.nbytes, 16))
        if not self.ptr:
            raise CsdrAmdError(ctx.err())

    def free(self):
        if self.ptr and self.ctx.h:                  # a buffer that outlives its (closed) context is gone with the process
            lib().csdr_amd_free(self.ctx.h, self.ptr)
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    def at(self, byte_offset):
        return C.c_void_p(self.ptr + int(byte_offset))


class Context:
    """csdr_amd_ctx: one GPU, one HIP stream."""

    def __init__(self, device=0, hip_stream=None):
        self.L = lib()
        self.h = self.L.csdr_amd_ctx_create(device, hip_stream)
        if not self.h:
            raise CsdrAmdError("csdr_amd_ctx_create failed: " + self.L.csdr_amd_last_error().decode())

    def err(self):
        return self.L.csdr_amd_last_error().decode()

    def check(self, rc, what=""):
        if rc < 0:
            raise CsdrAmdError("%s failed (%d): %s" % (what, rc, self.err()))
        return rc

    def close(self):
        if self.h:
            self.L.csdr_amd_ctx_destroy(self.h)
            self.h = None

    def sync(self):
        self.check(self.L.csdr_amd_ctx_sync(self.h), "sync")

    def arch(self):
        return self.L.csdr_amd_device_arch(self.h).decode()

    def alloc(self, nbytes):
        return DevBuf(self, nbytes)

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        b = DevBuf(self, arr.nbytes)
        if arr.nbytes:
            self.check(self.L.csdr_amd_h2d(self.h, b.ptr, _hp(arr), arr.nbytes), "h2d")
        return b

    def download(self, buf, dtype, count, byte_offset=0):
        out = np.empty(count, dtype=dtype)
        if out.nbytes:
            self.check(self.L.csdr_amd_d2h(self.h, _hp(out), buf.at(byte_offset), out.nbytes), "d2h")
        return out

    def timer_start(self):
        self.check(self.L.csdr_amd_timer_start(self.h), "timer_start")

    def timer_stop_ms(self):
        ms = C.c_float(0)
        self.check(self.L.csdr_amd_timer_stop_ms(self.h, C.byref(ms)), "timer_stop")
        return ms.value

    # ------------------------------------------------------------------ host-side design
    def firdes_filter_len(self, tbw):
        return self.L.csdr_amd_firdes_filter_len(tbw)

    def firdes_lowpass_f(self, length, cutoff, window="HAMMING"):
        t = np.zeros(length, f32); self.L.csdr_amd_firdes_lowpass_f(_hp(t), length, cutoff, WINDOWS[window]); return t

    def firdes_bandpass_c(self, length, lo, hi, window="HAMMING"):
        t = np.zeros(length, c64); self.L.csdr_amd_firdes_bandpass_c(_hp(t), length, lo, hi, WINDOWS[window]); return t

    def nfm_taps(self, sample_rate):
        p = C.c_void_p()
        n = self.L.csdr_amd_nfm_deemph_taps(sample_rate, C.byref(p))
        if not n:
            return np.zeros(0, f32)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_float)), (n,)).copy()

    # ------------------------------------------------------------------ numpy conveniences (single call, 2-D batches)
    def _conv(self, name, x, in_dt, out_dt, n_out=None, extra=()):
        x = np.ascontiguousarray(x, in_dt)
        n = x.size if in_dt != np.uint8 or name != "csdr_amd_convert_s24_f" else x.size // 3
        n_out = n if n_out is None else n_out
        di = self.upload(np.concatenate([x.ravel(), np.zeros(16, in_dt)]))
        do = self.alloc(np.dtype(out_dt).itemsize * (n_out + 16))
        self.check(getattr(self.L, name)(self.h, di.ptr, do.ptr, n, *extra), name)
        return self.download(do, out_dt, n_out)

    def convert_u8_f(self, x): return self._conv("csdr_amd_convert_u8_f", x, np.uint8, f32)
    def convert_s8_f(self, x): return self._conv("csdr_amd_convert_s8_f", x, np.int8, f32)
    def convert_s16_f(self, x): return self._conv("csdr_amd_convert_s16_f", x, np.int16, f32)
    def convert_f_u8(self, x): return self._conv("csdr_amd_convert_f_u8", x, f32, np.uint8)
    def convert_f_s8(self, x): return self._conv("csdr_amd_convert_f_s8", x, f32, np.int8)
    def convert_f_s16(self, x): return self._conv("csdr_amd_convert_f_s16", x, f32, np.int16)

    def convert_f_s24(self, x, bigendian=0):
        x = np.ascontiguousarray(x, f32)
        return self._conv("csdr_amd_convert_f_s24", x, f32, np.uint8, n_out=3 * x.size, extra=(int(bigendian),))

    def convert_s24_f(self, x, bigendian=0):
        x = np.ascontiguousarray(x, np.uint8)
        return self._conv("csdr_amd_convert_s24_f", x, np.uint8, f32, n_out=x.size // 3, extra=(int(bigendian),))

    @staticmethod
    def _2d(x, dt):
        x = np.ascontiguousarray(x, dt)
        return (x[None, :], True) if x.ndim == 1 else (x, False)

    def shift_cc(self, x, rate, variant="addition", phase=0.0, chunk=1024, aux=0):
        """x: [n] or [streams, n] complex64 -> (shifted, new_phase)"""
        x2, squeeze = self._2d(x, c64)
        s, n = x2.shape
        di = self.upload(x2); do = self.alloc(x2.nbytes + 64)
        ph = C.c_float(phase)
        self.check(self.L.csdr_amd_shift_cc(self.h, SHIFT[variant], rate, C.byref(ph), di.ptr, do.ptr, s, n, n, n, chunk, aux), "shift_cc")
        y = self.download(do, c64, s * n).reshape(s, n)
        return (y[0] if squeeze else y), ph.value

    # the reference's own names for the five shifter variants (libcsdr.h:108, 185-207; libcsdr_gpl.h:32-35), CLI chunking included
    def shift_addition_cc(self, x, rate, chunk=1024, phase=0.0): return self.shift_cc(x, rate, "addition", phase, chunk)
    def shift_math_cc(self, x, rate, phase=0.0): return self.shift_cc(x, rate, "math", phase)
    def shift_addfast_cc(self, x, rate, chunk=1024, phase=0.0): return self.shift_cc(x, rate, "addfast", phase, chunk)
    def shift_unroll_cc(self, x, rate, size=1024, phase=0.0): return self.shift_cc(x, rate, "unroll", phase, size, size)
    def shift_table_cc(self, x, rate, table_size=65536, phase=0.0): return self.shift_cc(x, rate, "table", phase, 1024, table_size)

    def shift_addition_fc(self, x, rate, phase=0.0, chunk=1024):
        x = np.ascontiguousarray(x, f32); n = x.size
        di = self.upload(x); do = self.alloc(8 * n + 64); rot = self.alloc(8 * n + 64)
        ph = C.c_float(phase)
        self.check(self.L.csdr_amd_rotator_generate(self.h, 0, rate, C.byref(ph), rot.ptr, n, chunk, 0), "rotator")
        self.check(self.L.csdr_amd_mix_fc(self.h, di.ptr, do.ptr, rot.ptr, 1, n, n, n), "mix_fc")
        return self.download(do, c64, n), ph.value

    def decimating_shift_addition_cc(self, x, rate, decimation, status=(0, 0.0, 0)):
        x = np.ascontiguousarray(x, c64); n = x.size
        dsa = np.zeros(3, f32); self.L.csdr_amd_shift_addition_init(np.float32(rate) * np.float32(decimation), _hp(dsa))
        st = np.zeros(3, np.int32); st[0] = status[0]; st[1] = np.array([status[1]], f32).view(np.int32)[0]; st[2] = status[2]
        di = self.upload(x); do = self.alloc(8 * (n // decimation + 4)); dd = self.upload(dsa); ds = self.upload(st)
        self.check(self.L.csdr_amd_decimating_shift_addition_cc(self.h, di.ptr, do.ptr, 1, n, n, n // decimation + 4, dd.ptr, decimation, ds.ptr), "dsa")
        st = self.download(ds, np.int32, 3)
        y = self.download(do, c64, int(st[2]))
        return y, (int(st[0]), float(st[1:2].view(f32)[0]), int(st[2]))

    def fir_decimate_cc(self, x, decimation, taps):
        x2, squeeze = self._2d(x, c64); taps = np.ascontiguousarray(taps, f32)
        s, n = x2.shape
        opitch = n // max(int(decimation), 1) + 2
        di = self.upload(x2); dt = self.upload(taps); do = self.alloc(8 * s * opitch + 64)
        no = self.check(self.L.csdr_amd_fir_decimate_cc(self.h, di.ptr, do.ptr, s, n, n, opitch, decimation, dt.ptr, taps.size), "fir_decimate_cc")
        y = self.download(do, c64, s * opitch).reshape(s, opitch)[:, :no]
        return y[0].copy() if squeeze else y.copy()

    def fir_ff(self, x, taps):
        x2, squeeze = self._2d(x, f32); taps = np.ascontiguousarray(taps, f32)
        s, n = x2.shape
        di = self.upload(x2); dt = self.upload(taps); do = self.alloc(4 * s * n + 64)
        no = self.check(self.L.csdr_amd_fir_ff(self.h, di.ptr, do.ptr, s, n, n, n, dt.ptr, taps.size), "fir_ff")
        y = self.download(do, f32, s * n).reshape(s, n)[:, :no]
        return y[0].copy() if squeeze else y.copy()


    # ---- f2 blocks
    def _cf_to_f(self, fn, x, *extra):
        x = np.ascontiguousarray(x, c64).ravel()
        di = self.upload(x); do = self.alloc(4 * x.size + 64)
        self.check(fn(self.h, di.ptr, do.ptr, x.size, *extra), fn.__name__)
        return self.download(do, f32, x.size)

    def amdemod_cf(self, x): return self._cf_to_f(self.L.csdr_amd_amdemod_cf, x)
    def amdemod_estimator_cf(self, x, alpha=0.0, beta=0.0): return self._cf_to_f(self.L.csdr_amd_amdemod_estimator_cf, x, alpha, beta)
    def realpart_cf(self, x): return self._cf_to_f(self.L.csdr_amd_realpart_cf, x)
    def logpower_cf(self, x, add_db=0.0): return self._cf_to_f(self.L.csdr_amd_logpower_cf, x, add_db)

    def fmdemod_atan_cf(self, x, last_phase=None, calls=1):
        x2, squeeze = self._2d(x, c64)
        s, n = x2.shape
        lp = np.zeros(s, f32) if last_phase is None else np.ascontiguousarray(last_phase, f32).reshape(s)
        di = self.upload(x2); do = self.alloc(4 * s * n + 64); dl = self.upload(lp)
        per = (n + calls - 1) // calls; at = 0
        while at < n:
            k = min(per, n - at)
            self.check(self.L.csdr_amd_fmdemod_atan_cf(self.h, di.at(8 * at), do.at(4 * at), s, k, n, n, dl.ptr), "fmdemod_atan"); at += k
        y = self.download(do, f32, s * n).reshape(s, n); lo = self.download(dl, f32, s)
        return (y[0], lo[0]) if squeeze else (y, lo)

    def dcblock_ff(self, x, a=0.0, state=None, calls=1):
        x2, squeeze = self._2d(x, f32)
        s, n = x2.shape
        st = np.zeros(2 * s, f32) if state is None else np.ascontiguousarray(state, f32).reshape(2 * s)
        di = self.upload(x2); do = self.alloc(4 * s * n + 64); ds = self.upload(st)
        per = (n + calls - 1) // calls; at = 0
        while at < n:
            k = min(per, n - at)
            self.check(self.L.csdr_amd_dcblock_ff(self.h, di.at(4 * at), do.at(4 * at), s, k, n, n, a, ds.ptr), "dcblock"); at += k
        y = self.download(do, f32, s * n).reshape(s, n); so = self.download(ds, f32, 2 * s).reshape(s, 2)
        return (y[0], so[0]) if squeeze else (y, so)

    def fastdcblock_ff(self, x, block=1024, last_dc=None, calls=1):
        x2, squeeze = self._2d(x, f32)
        s, n = x2.shape; nb = n // block
        ld = np.zeros(s, f32) if last_dc is None else np.ascontiguousarray(last_dc, f32).reshape(s)
        di = self.upload(x2); do = self.alloc(4 * s * n + 64); dl = self.upload(ld)
        per = max(1, (nb + calls - 1) // calls); b = 0
        while b < nb:
            k = min(per, nb - b)
            self.check(self.L.csdr_amd_fastdcblock_ff(self.h, di.at(4 * b * block), do.at(4 * b * block), s, k, block, n, n, dl.ptr), "fastdcblock"); b += k
        y = self.download(do, f32, s * n).reshape(s, n)[:, :nb * block]; lo = self.download(dl, f32, s)
        return (y[0].copy(), lo[0]) if squeeze else (y.copy(), lo)

    def agc_ff(self, x, block=1024, hang_time=200, reference=0.2, attack_rate=0.01, decay_rate=0.0001, max_gain=65536.0,
               attack_wait=0, filter_alpha=0.999, last_gain=None):
        x2, squeeze = self._2d(x, f32)
        s, n = x2.shape
        lg = np.ones(s, f32) if last_gain is None else np.ascontiguousarray(last_gain, f32).reshape(s)
        di = self.upload(x2); do = self.alloc(4 * s * n + 64); dl = self.upload(lg)
        self.check(self.L.csdr_amd_agc_ff(self.h, di.ptr, do.ptr, s, n, block, n, n, reference, attack_rate, decay_rate, max_gain, hang_time, attack_wait,
                                          filter_alpha, dl.ptr), "agc_ff")
        y = self.download(do, f32, s * n).reshape(s, n); lo = self.download(dl, f32, s)
        return (y[0], lo[0]) if squeeze else (y, lo)

    def precalculate_window(self, size, window="HAMMING"):
        w = np.zeros(size, f32); self.L.csdr_amd_precalculate_window(_hp(w), size, WINDOWS[window]); return w

    def fft_cc(self, x, fft_size, every_n, window="HAMMING", calls=1):
        x = np.ascontiguousarray(x, c64).ravel()
        frames_max = x.size // every_n + 1
        f = self.L.csdr_amd_fftcc_create(self.h, fft_size, every_n, WINDOWS[window], frames_max)
        if not f:
            raise CsdrAmdError(self.err())
        di = self.upload(x); do = self.alloc(8 * fft_size * frames_max + 64)
        per = (x.size + calls - 1) // calls; at = 0; total = 0; left = 0
        while at < x.size or left:
            take = min(per, x.size - at) + left
            start = at - left
            cons = C.c_size_t(0)
            nf = self.L.csdr_amd_fftcc_process(f, di.at(8 * start), take, do.at(8 * fft_size * total), C.byref(cons))
            self.check(nf, "fft_cc")
            total += nf
            at = start + take; left = take - cons.value
            if at >= x.size and (nf == 0 or left < every_n):
                break
        y = self.download(do, c64, fft_size * total)
        self.L.csdr_amd_fftcc_destroy(f)
        return y


    # ---- f3: IMA ADPCM
    def encode_ima_adpcm_i16_u8(self, x, state=None, calls=1):
        x2, squeeze = self._2d(x, np.int16)
        s, n = x2.shape
        st = np.zeros(2 * s, np.int32) if state is None else np.ascontiguousarray(state, np.int32).reshape(2 * s)
        di = self.upload(x2); do = self.alloc(s * (n // 2) + 64); ds = self.upload(st)
        per = ((n + calls - 1) // calls + 1) & ~1; at = 0
        while at < n:
            k = min(per, n - at)
            self.check(self.L.csdr_amd_encode_ima_adpcm_i16_u8(self.h, di.at(2 * at), do.at(at // 2), s, k, n, n // 2, ds.ptr), "adpcm encode"); at += k
        y = self.download(do, np.uint8, s * (n // 2)).reshape(s, n // 2); so = self.download(ds, np.int32, 2 * s).reshape(s, 2)
        return (y[0], so[0]) if squeeze else (y, so)

    def decode_ima_adpcm_u8_i16(self, x, state=None, calls=1):
        x2, squeeze = self._2d(x, np.uint8)
        s, n = x2.shape
        st = np.zeros(2 * s, np.int32) if state is None else np.ascontiguousarray(state, np.int32).reshape(2 * s)
        di = self.upload(x2); do = self.alloc(4 * s * n + 64); ds = self.upload(st)
        per = (n + calls - 1) // calls; at = 0
        while at < n:
            k = min(per, n - at)
            self.check(self.L.csdr_amd_decode_ima_adpcm_u8_i16(self.h, di.at(at), do.at(4 * at), s, k, n, 2 * n, ds.ptr), "adpcm decode"); at += k
        y = self.download(do, np.int16, 2 * s * n).reshape(s, 2 * n); so = self.download(ds, np.int32, 2 * s).reshape(s, 2)
        return (y[0], so[0]) if squeeze else (y, so)

    def compress_fft_adpcm_f_u8(self, x, fft_size):
        x = np.ascontiguousarray(x, f32).ravel(); nb = x.size // fft_size; ob = (fft_size + 10) // 2
        di = self.upload(x); do = self.alloc(nb * ob + 64)
        self.check(self.L.csdr_amd_compress_fft_adpcm_f_u8(self.h, di.ptr, do.ptr, nb, fft_size), "compress_fft")
        return self.download(do, np.uint8, nb * ob)

    def fmdemod_quadri_cf(self, x, last=None):
        x2, squeeze = self._2d(x, c64)
        s, n = x2.shape
        lst = np.zeros(s, c64) if last is None else np.ascontiguousarray(last, c64).reshape(s)
        di = self.upload(x2); do = self.alloc(4 * s * n + 64); dl = self.upload(lst)
        self.check(self.L.csdr_amd_fmdemod_quadri_cf(self.h, di.ptr, do.ptr, s, n, n, n, dl.ptr), "fmdemod")
        y = self.download(do, f32, s * n).reshape(s, n); lo = self.download(dl, c64, s)
        return (y[0], lo[0]) if squeeze else (y, lo)

    def limit_ff(self, x, m=1.0):
        x = np.ascontiguousarray(x, f32); di = self.upload(x); do = self.alloc(x.nbytes + 64)
        self.check(self.L.csdr_amd_limit_ff(self.h, di.ptr, do.ptr, x.size, m), "limit"); return self.download(do, f32, x.size).reshape(x.shape)

    def gain_ff(self, x, g):
        x = np.ascontiguousarray(x, f32); di = self.upload(x); do = self.alloc(x.nbytes + 64)
        self.check(self.L.csdr_amd_gain_ff(self.h, di.ptr, do.ptr, x.size, g), "gain"); return self.download(do, f32, x.size).reshape(x.shape)

    def deemphasis_wfm_ff(self, x, tau, sample_rate, last=None):
        x2, squeeze = self._2d(x, f32)
        s, n = x2.shape
        lst = np.zeros(s, f32) if last is None else np.ascontiguousarray(last, f32).reshape(s)
        di = self.upload(x2); do = self.alloc(4 * s * n + 64); dl = self.upload(lst)
        self.check(self.L.csdr_amd_deemphasis_wfm_ff(self.h, di.ptr, do.ptr, s, n, n, n, tau, int(sample_rate), dl.ptr), "deemph")
        y = self.download(do, f32, s * n).reshape(s, n); lo = self.download(dl, f32, s)
        return (y[0], lo[0]) if squeeze else (y, lo)

    def fastagc_ff(self, x, block=1024, reference=1.0, calls=1):
        """x: [n] or [streams, n]; whole blocks only; `calls` splits the blocks over several API calls (state carry)."""
        x2, squeeze = self._2d(x, f32)
        s, n = x2.shape; nb = n // block
        di = self.upload(x2); do = self.alloc(4 * s * n + 64)
        dst = self.upload(np.zeros(s * (2 * block + 4), f32))
        per = max(1, (nb + calls - 1) // calls); b = 0
        while b < nb:
            k = min(per, nb - b)
            self.check(self.L.csdr_amd_fastagc_ff(self.h, di.at(4 * b * block), do.at(4 * b * block), s, k, block, n, n, reference, dst.ptr), "fastagc")
            b += k
        y = self.download(do, f32, s * n).reshape(s, n)[:, :nb * block]
        return y[0].copy() if squeeze else y.copy()

    def fractional_decimator_ff(self, x, rate, num_poly_points=12, taps=None, bufsize=None):
        x2, squeeze = self._2d(x, f32)
        s, n = x2.shape
        tp = None if taps is None else np.ascontiguousarray(taps, f32)
        d = self.L.csdr_amd_fracdec_create(rate, num_poly_points, None if tp is None else _hp(tp), 0 if tp is None else tp.size)
        if not d:
            raise CsdrAmdError(self.err())
        if bufsize:                                   # the CLI's window loop (csdr.c:1511-1524) instead of one call over the whole array
            self.L.csdr_amd_fracdec_set_cli_bufsize(d, bufsize)
        di = self.upload(x2); do = self.alloc(4 * s * n + 64); proc = C.c_int(0)
        no = self.check(self.L.csdr_amd_fractional_decimator_ff(self.h, d, di.ptr, do.ptr, s, n, n, n, C.byref(proc)), "fracdec")
        self.sync(); self.L.csdr_amd_fracdec_destroy(d)
        y = self.download(do, f32, s * n).reshape(s, n)[:, :no]
        return y[0].copy() if squeeze else y.copy()

    def fft_c2c(self, x, forward=True):
        x = np.ascontiguousarray(x, c64); di = self.upload(x); do = self.alloc(x.nbytes + 64)
        self.check(self.L.csdr_amd_fft_c2c(self.h, di.ptr, do.ptr, x.size, int(forward)), "fft"); return self.download(do, c64, x.size)

    def bandpass_fir_fft_cc(self, x, taps, fft_size, blocks_per_call=None):
        x2, squeeze = self._2d(x, c64); taps = np.ascontiguousarray(taps, c64)
        s, n = x2.shape
        inp = fft_size - taps.size + 1; nb = n // inp
        per = nb if not blocks_per_call else blocks_per_call
        f = self.L.csdr_amd_fftfilt_create(self.h, fft_size, _hp(taps), taps.size, s, max(per, 1))
        if not f:
            raise CsdrAmdError(self.err())
        di = self.upload(x2); do = self.alloc(x2.nbytes + 64)
        b = 0
        while b < nb:
            k = min(per, nb - b)
            self.check(self.L.csdr_amd_fftfilt_process(f, di.at(8 * b * inp), do.at(8 * b * inp), k, n, n), "fftfilt")
            b += k
        y = self.download(do, c64, s * n).reshape(s, n)[:, :nb * inp]
        self.L.csdr_amd_fftfilt_destroy(f)
        return y[0].copy() if squeeze else y.copy()

    def fastddc_init(self, tbw, decimation, shift_rate):
        d = FastDDC(); err = self.L.csdr_amd_fastddc_init(C.byref(d), tbw, decimation, shift_rate); return d, err

    def fastddc_fwd_cc(self, x, ddc, blocks_per_call=None):
        x = np.ascontiguousarray(x, c64); nb = x.size // ddc.input_size
        per = nb if not blocks_per_call else blocks_per_call
        f = self.L.csdr_amd_fastddc_fwd_create(self.h, C.byref(ddc), max(per, 1))
        if not f:
            raise CsdrAmdError(self.err())
        di = self.upload(x); do = self.alloc(8 * nb * ddc.fft_size + 64)
        b = 0
        while b < nb:
            k = min(per, nb - b)
            self.check(self.L.csdr_amd_fastddc_fwd_process(f, di.at(8 * b * ddc.input_size), do.at(8 * b * ddc.fft_size), k), "fastddc_fwd")
            b += k
        y = self.download(do, c64, nb * ddc.fft_size).reshape(nb, ddc.fft_size)
        self.L.csdr_amd_fastddc_fwd_destroy(f)
        return y

    def fastddc_inv_cc(self, spectra, tbw, decimation, shift_rates, window="HAMMING", blocks_per_call=None):
        """spectra [n_blocks, fft] -> list of per-channel outputs."""
        spectra = np.ascontiguousarray(spectra, c64); nb = spectra.shape[0]
        rates = np.ascontiguousarray(shift_rates, f32); nc = rates.size
        per = nb if not blocks_per_call else blocks_per_call
        f = self.L.csdr_amd_fastddc_inv_create(self.h, tbw, decimation, _hp(rates), nc, WINDOWS[window], max(per, 1))
        if not f:
            raise CsdrAmdError(self.err())
        fft = spectra.shape[1]
        di = self.upload(spectra)
        outs = [[] for _ in range(nc)]
        b = 0
        while b < nb:
            k = min(per, nb - b)
            pitch = self.L.csdr_amd_fastddc_inv_max_output(f, k) + 8
            do = self.alloc(8 * nc * pitch)
            counts = np.zeros(nc, np.int32)
            self.check(self.L.csdr_amd_fastddc_inv_process(f, di.at(8 * b * fft), k, do.ptr, pitch, _hp(counts)), "fastddc_inv")
            y = self.download(do, c64, nc * pitch).reshape(nc, pitch)
            for c in range(nc):
                outs[c].append(y[c, :counts[c]].copy())
            b += k
        self.L.csdr_amd_fastddc_inv_destroy(f)
        return [np.concatenate(o) for o in outs]

    def fastddc_bank(self, x, tbw, decimation, shift_rates, window="HAMMING", blocks_per_call=None, retune=None, schedule=None, retunes=None):
        """forward + inverse in one object (csdr_amd_fastddc_bank_*): x = wideband samples -> list of per-channel outputs.
        retune = (call_index, channel, rate): applied before that call.  schedule = explicit list of blocks per call (instead of blocks_per_call);
        retunes = {call_index: [(channel, rate), ...]}.  x: complex64, or interleaved IQ as int16 / uint8 (csdr_amd_fastddc_bank_process_s16 / _u8)."""
        x, sfx, es = _bank_input(x)
        rates = np.ascontiguousarray(shift_rates, f32); nc = rates.size
        ddc, _ = self.fastddc_init(tbw, decimation, 0.0)
        nb = (x.size if es == 8 else x.size // 2) // ddc.input_size
        process = getattr(self.L, "csdr_amd_fastddc_bank_process" + sfx)
        per = nb if not blocks_per_call else blocks_per_call
        if schedule:
            per = max(schedule)
        bk = self.L.csdr_amd_fastddc_bank_create(self.h, tbw, decimation, _hp(rates), nc, WINDOWS[window], max(per, 1))
        if not bk:
            raise CsdrAmdError(self.err())
        di = self.upload(x)
        outs = [[] for _ in range(nc)]
        b = 0; call = 0
        while b < nb:
            k = min(per, nb - b) if not schedule else min(schedule[call % len(schedule)], nb - b)
            if retune and retune[0] == call:
                self.check(self.L.csdr_amd_fastddc_bank_set_rate(bk, retune[1], retune[2]), "bank_set_rate")
            for ch, rt in (retunes or {}).get(call, []):
                self.check(self.L.csdr_amd_fastddc_bank_set_rate(bk, ch, rt), "bank_set_rate")
            pitch = self.L.csdr_amd_fastddc_bank_max_output(bk, k) + 8
            do = self.alloc(8 * nc * pitch)
            counts = np.zeros(nc, np.int32)
            self.check(process(bk, di.at(es * b * ddc.input_size), k, do.ptr, pitch, _hp(counts)), "fastddc_bank")
            y = self.download(do, c64, nc * pitch).reshape(nc, pitch)
            for c in range(nc):
                outs[c].append(y[c, :counts[c]].copy())
            b += k; call += 1
        self.last_ddc_kernel = self.L.csdr_amd_fastddc_inv_kernel_name(self.L.csdr_amd_fastddc_bank_inverse(bk)).decode()
        self.L.csdr_amd_fastddc_bank_destroy(bk)
        return [np.concatenate(o) for o in outs]

    # (the multi-rank form of the bank on ONE GPU: sharded_bank_loopback below, module level -- one Context per rank thread)

    def wfm_chain(self, iq_u8, shift_rate, decimation, taps, frac_rate=5, tau=50e-6, audio_rate=48000, block=None, pitch_pad=0, retunes=None, want_float=True,
                  out_per_call=False):
        """iq_u8: [2n] or [streams, 2n] uint8 -> (s16 [streams, na], float audio [streams, na]); `block` = samples per call (or a list of call sizes);
        `pitch_pad` = extra bytes of row pitch (multiple of 16).  shift_rate: one float, or one per stream (csdr_amd_wfm_create_rates);
        retunes: {call index: [(stream, rate), ...]} applied in front of that call.  want_float=False: s16 only (the kernel's line-collecting store path).
        out_per_call: every call writes at the START of the (16-byte aligned) output rows, as a streaming caller with one output buffer does -- the kernel's
        aligned store path on every call (otherwise call k's audio follows call k - 1's in one row: aligned only by chance).
        The front-end kernel of the last call is left in `self.last_wfm_kernel`."""
        x2, squeeze = self._2d(iq_u8, np.uint8)
        s, nbytes = x2.shape; n = nbytes // 2
        pitch = (nbytes + 15) // 16 * 16 + pitch_pad
        xx = np.zeros((s, pitch), np.uint8); xx[:, :nbytes] = x2
        taps = np.ascontiguousarray(taps, f32)
        sched = list(block) if isinstance(block, (list, tuple)) else None
        block = n if block is None else (max(sched) if sched else block)
        if np.ndim(shift_rate) == 0:
            w = self.L.csdr_amd_wfm_create(self.h, s, shift_rate, decimation, _hp(taps), taps.size, frac_rate, tau, audio_rate, max(block, 1024))
        else:
            rates = np.ascontiguousarray(shift_rate, f32); assert rates.size == s
            w = self.L.csdr_amd_wfm_create_rates(self.h, s, _hp(rates), decimation, _hp(taps), taps.size, frac_rate, tau, audio_rate, max(block, 1024))
        if not w:
            raise CsdrAmdError(self.err())
        di = self.upload(xx)
        apitch = (n // (decimation * frac_rate) + 64 + 63) // 64 * 64
        ds = self.alloc(2 * s * apitch); df = self.alloc(4 * s * apitch) if want_float else None
        pos = 0; na = 0; call = 0; parts_s = []; parts_f = []
        while pos < n:
            for st, r in (retunes or {}).get(call, []):
                self.check(self.L.csdr_amd_wfm_set_rate(w, st, r), "wfm_set_rate")
            k = min(sched[call] if (sched and call < len(sched)) else block, n - pos); call += 1
            if out_per_call:
                got = self.check(self.L.csdr_amd_wfm_process(w, di.at(2 * pos), pitch, k, ds.ptr, df.ptr if want_float else None, apitch), "wfm_process")
                parts_s.append(self.download(ds, np.int16, s * apitch).reshape(s, apitch)[:, :got].copy())
                if want_float: parts_f.append(self.download(df, f32, s * apitch).reshape(s, apitch)[:, :got].copy())
            else:
                got = self.check(self.L.csdr_amd_wfm_process(w, di.at(2 * pos), pitch, k, ds.at(2 * na), df.at(4 * na) if want_float else None, apitch), "wfm_process")
            pos += k; na += got
        if out_per_call:
            s16 = np.concatenate(parts_s, axis=1) if parts_s else np.zeros((s, 0), np.int16)
            af = np.concatenate(parts_f, axis=1) if parts_f else np.zeros((s, 0), f32)
        else:
            s16 = self.download(ds, np.int16, s * apitch).reshape(s, apitch)[:, :na]
            af = self.download(df, f32, s * apitch).reshape(s, apitch)[:, :na] if want_float else np.zeros((s, 0), f32)
        self.last_wfm_kernel = self.L.csdr_amd_wfm_kernel_name(w).decode()
        self.L.csdr_amd_wfm_destroy(w)
        return (s16[0].copy(), af[0].copy()) if squeeze else (s16.copy(), af.copy())

    def ddc_u8(self, iq_u8, shift_rate, decimation, taps, block=None, pitch_pad=0, retunes=None):
        """Fused front end convert_u8_f | shift_addition_cc | fir_decimate_cc (csdr_amd_ddc_*): iq_u8 [2n] or [streams, 2n] uint8 ->
        complex64 [streams, n_out].  `block` = samples per call; `pitch_pad` = extra bytes of row pitch (a pitch that is not a multiple of
        128 selects the plain kernel).  The kernel of the last call is left in `self.last_ddc_kernel`, all kernels used in `self.ddc_kernels`.
        shift_rate: one float, or one per stream (csdr_amd_ddc_create_rates); retunes: {call index: [(stream, rate), ...]} applied in front of that call."""
        x2, squeeze = self._2d(iq_u8, np.uint8)
        s, nbytes = x2.shape; n = nbytes // 2
        pitch = (nbytes + 127) // 128 * 128 + pitch_pad
        xx = np.full((s, pitch), 0x80, np.uint8); xx[:, :nbytes] = x2
        taps = np.ascontiguousarray(taps, f32)
        sched = list(block) if isinstance(block, (list, tuple)) else None          # `block` may be a list of call sizes (the rest of the stream follows in one call)
        block = n if block is None else (max(sched) if sched else block)
        if np.ndim(shift_rate) == 0:
            d = self.L.csdr_amd_ddc_create(self.h, s, shift_rate, decimation, _hp(taps), taps.size, max(block, 1024))
        else:
            rates = np.ascontiguousarray(shift_rate, f32); assert rates.size == s
            d = self.L.csdr_amd_ddc_create_rates(self.h, s, _hp(rates), decimation, _hp(taps), taps.size, max(block, 1024))
        if not d:
            raise CsdrAmdError(self.err())
        di = self.upload(xx)
        opitch = n // decimation + 64
        do = self.alloc(8 * s * opitch)
        pos = 0; no = 0; self.ddc_kernels = set(); call = 0
        while pos < n:
            for st, r in (retunes or {}).get(call, []):
                self.check(self.L.csdr_amd_ddc_set_rate(d, st, r), "ddc_set_rate")
            k = min(sched[call] if (sched and call < len(sched)) else block, n - pos); call += 1
            got = self.check(self.L.csdr_amd_ddc_process(d, di.at(2 * pos), pitch, k, do.at(8 * no), opitch), "ddc_process")
            self.ddc_kernels.add(self.L.csdr_amd_ddc_kernel_name(d).decode())
            pos += k; no += got
        y = self.download(do, c64, s * opitch).reshape(s, opitch)[:, :no]
        self.last_ddc_kernel = self.L.csdr_amd_ddc_kernel_name(d).decode()
        self.L.csdr_amd_ddc_destroy(d)
        return y[0].copy() if squeeze else y.copy()

    def wfm_ring_chain(self, iq_u8, shift_rate, decimation, taps, block=16384, n_slots=8, frac_rate=5, tau=50e-6, audio_rate=48000, retunes=None, in_flight=None,
                       idle_us=None, life_ms=None, pause_every=None, pause_s=0.0):
        """The resident form (csdr_amd_wfm_ring_*): iq_u8 [streams, 2n] uint8, n a multiple of `block` -> s16 [streams, na] (all blocks' audio in stream order).
        Blocks are copied into the input ring slot by slot (hipMemcpy H2D), posted, and collected `in_flight` (default n_slots - 2) blocks later.
        retunes: {block index: rate} applied in front of that block.  pause_every / pause_s: sleep so long every so many blocks (the grid leaves and is relaunched).
        Leaves (launches, grid) in self.last_ring."""
        import time
        x2, squeeze = self._2d(iq_u8, np.uint8)
        s, nbytes = x2.shape; n = nbytes // 2
        assert n % block == 0
        nb = n // block
        taps = np.ascontiguousarray(taps, f32)
        r = self.L.csdr_amd_wfm_ring_create(self.h, s, shift_rate, decimation, _hp(taps), taps.size, frac_rate, tau, audio_rate, block, n_slots)
        if not r:
            raise CsdrAmdError(self.err())
        try:
            if idle_us is not None or life_ms is not None:
                self.check(self.L.csdr_amd_wfm_ring_set_timeouts(r, 200.0 if idle_us is None else idle_us, 250.0 if life_ms is None else life_ms), "ring_set_timeouts")
            depth = (n_slots - 2) if in_flight is None else in_flight
            outs = []; pitch = C.c_size_t(0); opitch = C.c_size_t(0)

            def collect(k):
                na = self.check(self.L.csdr_amd_wfm_ring_wait(r, k, 0.0), "ring_wait")
                po = self.L.csdr_amd_wfm_ring_output(r, k, C.byref(opitch))
                buf = np.empty((s, opitch.value), np.int16)
                self.check(self.L.csdr_amd_d2h(self.h, _hp(buf), po, buf.nbytes), "d2h")
                outs.append(buf[:, :na].copy())
            for k in range(nb):
                if retunes and k in retunes:
                    while len(outs) < k:
                        collect(len(outs))
                    self.check(self.L.csdr_amd_wfm_ring_set_rate(r, retunes[k]), "ring_set_rate")
                if pause_every and k and k % pause_every == 0:
                    time.sleep(pause_s)
                self.check(self.L.csdr_amd_wfm_ring_acquire(r, k, 0.0), "ring_acquire")
                pi = self.L.csdr_amd_wfm_ring_input(r, k, C.byref(pitch))
                blk = np.zeros((s, pitch.value), np.uint8); blk[:, :2 * block] = x2[:, 2 * k * block:2 * (k + 1) * block]
                self.check(self.L.csdr_amd_h2d(self.h, pi, _hp(blk), blk.nbytes), "h2d")
                got = self.L.csdr_amd_wfm_ring_submit(r)
                if got != k:
                    raise CsdrAmdError("ring_submit: %d (%s)" % (got, self.err()))
                while len(outs) + depth <= k:
                    collect(len(outs))
            while len(outs) < nb:
                collect(len(outs))
            self.last_ring = {"launches": self.L.csdr_amd_wfm_ring_launches(r), "grid": self.L.csdr_amd_wfm_ring_grid(r)}
        finally:
            self.L.csdr_amd_wfm_ring_destroy(r)
        y = np.concatenate(outs, axis=1)
        return y[0].copy() if squeeze else y

    def nfm_chain(self, iq_u8, shift_rate, decimation=50, tbw=0.005, audio_rate=48000, agc_block=1024, block=None, retunes=None):
        """BASELINE config 5 / README.md:87 through the chain object csdr_amd_nfm_* (matrix-core front end + audio-rate back end):
        iq_u8 [streams, 2n] uint8 -> (s16 [streams, na], float audio [streams, na]); `block` = samples per call.
        shift_rate: one float, or one per stream (csdr_amd_nfm_create_rates); retunes: {call index: [(stream, rate), ...]}."""
        x2, squeeze = self._2d(iq_u8, np.uint8)
        S, nbytes = x2.shape; n = nbytes // 2
        pitch = (nbytes + 127) // 128 * 128
        xx = np.full((S, pitch), 0x80, np.uint8); xx[:, :nbytes] = x2
        nt = self.firdes_filter_len(tbw)
        taps = np.ascontiguousarray(self.firdes_lowpass_f(nt, 0.5 / decimation), f32)
        sched = list(block) if isinstance(block, (list, tuple)) else None
        block = n if block is None else (max(sched) if sched else block)
        if np.ndim(shift_rate) == 0:
            w = self.L.csdr_amd_nfm_create(self.h, S, shift_rate, decimation, _hp(taps), taps.size, audio_rate, agc_block, 1.0, 1.0, max(block, 1024))
        else:
            rates = np.ascontiguousarray(shift_rate, f32); assert rates.size == S
            w = self.L.csdr_amd_nfm_create_rates(self.h, S, _hp(rates), decimation, _hp(taps), taps.size, audio_rate, agc_block, 1.0, 1.0, max(block, 1024))
        if not w:
            raise CsdrAmdError(self.err())
        di = self.upload(xx)
        apitch = n // decimation + 1024 + 64
        ds = self.alloc(2 * S * apitch); df = self.alloc(4 * S * apitch)
        pos = 0; na = 0; call = 0
        while pos < n:
            for st, r in (retunes or {}).get(call, []):
                self.check(self.L.csdr_amd_nfm_set_rate(w, st, r), "nfm_set_rate")
            k = min(sched[call] if (sched and call < len(sched)) else block, n - pos); call += 1
            got = self.check(self.L.csdr_amd_nfm_process(w, di.at(2 * pos), pitch, k, ds.at(2 * na), df.at(4 * na), apitch), "nfm_process")
            pos += k; na += got
        pcm = self.download(ds, np.int16, S * apitch).reshape(S, apitch)[:, :na]
        af = self.download(df, f32, S * apitch).reshape(S, apitch)[:, :na]
        self.last_ddc_kernel = self.L.csdr_amd_ddc_kernel_name(self.L.csdr_amd_nfm_front_end(w)).decode()
        self.L.csdr_amd_nfm_destroy(w)
        return (pcm[0].copy(), af[0].copy()) if squeeze else (pcm.copy(), af.copy())

    def nfm_chain_unfused(self, iq_u8, shift_rate, decimation=50, tbw=0.005, audio_rate=48000, agc_block=1024):
        """BASELINE config 5 / README.md:87, stage by stage through the device batch API with the data resident on the GPU between
        stages: convert_u8_f | shift_addition_cc | fir_decimate_cc D tbw HAMMING | fmdemod_quadri_cf | limit_ff | deemphasis_nfm_ff |
        fastagc_ff | convert_f_s16.   iq_u8: [streams, 2n] uint8  ->  (s16 [streams, na], float audio [streams, na])."""
        x2, squeeze = self._2d(iq_u8, np.uint8)
        S, nbytes = x2.shape; n = nbytes // 2
        L = self.L
        d_u8 = self.upload(x2)
        d_f = self.alloc(4 * S * nbytes + 64)
        self.check(L.csdr_amd_convert_u8_f(self.h, d_u8.ptr, d_f.ptr, S * nbytes), "convert_u8_f")
        d_sh = self.alloc(8 * S * n + 64)
        ph = C.c_float(0.0)
        self.check(L.csdr_amd_shift_cc(self.h, SHIFT["addition"], shift_rate, C.byref(ph), d_f.ptr, d_sh.ptr, S, n, n, n, 1024, 0), "shift")
        nt = self.firdes_filter_len(tbw)
        taps = self.upload(self.firdes_lowpass_f(nt, 0.5 / decimation))
        pd = n // decimation + 2
        d_dec = self.alloc(8 * S * pd + 64)
        nd = self.check(L.csdr_amd_fir_decimate_cc(self.h, d_sh.ptr, d_dec.ptr, S, n, n, pd, decimation, taps.ptr, nt), "fir_decimate_cc")
        d_dem = self.alloc(4 * S * pd + 64); d_last = self.upload(np.zeros(S, c64))
        self.check(L.csdr_amd_fmdemod_quadri_cf(self.h, d_dec.ptr, d_dem.ptr, S, nd, pd, pd, d_last.ptr), "fmdemod")
        pre = 1024                         # `csdr deemphasis_nfm_ff` filters  the_bufsize zeros ++ stream  (csdr.c:1076-1081)
        pd2 = pd + pre
        d_lim0 = self.alloc(4 * S * pd + 64)
        self.check(L.csdr_amd_limit_ff(self.h, d_dem.ptr, d_lim0.ptr, S * pd, 1.0), "limit")
        lim = np.zeros((S, pd2), f32); lim[:, pre:] = self.download(d_lim0, f32, S * pd).reshape(S, pd)
        d_lim = self.upload(lim)
        dtaps_h = self.nfm_taps(audio_rate)
        d_dt = self.upload(dtaps_h)
        d_de = self.alloc(4 * S * pd2 + 64)
        ne = self.check(L.csdr_amd_fir_ff(self.h, d_lim.ptr, d_de.ptr, S, nd + pre, pd2, pd2, d_dt.ptr, dtaps_h.size), "deemphasis_nfm")
        nb = ne // agc_block
        d_agc = self.alloc(4 * S * pd2 + 64)
        d_st = self.upload(np.zeros(S * (2 * agc_block + 4), f32))
        self.check(L.csdr_amd_fastagc_ff(self.h, d_de.ptr, d_agc.ptr, S, nb, agc_block, pd2, pd2, 1.0, d_st.ptr), "fastagc")
        na = nb * agc_block
        d_pcm = self.alloc(2 * S * pd2 + 64)
        self.check(L.csdr_amd_convert_f_s16(self.h, d_agc.ptr, d_pcm.ptr, S * pd2), "convert_f_s16")
        af = self.download(d_agc, f32, S * pd2).reshape(S, pd2)[:, :na]
        pcm = self.download(d_pcm, np.int16, S * pd2).reshape(S, pd2)[:, :na]
        return (pcm[0].copy(), af[0].copy()) if squeeze else (pcm.copy(), af.copy())


SHARD = {"channels": 0, "blocks": 1}


def _bank_input(x):
    """the wideband stream as the bank takes it: (array, entry-point suffix, bytes per complex sample)"""
    x = np.asarray(x)
    if x.dtype == np.int16:
        return np.ascontiguousarray(x), "_s16", 4
    if x.dtype == np.uint8:
        return np.ascontiguousarray(x), "_u8", 2
    return np.ascontiguousarray(x, c64), "", 8


def sharded_bank_loopback(world, x, tbw, decimation, shift_rates, schedule, mode="blocks", window="HAMMING", retunes=None, pipelined=True,
                          local_input=False, device=0, retune_while_staged=False, superseded_retune=False):
    """The multi-rank fastddc bank (csdr_amd_fastddc_bank_create_sharded_by) run for real on ONE GPU: `world` rank threads, one Context each, joined by the
    library's loopback communicator (every exchange = stream-ordered device copies).  x = the wideband stream (on rank 0; local_input: every rank is handed
    its own run of each batch instead), schedule = blocks per batch, retunes = {batch index: [(global channel, rate), ...]} applied before that batch.
    pipelined: submit(k + 1) is queued before collect(k) wherever no retune sits in between -- or, with retune_while_staged, everywhere: batch k + 1's retunes are
    then issued while batch k is still staged (they must leave batch k alone and apply from batch k + 1 on, in both sharding modes).  superseded_retune (with
    retune_while_staged, unpipelined): the retune issued while batch k is staged goes to a DECOY rate and the real one follows after collect(k), when nothing is
    staged -- the held-back decoy must not be replayed on top of it at collect(k + 1).  Returns the per-channel outputs (all channels, gathered from the ranks' slices)."""
    import threading
    L = lib()
    x, sfx, es = _bank_input(x)                          # complex64, or interleaved IQ as int16 / uint8 (the _s16 / _u8 entry points: the raw integers are scattered)
    spc = 1 if es == 8 else 2                            # array elements per complex sample
    rates = np.ascontiguousarray(shift_rates, f32); nc = rates.size
    retunes = retunes or {}
    grp = L.csdr_amd_loopback_create(world)
    if not grp:
        raise CsdrAmdError(L.csdr_amd_last_error().decode())
    outs = [None] * nc
    errors = []

    def rank_main(rank):
        ctx = None
        try:
            ctx = Context(device)
            comm = L.csdr_amd_comm_create_loopback(ctx.h, grp, rank)
            if not comm:
                raise CsdrAmdError(ctx.err())
            per = max(schedule)
            bank = L.csdr_amd_fastddc_bank_create_sharded_by(ctx.h, tbw, decimation, _hp(rates), nc, WINDOWS[window], per, comm, SHARD[mode])
            if not bank:
                raise CsdrAmdError(ctx.err())
            first = C.c_int(); count = C.c_int()
            L.csdr_amd_fastddc_bank_channel_slice(bank, C.byref(first), C.byref(count))
            first, count = first.value, count.value
            inp = L.csdr_amd_fastddc_bank_input_size(bank); ovl = L.csdr_amd_fastddc_bank_overlap(bank)
            starts = np.concatenate([[0], np.cumsum(schedule)])
            di = ctx.upload(x) if (rank == 0 and not local_input) else None
            mine = [[] for _ in range(count)]
            held = {}

            def submit(k):
                nb = schedule[k]
                if local_input:
                    f0 = C.c_int(); n0 = C.c_int()
                    L.csdr_amd_fastddc_bank_local_blocks(bank, nb, C.byref(f0), C.byref(n0))
                    a = (starts[k] + f0.value) * inp
                    run = np.zeros((ovl + n0.value * inp) * spc, x.dtype)          # (zeros in front of the stream: exact for complexf and s16)
                    lo = max(0, a - ovl)
                    run[(ovl - (a - lo)) * spc:] = x[lo * spc:(a + n0.value * inp) * spc]
                    held[k] = ctx.upload(run)
                    ctx.check(getattr(L, "csdr_amd_fastddc_bank_submit_local" + sfx)(bank, held[k].ptr, nb), "bank_submit_local")
                else:
                    ctx.check(getattr(L, "csdr_amd_fastddc_bank_submit" + sfx)(bank, di.at(es * starts[k] * inp) if di is not None else None, nb), "bank_submit")

            submitted = -1
            for k in range(len(schedule)):
                if not (retune_while_staged and k > 0):
                    for ch, rt in retunes.get(k, []):
                        ctx.check(L.csdr_amd_fastddc_bank_set_rate_global(bank, ch, rt), "bank_set_rate_global")
                if submitted < k:
                    submit(k); submitted = k
                if retune_while_staged:                               # batch k is staged, not collected: the next batch's retunes arrive now
                    for ch, rt in retunes.get(k + 1, []):
                        ctx.check(L.csdr_amd_fastddc_bank_set_rate_global(bank, ch, rt + 0.0517 if superseded_retune else rt), "bank_set_rate_global")
                if pipelined and k + 1 < len(schedule) and (retune_while_staged or (k + 1) not in retunes):
                    submit(k + 1); submitted = k + 1
                pitch = L.csdr_amd_fastddc_bank_max_output(bank, schedule[k]) + 8
                do = ctx.alloc(8 * count * pitch)
                ctx.check(L.csdr_amd_fastddc_bank_collect(bank, do.ptr, pitch, None), "bank_collect")
                counts = np.zeros(count, np.int32)
                ctx.check(L.csdr_amd_fastddc_bank_finish(bank, _hp(counts)), "bank_finish")
                if superseded_retune:                                 # nothing is staged now: the real rate, applied at once
                    for ch, rt in retunes.get(k + 1, []):
                        ctx.check(L.csdr_amd_fastddc_bank_set_rate_global(bank, ch, rt), "bank_set_rate_global")
                y = ctx.download(do, c64, count * pitch).reshape(count, pitch)
                for c in range(count):
                    mine[c].append(y[c, :counts[c]].copy())
                held.pop(k, None)
            for c in range(count):
                outs[first + c] = np.concatenate(mine[c])
            L.csdr_amd_fastddc_bank_destroy(bank); L.csdr_amd_comm_destroy(comm)
        except BaseException as e:      # a dead rank must not leave the others waiting at a rendezvous
            errors.append((rank, e))
            L.csdr_amd_loopback_abort(grp)
        finally:
            if ctx is not None:
                ctx.close()

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    L.csdr_amd_loopback_destroy(grp)
    if errors:
        raise CsdrAmdError("rank %d: %r" % errors[0])
    return outs
