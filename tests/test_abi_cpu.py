"""CPU-only checks of the boundary: libcsdr_amd.so loads and exports every symbol that include/*.h declares,
struct layouts match the reference ABI, and without a GPU the library fails loudly."""
import ctypes as C
import os
import re
import subprocess
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    txt = re.sub(r"//.*", "", txt)
    txt = re.sub(r"#define[^\n]*\n", "\n", txt)
    names = set()
    for m in re.finditer(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\(", txt):
        n = m.group(1)
        if n in ("defined", "sizeof") or n.isupper():
            continue
        names.add(n)
    return names


@pytest.fixture(scope="module")
def libpath():
    import csdr_amd
    if not os.path.exists(csdr_amd.LIB_PATH):
        csdr_amd.build()
    return csdr_amd.LIB_PATH


def test_exports_every_declared_symbol(libpath):
    out = subprocess.run(["nm", "-D", "--defined-only", libpath], capture_output=True, text=True, check=True).stdout
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l}
    for header in ("csdr_amd.h", "libcsdr_amd_compat.h"):
        missing = sorted(declared_symbols(header) - exported)
        assert not missing, "%s declares symbols the library does not export: %s" % (header, missing)


def test_library_loads_and_fails_loudly_without_gpu(libpath):
    import csdr_amd
    L = csdr_amd.lib()
    if L.csdr_amd_device_count() > 0:
        pytest.skip("a GPU is present; the no-GPU failure path is exercised on CPU-only hosts")
    with pytest.raises(csdr_amd.CsdrAmdError):
        csdr_amd.Context(0)
    assert b"no HIP device" in L.csdr_amd_last_error() or b"HIP" in L.csdr_amd_last_error()


def test_host_design_matches_oracle(libpath, port):
    """firdes / geometry code of the product (host side, no GPU needed) against the oracle."""
    import numpy as np
    import csdr_amd
    L = csdr_amd.lib()
    for tbw in [0.05, 0.005, 0.03, 0.001]:
        assert L.csdr_amd_firdes_filter_len(tbw) == port.firdes_filter_len(tbw)
    for x in [0, 1, 2, 3, 127, 128, 129, 65536]:
        assert L.csdr_amd_next_pow2(x) == port.next_pow2(x) and L.csdr_amd_log2n(x) == port.log2n(x)
    for (n, fc, w) in [(79, 0.05, "HAMMING"), (801, 0.01, "BLACKMAN"), (133, 0.125, "BOXCAR")]:
        t = np.zeros(n, np.float32)
        L.csdr_amd_firdes_lowpass_f(t.ctypes.data_as(C.c_void_p), n, fc, csdr_amd.WINDOWS[w])
        assert np.array_equal(t, port.firdes_lowpass_f(n, fc, w))
    t = np.zeros(4095, np.complex64)
    L.csdr_amd_firdes_bandpass_c(t.ctypes.data_as(C.c_void_p), 4095, -0.1, 0.2, 2)
    assert np.array_equal(t, port.firdes_bandpass_c(4095, -0.1, 0.2))
    for D in [2, 6, 10, 16, 50, 256]:
        for tbw in [0.05, 0.005, 0.001]:
            for s in [0.0, -0.1, 0.4, 0.123456, -0.5 + 0.5 / 256]:
                a = csdr_amd.FastDDC(); ea = L.csdr_amd_fastddc_init(C.byref(a), tbw, D, s)
                b, eb = port.fastddc_init(tbw, D, s)
                A, B = a.as_dict(), b.as_dict()
                assert ea == eb and A == B, (D, tbw, s, A, B)
    golden = np.load(os.path.join(ROOT, "tests", "golden", "nfm_deemph_taps.npz"))
    for sr in [48000, 44100, 8000, 11025]:
        p = C.c_void_p(); n = L.csdr_amd_nfm_deemph_taps(sr, C.byref(p))
        got = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_float)), (n,))
        assert np.array_equal(got.view(np.uint32), golden["sr%d" % sr].view(np.uint32))
    assert L.csdr_amd_nfm_deemph_taps(12345, None) == 0


def test_struct_layouts():
    """sizeof/offsets of the by-value structs via a tiny C program against our header."""
    src = r'''
#include <stdio.h>
#include <stddef.h>
#include "libcsdr_amd_compat.h"
int main(void){
 printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(complexf), sizeof(shift_table_data_t), sizeof(shift_addfast_data_t),
   sizeof(shift_unroll_data_t), sizeof(shift_addition_data_t), sizeof(decimating_shift_addition_status_t),
   sizeof(fastagc_ff_t), sizeof(fractional_decimator_ff_t), sizeof(fastddc_t));
 printf("%zu %zu %zu %zu %zu\n", offsetof(fastagc_ff_t, last_gain), offsetof(fractional_decimator_ff_t, taps_length),
   offsetof(fastddc_t, dsadata), offsetof(shift_unroll_data_t, size), sizeof(struct fft_plan_s));
 return 0; }'''
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(src)
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "t.c"), "-o", os.path.join(d, "t")], check=True)
        out = subprocess.run([os.path.join(d, "t")], capture_output=True, text=True, check=True).stdout.split()
    assert [int(v) for v in out] == [8, 16, 36, 24, 12, 12, 48, 72, 76, 40, 64, 64, 20, 32]


def test_reference_client_links_against_our_library(libpath, tmp_path):
    """a client compiled against the REFERENCE headers links against libcsdr_amd.so."""
    ref = "/root/reference"
    if not os.path.exists(os.path.join(ref, "libcsdr.h")):
        pytest.skip("reference headers not present on this host")
    exe = str(tmp_path / "client")
    r = subprocess.run(["gcc", "-std=gnu99", "-DUSE_FFTW", "-DLIBCSDR_GPL", "-I", ref, "-I", os.path.join(ROOT, "oracle"),
                        os.path.join(ROOT, "tests", "data", "dropin_client.c"), "-L", os.path.dirname(libpath), "-l:libcsdr_amd.so",
                        "-Wl,-rpath," + os.path.dirname(libpath), "-o", exe, "-lm"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    import shutil
    shutil.copy(exe, os.path.join(ROOT, "tests", "data", "dropin_client.bin"))      # travels to the GPU box for the -m gpu run


def test_reference_fft_malloc_macro_needs_no_fftw(libpath, tmp_path):
    """The reference's fft_fftw.h:11-12 expands fft_malloc / fft_free to fftwf_malloc / fftwf_free.  A client that uses the macro, compiled against the REFERENCE
    headers, must link against libcsdr_amd.so alone (no -lfftw3f) -- the two symbols are exported (VERDICT r3 missing #5).  Link only: nothing is executed here."""
    ref = "/root/reference"
    if not os.path.exists(os.path.join(ref, "fft_fftw.h")):
        pytest.skip("reference headers not present on this host")
    src = tmp_path / "m.c"
    src.write_text('#include "libcsdr.h"\n#include "fft_fftw.h"\nint main(void) { complexf *a = (complexf *)fft_malloc(sizeof(complexf) * 1024), *b = (complexf *)fft_malloc(sizeof(complexf) * 1024);\n'
                   '  FFT_PLAN_T *p = make_fft_c2c(1024, a, b, 1, 0); fft_execute(p); fft_destroy(p); fft_free(a); fft_free(b); return 0; }\n')
    exe = str(tmp_path / "m")
    r = subprocess.run(["gcc", "-std=gnu99", "-DUSE_FFTW", "-DLIBCSDR_GPL", "-I", ref, "-I", os.path.join(ROOT, "oracle"), str(src), "-L", os.path.dirname(libpath),
                        "-l:libcsdr_amd.so", "-Wl,-rpath," + os.path.dirname(libpath), "-o", exe, "-lm"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    nm = subprocess.run(["nm", "-D", "--defined-only", libpath], capture_output=True, text=True, check=True).stdout
    assert " fftwf_malloc" in nm and " fftwf_free" in nm


def test_soname_build_for_existing_binaries(libpath):
    """`make soname`: the same objects as libcsdr.so.0.15 (the reference's soname, Makefile:56-57), so that an existing binary runs with LD_LIBRARY_PATH alone."""
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "csdr_amd", "csrc"), "soname"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    so = os.path.join(ROOT, "csdr_amd", "libcsdr.so.0.15")
    out = subprocess.run(["readelf", "-d", so], capture_output=True, text=True, check=True).stdout
    assert "libcsdr.so.0.15" in [ln.split("[")[-1].rstrip("]") for ln in out.splitlines() if "SONAME" in ln]


def test_dft16_butterfly():
    """The register-level 16-point butterfly of the three-pass 65536-point transform (csdr_amd/csrc/fft64k.hip), forward and inverse, against numpy."""
    import ctypes as C
    import numpy as np
    import csdr_amd
    L = csdr_amd.lib()
    L.csdr_amd_debug_dft16.restype = None
    L.csdr_amd_debug_dft16.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    rng = np.random.default_rng(16)
    for _ in range(8):
        x = (rng.uniform(-1, 1, 16) + 1j * rng.uniform(-1, 1, 16)).astype(np.complex64)
        y = np.zeros(16, np.complex64)
        L.csdr_amd_debug_dft16(x.ctypes.data, y.ctypes.data, 0)
        assert np.abs(y - np.fft.fft(x.astype(np.complex128))).max() < 2e-6
        L.csdr_amd_debug_dft16(x.ctypes.data, y.ctypes.data, 1)
        assert np.abs(y - 16 * np.fft.ifft(x.astype(np.complex128))).max() < 2e-6


def test_dft8_butterfly():
    """The 8-point butterfly of the channelizer's 512 = 8 x 8 x 8 inverse transforms (csdr_amd/csrc/fft_butterflies.hpp) against numpy."""
    import ctypes as C
    import numpy as np
    import csdr_amd
    L = csdr_amd.lib()
    L.csdr_amd_debug_dft8.restype = None
    L.csdr_amd_debug_dft8.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    rng = np.random.default_rng(8)
    for _ in range(8):
        x = (rng.uniform(-1, 1, 8) + 1j * rng.uniform(-1, 1, 8)).astype(np.complex64)
        y = np.zeros(8, np.complex64)
        L.csdr_amd_debug_dft8(x.ctypes.data, y.ctypes.data, 0)
        assert np.abs(y - np.fft.fft(x.astype(np.complex128))).max() < 1e-6
        L.csdr_amd_debug_dft8(x.ctypes.data, y.ctypes.data, 1)
        assert np.abs(y - 8 * np.fft.ifft(x.astype(np.complex128))).max() < 1e-6


def test_nfm_deemph_digit_planes():
    """The NFM chain's de-emphasis FIR on int8 digit planes (csdr_amd/csrc/nfm.hip): 24-bit fixed-point samples x 23-bit taps with the
    low x low digit pair dropped must reproduce the float64 convolution to ~1e-7 of full scale."""
    import ctypes as C
    import os
    import numpy as np
    import csdr_amd
    L = csdr_amd.lib()
    fn = L.csdr_amd_debug_nfm_deemph_tile
    fn.argtypes = [C.c_int, C.c_float, C.c_void_p, C.c_void_p]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tabs = np.load(os.path.join(root, "tests", "golden", "nfm_deemph_taps.npz"))
    rng = np.random.default_rng(21)
    out = np.zeros(16, np.float32)
    for sr, amp in ((48000, 1.0), (44100, 0.5), (8000, 2.0), (11025, 1.0)):
        taps = tabs["sr%d" % sr].astype(np.float64)
        for trial in range(4):
            x = rng.uniform(-amp, amp, 256).astype(np.float32)
            if trial == 0:
                x[:8] = [amp, -amp, 0.0, amp * 1e-6, -amp * 1e-6, amp / 3, -amp / 3, amp]     # full scale, zero and tiny values
            assert fn(sr, amp, x.ctypes.data, out.ctypes.data) == 0
            want = np.array([np.dot(taps, x[i:i + taps.size].astype(np.float64)) for i in range(16)])
            assert np.abs(out - want).max() < 4e-7 * amp * np.abs(taps).sum(), (sr, np.abs(out - want).max())
    assert fn(12345, 1.0, x.ctypes.data, out.ctypes.data) == -1


@pytest.mark.parametrize("n,ntaps,m", [(4096, 63, 9000), (4096, 1023, 7000), (8192, 1023, 20000), (8192, 2047, 13000), (16384, 4095, 30000), (4096, 1, 4097), (8192, 500, 100),
                                        (-4096, 63, 9000), (-4096, 1023, 7000), (-4096, 1, 4097), (-4096, 500, 100),
                                        (-8192, 2047, 13000), (-8192, 1023, 9000), (-16384, 4095, 30000), (-16384, 3000, 14000)])
def test_fftfilt_lds_stages_on_cpu(n, ntaps, m):
    """The one-pass FFT filter kernel (fftfilt_lds.hip) is built from __host__ __device__ stage functions: the CPU runs the same index algebra (in-place
    decimation-in-frequency stages, taps spectrum in digit-reversed slot order, mirrored inverse stages, overlap-save windows) thread by thread and must
    reproduce the linear convolution bandpass_fir_fft_cc computes (libcsdr.c:814-849).  n = -4096: the wave-per-window form of the 4096-point window (fftfilt_wave.hpp:
    64 lanes x 64 points, rows numbered by fw_pi, the transposes as the index maps the kernel uses, the spectrum in the order its 16-byte loads ask for it);
    n = -8192 / -16384: the team form (fftfilt_team.hpp: 128 / 256 threads x 64 points, the exchanges' address maps, the radix-2 / radix-4 step over neighbouring lanes as its
    two butterfly stages with lane 3's rotation, the spectrum by (register pair, thread) with the k_d slots bit-reversed, both twiddle tables)."""
    import numpy as np
    import csdr_amd
    L = csdr_amd.lib()
    f = L.csdr_amd_debug_fftfilt_lds
    f.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_long, C.c_void_p]; f.restype = C.c_int
    rng = np.random.default_rng(abs(n) + ntaps)
    h = ((rng.standard_normal(ntaps) + 1j * rng.standard_normal(ntaps)) / ntaps).astype(np.complex64)
    x = (rng.standard_normal(m) + 1j * rng.standard_normal(m)).astype(np.complex64)
    y = np.zeros(m, np.complex64)
    assert f(n, h.ctypes.data, ntaps, x.ctypes.data, m, y.ctypes.data) == 0
    want = np.convolve(x.astype(np.complex128), h.astype(np.complex128))[:m]
    assert np.sqrt(np.mean(np.abs(y - want) ** 2) / np.mean(np.abs(want) ** 2)) < 2e-6
    assert f(4096, h.ctypes.data, 5000, x.ctypes.data, m, y.ctypes.data) != 0        # taps that do not fit the window are refused
    assert f(-4096, h.ctypes.data, 5000, x.ctypes.data, m, y.ctypes.data) != 0


def test_ddc_chain_fast_path_is_bit_identical(libpath, port):
    import numpy as np
    """The channelizer's shift state per block (fastddc.c:152-164 -> decimating_shift_addition_cc, libcsdr_gpl.c:131-160) is walked by every rank of a time-sliced bank over
    the WHOLE batch: the kernels take a five-operation fast path when a block changes neither `remain` nor the sample count.  Both forms against the oracle's
    decimating_shift_addition_cc on zero input (the state is data independent), 3000 blocks, every channel rate of config 4 plus awkward ones: same phases, bit for bit."""
    L = C.CDLL(libpath)
    fn = L.csdr_amd_debug_ddc_chain
    fn.argtypes = [C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_float), C.c_void_p, C.POINTER(C.c_int)]
    d0, _ = port.fastddc_init(0.001, 256, 0.0)
    post_in, post_dec, nb = d0.post_input_size, d0.post_decimation, 3000
    assert (post_in, post_dec) == (448, 2)
    rates = [-0.5 + (c + 0.5) / 256 for c in range(0, 256, 17)] + [0.0, 0.123, -0.3711, 0.25, -0.4999, 0.49999]
    zeros = np.zeros(post_in, np.complex64)
    for rate in rates:
        d, _ = port.fastddc_init(0.001, 256, float(rate))
        rate2 = float(d.dsadata.rate)                                # the channel's decimating_shift_addition rate (fastddc.c:60-66)
        got = []
        for mode in (0, 1):
            rem = C.c_int(0); ph = C.c_float(0.0); cnt = C.c_int(0)
            phases = np.zeros(nb, np.float32)
            rc = fn(mode, rate2, post_in, post_dec, nb, C.byref(rem), C.byref(ph), phases.ctypes.data, C.byref(cnt))
            assert rc == 0, (rate, mode)
            got.append((phases, rem.value, np.float32(ph.value), cnt.value))
        assert np.array_equal(got[0][0].view(np.uint32), got[1][0].view(np.uint32)) and got[0][1:] == got[1][1:], rate
        # the oracle: decimating_shift_addition_cc block by block (libcsdr_gpl.c:131-160), state carried
        st = (0, 0.0, 0); total = 0
        for b in range(400):
            assert np.float32(st[1]).view(np.uint32) == got[0][0][b].view(np.uint32), (rate, b)
            _, st = port.decimating_shift_addition_cc(zeros, float(d.post_shift), post_dec, st)
            total += st[2]
        assert total == 400 * (post_in // post_dec)
    # a geometry where the fast path must refuse (post_in not a multiple of post_dec): the general step still runs
    rem = C.c_int(0); ph = C.c_float(0.0); cnt = C.c_int(0); phases = np.zeros(10, np.float32)
    assert fn(1, 0.01, 449, 2, 10, C.byref(rem), C.byref(ph), phases.ctypes.data, C.byref(cnt)) == -1
    assert fn(0, 0.01, 449, 2, 10, C.byref(rem), C.byref(ph), phases.ctypes.data, C.byref(cnt)) == 0 and cnt.value in (2245, 2246)


def test_phase_chain_plan_is_the_reference_loop(libpath, port):
    """seeds.hpp: the phase bookkeeping of `csdr shift_addition_cc` (libcsdr_gpl.c:48-51: ph += rate*PI*1024; while (ph > PI) ph -= 2 PI; while (ph < -PI) ph += 2 PI,
    every step rounded to float) without the loops' iterations -- the per-stream chain objects replay it for every 1024-chunk of every stream, one lane per stream.
    The host build of the kernels' plan (csdr_amd_debug_phase_chain) against the loop in numpy float32: chains of 400 chunks at rates across the band (each starts
    from a different phase), incl. the largest ones (410 wraps per chunk) and rates whose steps sit next to a binade edge."""
    import numpy as np
    fn = C.CDLL(libpath).csdr_amd_debug_phase_chain
    fn.restype = None; fn.argtypes = [C.c_float, C.c_float, C.c_int, C.c_void_p]
    pi = np.float32(3.14159265358979323846); two_pi = np.float32(2) * pi

    def loop(x):
        x = np.float32(x)
        while x > pi: x = np.float32(x - two_pi)
        while x < -pi: x = np.float32(x + two_pi)
        return x

    rng = np.random.default_rng(7)
    edge = [e / (2 * np.pi * 1024) * s for e in (16, 32, 64, 256, 512, 1024, 2048) for s in (0.999, 1.0, 1.001, -1.0)]
    rates = [0.4999, -0.4321, 0.25, -0.05, 0.085, 0.3333, -0.11, 0.0123, 0.5, 0.0, -0.5, 1e-4] + edge + list(rng.uniform(-0.5, 0.5, 40))
    n = 400
    out = np.zeros(n, np.float32)
    for i, rate in enumerate(rates):
        rate = np.float32(rate)
        ph = np.float32(rng.uniform(-3.14, 3.14)) if i % 2 else np.float32(0)
        fn(rate, ph, n, out.ctypes.data)
        step = np.float32(np.float32(rate * np.float32(2)) * pi) * np.float32(1024)
        for k in range(n):
            ph = loop(np.float32(ph + step))
            assert out[k].view(np.uint32) == ph.view(np.uint32), (float(rate), k)


def test_phase_chain_plan_on_two_million_values(libpath):
    """The same plan against the loop on 2000 random rates x 1000 chunks (the loop vectorised over the rates in numpy float32; tools/probes/plan_check.c is the
    48 M-value form of this check)."""
    import numpy as np
    fn = C.CDLL(libpath).csdr_amd_debug_phase_chain
    fn.restype = None; fn.argtypes = [C.c_float, C.c_float, C.c_int, C.c_void_p]
    pi = np.float32(3.14159265358979323846); two_pi = np.float32(2) * pi
    rng = np.random.default_rng(11)
    R, n = 2000, 1000
    rates = rng.uniform(-0.5, 0.5, R).astype(np.float32)
    rates[::97] = (np.array([16, 32, 64, 256, 512, 1024, 2048])[np.arange(len(rates[::97])) % 7] / (2 * np.pi * 1024)).astype(np.float32)      # binade edges of the step
    ph0 = np.where(np.arange(R) % 2 == 1, rng.uniform(-3.14, 3.14, R), 0.0).astype(np.float32)
    got = np.zeros((R, n), np.float32)
    row = np.zeros(n, np.float32)
    for i in range(R):
        fn(float(rates[i]), float(ph0[i]), n, row.ctypes.data); got[i] = row
    step = ((rates * np.float32(2)) * pi) * np.float32(1024)
    assert step.dtype == np.float32
    ph = ph0.copy()
    for k in range(n):
        x = (ph + step).astype(np.float32)
        while True:                                                   # libcsdr_gpl.c:50-51, every subtraction rounded to float
            hi = x > pi
            if not hi.any(): break
            x = np.where(hi, (x - two_pi).astype(np.float32), x)
        while True:
            lo = x < -pi
            if not lo.any(): break
            x = np.where(lo, (x + two_pi).astype(np.float32), x)
        ph = x
        bad = np.nonzero(got[:, k].view(np.uint32) != ph.view(np.uint32))[0]
        assert bad.size == 0, (k, float(rates[bad[0]]))


def test_sharded_bank_default_schedule_and_rccl_abi(libpath):
    """csdr_amd_fastddc_bank_create_sharded picks its schedule from the world size (north_star's channel shards up to two GPUs, time slices beyond: DESIGN.md section 6);
    comm.cpp takes every RCCL type and enum value from <rccl/rccl.h> (decltype of the entry points) -- a change of the ABI fails the BUILD, not the first N > 1 run."""
    L = C.CDLL(libpath)
    L.csdr_amd_fastddc_bank_default_shard_mode.argtypes = [C.c_int]
    assert [L.csdr_amd_fastddc_bank_default_shard_mode(w) for w in (1, 2, 3, 4, 8)] == [0, 0, 1, 1, 1]
    src = open(os.path.join(ROOT, "csdr_amd", "csrc", "comm.cpp")).read()
    assert "#include <rccl/rccl.h>" in src and "enum { ncclFloat32" not in src and "typedef int ncclResult_t" not in src
    for sym in ("ncclGetUniqueId", "ncclCommInitRank", "ncclCommDestroy", "ncclGroupStart", "ncclGroupEnd", "ncclSend", "ncclRecv", "ncclAllGather", "ncclBroadcast", "ncclGetErrorString"):
        assert "decltype(&%s)" % sym in src, sym
