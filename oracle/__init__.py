"""oracle/ -- TEST INFRASTRUCTURE, not product code.

`oracle.port`  : ctypes view of liboracle.so  (our plain-C restatement, oracle/csdr_oracle.c)
`oracle.ref`   : ctypes view of _ref/libcsdr_ref.so (the unmodified reference compiled by oracle/Makefile;
                 present in this container and, as a prebuilt file, on the GPU box)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""
import ctypes as C
import os
import subprocess
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PORT = os.path.join(HERE, "liboracle.so")
LIB_REF = os.path.join(HERE, "_ref", "libcsdr_ref.so")
REF_CLI = os.path.join(HERE, "_ref", "csdr")

c64 = np.complex64
f32 = np.float32


def build(quiet=True):
    """(Re)build liboracle.so and, when /root/reference exists, _ref/."""
    out = subprocess.run(["make", "-C", HERE], capture_output=True, text=True)
    if out.returncode != 0:
        raise RuntimeError("oracle build failed:\n" + out.stdout + out.stderr)
    if not quiet:
        print(out.stdout)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class _CF(C.Structure):
    _fields_ = [("i", C.c_float), ("q", C.c_float)]


class _ShiftAdd(C.Structure):       # libcsdr_gpl.h:26-31
    _fields_ = [("sindelta", C.c_float), ("cosdelta", C.c_float), ("rate", C.c_float)]


class _DsaStatus(C.Structure):      # libcsdr_gpl.h:39-44
    _fields_ = [("decimation_remain", C.c_int), ("starting_phase", C.c_float), ("output_size", C.c_int)]


class _DcBlock(C.Structure):       # libcsdr.h:110-114
    _fields_ = [("last_input", C.c_float), ("last_output", C.c_float)]


class _Adpcm(C.Structure):         # ima_adpcm.h:5-8
    _fields_ = [("index", C.c_int), ("previousValue", C.c_int)]


class _FastDDC(C.Structure):        # fastddc.h:5-24
    _fields_ = [(n, C.c_int) for n in ("pre_decimation", "post_decimation", "taps_length", "taps_min_length",
                                       "overlap_length", "fft_size", "fft_inv_size", "input_size",
                                       "post_input_size")] + \
               [("pre_shift", C.c_float), ("startbin", C.c_int), ("v", C.c_int), ("offsetbin", C.c_int),
                ("post_shift", C.c_float), ("output_scrape", C.c_int), ("scrap", C.c_int), ("dsadata", _ShiftAdd)]

    def as_dict(self):
        d = {n: getattr(self, n) for n, _ in self._fields_ if n != "dsadata"}
        d["dsadata"] = (self.dsadata.sindelta, self.dsadata.cosdelta, self.dsadata.rate)
        return d


WINDOWS = {"BOXCAR": 0, "BLACKMAN": 1, "HAMMING": 2}


def _cf(a):
    a = np.ascontiguousarray(a, dtype=c64)
    return a


# =====================================================================================
class Port:
    """Our C restatement (liboracle.so). numpy in, numpy out; state passed explicitly."""

    def __init__(self):
        if not os.path.exists(LIB_PORT):
            build()
        L = self.L = C.CDLL(LIB_PORT)
        L.orc_shift_math_cc.restype = C.c_float
        L.orc_shift_table_cc.restype = C.c_float
        L.orc_shift_unroll_init.restype = C.c_float
        L.orc_shift_unroll_cc.restype = C.c_float
        L.orc_shift_addfast_init.restype = C.c_float
        L.orc_shift_addfast_cc.restype = C.c_float
        L.orc_shift_addition_init.restype = _ShiftAdd
        L.orc_shift_addition_cc.restype = C.c_float
        L.orc_shift_addition_fc.restype = C.c_float
        L.orc_decimating_shift_addition_init.restype = _ShiftAdd
        L.orc_decimating_shift_addition_cc.restype = _DsaStatus
        L.orc_fmdemod_quadri_cf.restype = _CF
        L.orc_deemphasis_wfm_ff.restype = C.c_float
        L.orc_fastddc_inv_cc.restype = _DsaStatus
        L.orc_stream_shift_addition_cc.restype = C.c_float
        L.orc_stream_fir_decimate_cc.restype = C.c_long
        L.orc_stream_wfm_chain.restype = C.c_long
        L.orc_fmdemod_atan_cf.restype = C.c_float
        L.orc_dcblock_ff.restype = _DcBlock
        L.orc_fastdcblock_ff.restype = C.c_float
        L.orc_agc_ff.restype = C.c_float
        L.orc_encode_ima_adpcm_i16_u8.restype = _Adpcm
        L.orc_decode_ima_adpcm_u8_i16.restype = _Adpcm

    # ---- design
    def firdes_filter_len(self, tbw):
        return self.L.orc_firdes_filter_len(C.c_float(tbw))

    def firdes_lowpass_f(self, length, cutoff, window="HAMMING"):
        t = np.zeros(length, f32)
        self.L.orc_firdes_lowpass_f(_p(t), length, C.c_float(cutoff), WINDOWS[window])
        return t

    def firdes_bandpass_c(self, length, lo, hi, window="HAMMING"):
        t = np.zeros(length, c64)
        self.L.orc_firdes_bandpass_c(_p(t), length, C.c_float(lo), C.c_float(hi), WINDOWS[window])
        return t

    def next_pow2(self, x):
        return self.L.orc_next_pow2(x)

    def log2n(self, x):
        return self.L.orc_log2n(x)

    # ---- converters
    def _conv(self, name, x, in_dt, out_dt, n_out=None, extra=()):
        x = np.ascontiguousarray(x, dtype=in_dt)
        n = x.size
        y = np.zeros(n if n_out is None else n_out, out_dt)
        getattr(self.L, name)(_p(x), _p(y), n if n_out is None or in_dt == f32 else n_out, *extra)
        return y

    def convert_u8_f(self, x): return self._conv("orc_convert_u8_f", x, np.uint8, f32)
    def convert_s8_f(self, x): return self._conv("orc_convert_s8_f", x, np.int8, f32)
    def convert_s16_f(self, x): return self._conv("orc_convert_s16_f", x, np.int16, f32)
    def convert_f_u8(self, x): return self._conv("orc_convert_f_u8", x, f32, np.uint8)
    def convert_f_s8(self, x): return self._conv("orc_convert_f_s8", x, f32, np.int8)
    def convert_f_s16(self, x): return self._conv("orc_convert_f_s16", x, f32, np.int16)

    def convert_f_s24(self, x, bigendian=0):
        x = np.ascontiguousarray(x, f32)
        return self._conv("orc_convert_f_s24", x, f32, np.uint8, n_out=3 * x.size, extra=(int(bigendian),))

    def convert_s24_f(self, x, bigendian=0):
        x = np.ascontiguousarray(x, np.uint8)
        return self._conv("orc_convert_s24_f", x, np.uint8, f32, n_out=x.size // 3, extra=(int(bigendian),))

    # ---- shifters
    def shift_math_cc(self, x, rate, phase=0.0):
        x = _cf(x); y = np.zeros_like(x)
        ph = self.L.orc_shift_math_cc(_p(x), _p(y), x.size, C.c_float(rate), C.c_float(phase))
        return y, ph

    def shift_table_cc(self, x, rate, table_size=65536, phase=0.0):
        x = _cf(x); y = np.zeros_like(x)
        tab = np.zeros(table_size, f32)
        self.L.orc_shift_table_init(_p(tab), table_size)
        ph = self.L.orc_shift_table_cc(_p(x), _p(y), x.size, C.c_float(rate), _p(tab), table_size, C.c_float(phase))
        return y, ph

    def shift_unroll_cc(self, x, rate, size=1024, phase=0.0):
        """CLI framing (csdr.c:821-845): the table holds `size` entries, calls are `size`-sample chunks."""
        x = _cf(x); y = np.zeros_like(x)
        ds = np.zeros(size, f32); dc = np.zeros(size, f32)
        inc = self.L.orc_shift_unroll_init(C.c_float(rate), size, _p(ds), _p(dc))
        for pos in range(0, x.size, size):
            n = min(size, x.size - pos)
            phase = self.L.orc_shift_unroll_cc(_p(x[pos:]), _p(y[pos:]), n, _p(ds), _p(dc), C.c_float(inc), C.c_float(phase))
        return y, phase

    def shift_addfast_cc(self, x, rate, chunk=1024, phase=0.0):
        x = _cf(x); y = np.zeros_like(x)
        ds = np.zeros(4, f32); dc = np.zeros(4, f32)
        inc = self.L.orc_shift_addfast_init(C.c_float(rate), _p(ds), _p(dc))
        for pos in range(0, x.size, chunk):
            n = min(chunk, x.size - pos)
            phase = self.L.orc_shift_addfast_cc(_p(x[pos:]), _p(y[pos:]), n, _p(ds), _p(dc), C.c_float(inc), C.c_float(phase))
        return y, phase

    def shift_addition_cc(self, x, rate, chunk=1024, phase=0.0):
        x = _cf(x); y = np.zeros_like(x)
        ph = self.L.orc_stream_shift_addition_cc(_p(x), _p(y), C.c_long(x.size), C.c_float(rate), C.c_float(phase), chunk)
        return y, ph

    def shift_addition_fc(self, x, rate, chunk=1024, phase=0.0):
        x = np.ascontiguousarray(x, f32); y = np.zeros(x.size, c64)
        d = self.L.orc_shift_addition_init(C.c_float(rate))
        for pos in range(0, x.size, chunk):
            n = min(chunk, x.size - pos)
            phase = self.L.orc_shift_addition_fc(_p(x[pos:]), _p(y[pos:]), n, d, C.c_float(phase))
        return y, phase

    def decimating_shift_addition_cc(self, x, rate, decimation, status=(0, 0.0, 0)):
        x = _cf(x); y = np.zeros(x.size // decimation + 2, c64)
        d = self.L.orc_decimating_shift_addition_init(C.c_float(rate), decimation)
        st = _DsaStatus(*status)
        st = self.L.orc_decimating_shift_addition_cc(_p(x), _p(y), x.size, d, decimation, st)
        return y[:st.output_size].copy(), (st.decimation_remain, st.starting_phase, st.output_size)

    # ---- filters etc.
    def fir_decimate_cc(self, x, decimation, taps):
        x = _cf(x); taps = np.ascontiguousarray(taps, f32)
        y = np.zeros(x.size // decimation + 1, c64)
        n = self.L.orc_stream_fir_decimate_cc(_p(x), _p(y), C.c_long(x.size), decimation, _p(taps), taps.size)
        return y[:n].copy()

    def fmdemod_quadri_cf(self, x, last=(0.0, 0.0)):
        x = _cf(x); y = np.zeros(x.size, f32)
        r = self.L.orc_fmdemod_quadri_cf(_p(x), _p(y), x.size, _CF(*last))
        return y, (r.i, r.q)

    def deemphasis_wfm_ff(self, x, tau, sample_rate, last=0.0):
        x = np.ascontiguousarray(x, f32); y = np.zeros_like(x)
        r = self.L.orc_deemphasis_wfm_ff(_p(x), _p(y), x.size, C.c_float(tau), int(sample_rate), C.c_float(last))
        return y, r

    def deemphasis_nfm_ff(self, x, taps):
        x = np.ascontiguousarray(x, f32); taps = np.ascontiguousarray(taps, f32)
        y = np.zeros(x.size, f32)
        n = self.L.orc_deemphasis_nfm_ff(_p(x), _p(y), x.size, _p(taps), taps.size)
        return y[:max(n, 0)].copy()

    def deemphasis_nfm_ff_cli(self, x, taps, bufsize=1024):
        """Stream model of `csdr deemphasis_nfm_ff` (csdr.c:1068-1087): the loop runs the FIR over its freshly allocated (zero) buffer before it
        reads anything -- `processed` starts at 0, so the first fread is empty -- i.e. the output is the FIR of  bufsize zeros ++ stream
        (verified against the reference binary: tests/test_oracle_vs_ref.py::test_nfm_chain_vs_reference_cli)."""
        return self.deemphasis_nfm_ff(np.concatenate([np.zeros(bufsize, f32), np.ascontiguousarray(x, f32)]), taps)

    def limit_ff(self, x, m=1.0):
        x = np.ascontiguousarray(x, f32); y = np.zeros_like(x)
        self.L.orc_limit_ff(_p(x), _p(y), x.size, C.c_float(m)); return y

    def gain_ff(self, x, g):
        x = np.ascontiguousarray(x, f32); y = np.zeros_like(x)
        self.L.orc_gain_ff(_p(x), _p(y), x.size, C.c_float(g)); return y

    def fastagc_ff(self, x, block=1024, reference=1.0):
        """CLI framing (csdr.c:1377-1406): whole blocks only; output delayed by two blocks."""
        class St(C.Structure):
            _fields_ = [("buffer_1", C.c_void_p), ("buffer_2", C.c_void_p), ("buffer_input", C.c_void_p),
                        ("peak_1", C.c_float), ("peak_2", C.c_float), ("input_size", C.c_int),
                        ("reference", C.c_float), ("last_gain", C.c_float)]
        x = np.ascontiguousarray(x, f32)
        nb = x.size // block
        bufs = [np.zeros(block, f32) for _ in range(3)]
        st = St(_p(bufs[0]), _p(bufs[1]), _p(bufs[2]), 0, 0, block, reference, 0)
        y = np.zeros(nb * block, f32); ob = np.zeros(block, f32)
        for b in range(nb):
            C.memmove(st.buffer_input, _p(x[b * block:]), 4 * block)
            self.L.orc_fastagc_ff(C.byref(st), _p(ob))
            y[b * block:(b + 1) * block] = ob
        return y

    def fractional_decimator_ff(self, x, rate, num_poly_points=12, taps=None, bufsize=None):
        """bufsize=None: one call over the whole array (the library function); else the CLI loop (csdr.c:1511-1524): the function is called on
        bufsize-sample windows and the unprocessed tail is re-presented -- for rates that are not exact in float the positions differ."""
        class D(C.Structure):
            _fields_ = [("where", C.c_float), ("input_processed", C.c_int), ("output_size", C.c_int),
                        ("num_poly_points", C.c_int), ("denom", C.c_float * 64), ("xifirst", C.c_int),
                        ("xilast", C.c_int), ("rate", C.c_float), ("taps", C.c_void_p), ("taps_length", C.c_int)]
        x = np.ascontiguousarray(x, f32)
        d = D()
        if taps is not None:
            taps = np.ascontiguousarray(taps, f32)
            self.L.orc_fractional_decimator_ff_init(C.byref(d), C.c_float(rate), num_poly_points, _p(taps), taps.size)
        else:
            self.L.orc_fractional_decimator_ff_init(C.byref(d), C.c_float(rate), num_poly_points, None, 0)
        if bufsize is None:
            y = np.zeros(int(x.size / rate) + 4, f32)
            self.L.orc_fractional_decimator_ff(_p(x), _p(y), x.size, C.byref(d))
            return y[:d.output_size].copy()
        ob = np.zeros(bufsize, f32); outs = []; base = 0
        while base + bufsize <= x.size:                      # window = the stream from the first unprocessed sample on
            win = np.ascontiguousarray(x[base:base + bufsize])
            self.L.orc_fractional_decimator_ff(_p(win), _p(ob), bufsize, C.byref(d))
            outs.append(ob[:d.output_size].copy())
            if d.input_processed <= 0:
                break
            base += d.input_processed
        return np.concatenate(outs) if outs else np.zeros(0, f32)


    # ---- f2 blocks (AM/SSB chains, waterfall path)
    def amdemod_cf(self, x):
        x = _cf(x); y = np.zeros(x.size, f32); self.L.orc_amdemod_cf(_p(x), _p(y), x.size); return y

    def amdemod_estimator_cf(self, x, alpha=0.0, beta=0.0):
        x = _cf(x); y = np.zeros(x.size, f32)
        self.L.orc_amdemod_estimator_cf(_p(x), _p(y), x.size, C.c_float(alpha), C.c_float(beta)); return y

    def fmdemod_atan_cf(self, x, last_phase=0.0):
        x = _cf(x); y = np.zeros(x.size, f32)
        ph = self.L.orc_fmdemod_atan_cf(_p(x), _p(y), x.size, C.c_float(last_phase)); return y, ph

    def dcblock_ff(self, x, a=0.0, state=(0.0, 0.0)):
        x = np.ascontiguousarray(x, f32); y = np.zeros_like(x)
        st = self.L.orc_dcblock_ff(_p(x), _p(y), x.size, C.c_float(a), _DcBlock(*state)); return y, (st.last_input, st.last_output)

    def fastdcblock_ff(self, x, block=1024, last_dc=0.0):
        """csdr.c:943-960 framing: whole blocks only."""
        x = np.ascontiguousarray(x, f32); nb = x.size // block; y = np.zeros(nb * block, f32)
        for b in range(nb):
            last_dc = self.L.orc_fastdcblock_ff(_p(x[b * block:]), _p(y[b * block:]), block, C.c_float(last_dc))
        return y, last_dc

    def agc_ff(self, x, block=1024, hang_time=200, reference=0.2, attack_rate=0.01, decay_rate=0.0001, max_gain=65536.0,
               attack_wait=0, filter_alpha=0.999, last_gain=1.0):
        """csdr.c:1338-1375 framing: one call per the_bufsize block, counters restart, last_gain carried."""
        x = np.ascontiguousarray(x, f32); y = np.zeros_like(x)
        for at in range(0, x.size, block):
            n = min(block, x.size - at)
            last_gain = self.L.orc_agc_ff(_p(x[at:]), _p(y[at:]), n, C.c_float(reference), C.c_float(attack_rate), C.c_float(decay_rate),
                                          C.c_float(max_gain), C.c_short(hang_time), C.c_short(attack_wait), C.c_float(filter_alpha), C.c_float(last_gain))
        return y, last_gain

    def realpart_cf(self, x):
        x = _cf(x); y = np.zeros(x.size, f32); self.L.orc_realpart_cf(_p(x), _p(y), x.size); return y

    def logpower_cf(self, x, add_db=0.0):
        x = _cf(x); y = np.zeros(x.size, f32); self.L.orc_logpower_cf(_p(x), _p(y), x.size, C.c_float(add_db)); return y

    def precalculate_window(self, size, window="HAMMING"):
        w = np.zeros(size, f32); self.L.orc_precalculate_window(_p(w), size, WINDOWS[window]); return w

    def fft_cc(self, x, fft_size, every_n, window="HAMMING"):
        """csdr.c:1569-1641 stream model (clean EOF): windowed FFT of the last fft_size samples every `every_n` new samples."""
        x = _cf(x); w = self.precalculate_window(fft_size, window)
        buf = np.zeros(fft_size, c64); out = []
        pos = 0
        while True:
            if every_n > fft_size:
                if pos + every_n > x.size: break
                buf[:] = x[pos:pos + fft_size]; pos += every_n
            else:
                if pos + every_n > x.size: break
                buf[:fft_size - every_n] = buf[every_n:].copy(); buf[fft_size - every_n:] = x[pos:pos + every_n]; pos += every_n
            win = np.zeros(fft_size, c64)
            self.L.orc_apply_precalculated_window_c(_p(buf), _p(win), fft_size, _p(w))
            out.append(self.fft_c2c(win, True))
        return np.concatenate(out) if out else np.zeros(0, c64)


    # ---- f3: IMA ADPCM
    def encode_ima_adpcm_i16_u8(self, x, state=(0, 0)):
        x = np.ascontiguousarray(x, np.int16); y = np.zeros(x.size // 2, np.uint8)
        st = self.L.orc_encode_ima_adpcm_i16_u8(_p(x), _p(y), x.size, _Adpcm(*state)); return y, (st.index, st.previousValue)

    def decode_ima_adpcm_u8_i16(self, x, state=(0, 0)):
        x = np.ascontiguousarray(x, np.uint8); y = np.zeros(2 * x.size, np.int16)
        st = self.L.orc_decode_ima_adpcm_u8_i16(_p(x), _p(y), x.size, _Adpcm(*state)); return y, (st.index, st.previousValue)

    def compress_fft_adpcm_f_u8(self, x, fft_size):
        """csdr.c:1745-1768 stream model: whole blocks of fft_size floats -> (fft_size+10)/2 bytes each."""
        x = np.ascontiguousarray(x, f32); nb = x.size // fft_size; ob = (fft_size + 10) // 2
        y = np.zeros(nb * ob, np.uint8)
        for b in range(nb):
            self.L.orc_compress_fft_adpcm_f_u8(_p(x[b * fft_size:]), _p(y[b * ob:]), fft_size)
        return y

    # ---- FFT paths
    def fft_c2c(self, x, forward=True):
        x = _cf(x); y = np.zeros_like(x)
        self.L.orc_fft_c2c(_p(x), _p(y), x.size, int(forward)); return y

    def bandpass_fir_fft_cc(self, x, taps, fft_size):
        """csdr.c:1832-1880 stream model at a given fft_size: whole blocks of input_size only."""
        x = _cf(x); taps = _cf(taps)
        L = taps.size; inp = fft_size - L + 1; ovl = L - 1
        tp = np.zeros(fft_size, c64); tp[:L] = taps
        tf = self.fft_c2c(tp, True)
        nb = x.size // inp
        y = np.zeros(nb * inp, c64)
        prev = np.zeros(fft_size, c64); cur = np.zeros(fft_size, c64); buf = np.zeros(fft_size, c64)
        for b in range(nb):
            buf[:inp] = x[b * inp:(b + 1) * inp]
            last = np.ascontiguousarray(prev[inp:inp + ovl])
            self.L.orc_apply_fir_fft_cc(_p(buf), _p(cur), fft_size, _p(tf), _p(last), ovl)
            y[b * inp:(b + 1) * inp] = cur[:inp]
            prev, cur = cur, prev
        return y

    def fastddc_init(self, tbw, decimation, shift_rate):
        d = _FastDDC()
        err = self.L.orc_fastddc_init(C.byref(d), C.c_float(tbw), decimation, C.c_float(shift_rate))
        return d, err

    def fastddc_fwd_cc(self, x, ddc):
        """csdr.c:2289-2299: overlap-save framing + forward FFT; returns [n_blocks, fft_size] spectra."""
        x = _cf(x)
        nb = x.size // ddc.input_size
        buf = np.zeros(ddc.fft_size, c64)
        out = np.zeros((nb, ddc.fft_size), c64)
        for b in range(nb):
            buf[:ddc.overlap_length] = buf[ddc.input_size:ddc.input_size + ddc.overlap_length].copy()
            buf[ddc.overlap_length:] = x[b * ddc.input_size:(b + 1) * ddc.input_size]
            out[b] = self.fft_c2c(buf, True)
        return out

    def fastddc_taps_fft(self, ddc, shift_rate, decimation, window="HAMMING"):
        """csdr.c:2338-2351"""
        half = np.float32(0.5) / np.float32(decimation)   # float filter_half_bw = 0.5/decimation
        lo = np.float32(-np.float32(shift_rate)) - half
        hi = np.float32(-np.float32(shift_rate)) + half
        taps = self.firdes_bandpass_c(ddc.taps_length, float(lo), float(hi), window)
        tp = np.zeros(ddc.fft_size, c64); tp[:ddc.taps_length] = taps
        tf = self.fft_c2c(tp, True)
        self.L.orc_fft_swap_sides(_p(tf), ddc.fft_size)
        return tf

    def fastddc_inv_cc(self, spectra, ddc, taps_fft, status=None):
        """status = (decimation_remain, starting_phase) to continue a stream; with a status the call returns (samples, status after the last block)."""
        spectra = np.ascontiguousarray(spectra, c64); taps_fft = _cf(taps_fft)
        st = _DsaStatus(0, 0.0, 0) if status is None else _DsaStatus(int(status[0]), float(status[1]), 0)
        outs = []
        ob = np.zeros(ddc.post_input_size + 2, c64)
        for b in range(spectra.shape[0]):
            st = self.L.orc_fastddc_inv_cc(_p(spectra[b]), _p(ob), C.byref(ddc), _p(taps_fft), st)
            outs.append(ob[:st.output_size].copy())
        y = np.concatenate(outs) if outs else np.zeros(0, c64)
        return y if status is None else (y, (st.decimation_remain, st.starting_phase))

    # ---- chains
    def wfm_chain(self, iq_u8, shift_rate, decimation, taps, frac_rate=5, tau=50e-6, audio_rate=48000):
        iq_u8 = np.ascontiguousarray(iq_u8, np.uint8); taps = np.ascontiguousarray(taps, f32)
        n = iq_u8.size // 2
        na_max = n // (decimation * frac_rate) + 2
        s16 = np.zeros(na_max, np.int16); af = np.zeros(na_max, f32)
        na = self.L.orc_stream_wfm_chain(_p(iq_u8), C.c_long(n), C.c_float(shift_rate), decimation, _p(taps), taps.size,
                                         frac_rate, C.c_float(tau), audio_rate, _p(s16), _p(af))
        return s16[:na].copy(), af[:na].copy()


    def nfm_chain(self, iq_u8, shift_rate, nfm_taps, decimation=50, tbw=0.005, agc_block=1024):
        """README.md:87 stage by stage on one stream (stream models of the CLI loops)."""
        xf = self.convert_u8_f(iq_u8).view(c64)
        sh, _ = self.shift_addition_cc(xf, shift_rate)
        nt = self.firdes_filter_len(tbw)
        dec = self.fir_decimate_cc(sh, decimation, self.firdes_lowpass_f(nt, 0.5 / decimation))
        dem, _ = self.fmdemod_quadri_cf(dec)
        de = self.deemphasis_nfm_ff_cli(self.limit_ff(dem, 1.0), nfm_taps)
        agc = self.fastagc_ff(de, agc_block, 1.0)
        return self.convert_f_s16(agc), agc


# =====================================================================================
class Ref:
    """The unmodified reference, compiled into _ref/libcsdr_ref.so.  Signatures: libcsdr.h:85-229,
    libcsdr_gpl.h:26-46, fastddc.h:26-29, fft_fftw.h:24-27."""

    class FftPlan(C.Structure):     # fft_fftw.h:14-20
        _fields_ = [("size", C.c_int), ("input", C.c_void_p), ("output", C.c_void_p), ("plan", C.c_void_p)]

    class FastAgc(C.Structure):     # libcsdr.h:118-128
        _fields_ = [("buffer_1", C.c_void_p), ("buffer_2", C.c_void_p), ("buffer_input", C.c_void_p),
                    ("peak_1", C.c_float), ("peak_2", C.c_float), ("input_size", C.c_int),
                    ("reference", C.c_float), ("last_gain", C.c_float)]

    class FracDec(C.Structure):     # libcsdr.h:151-168
        _fields_ = [("where", C.c_float), ("input_processed", C.c_int), ("output_size", C.c_int),
                    ("num_poly_points", C.c_int), ("poly_precalc_denomiator", C.c_void_p),
                    ("coeffs_buf", C.c_void_p), ("filtered_buf", C.c_void_p), ("xifirst", C.c_int),
                    ("xilast", C.c_int), ("rate", C.c_float), ("taps", C.c_void_p), ("taps_length", C.c_int)]

    class ShiftTable(C.Structure):  # libcsdr.h:180-184
        _fields_ = [("table", C.c_void_p), ("table_size", C.c_int)]

    class ShiftAddfast(C.Structure):  # libcsdr.h:189-194
        _fields_ = [("dsin", C.c_float * 4), ("dcos", C.c_float * 4), ("phase_increment", C.c_float)]

    class ShiftUnroll(C.Structure):   # libcsdr.h:199-205
        _fields_ = [("dsin", C.c_void_p), ("dcos", C.c_void_p), ("phase_increment", C.c_float), ("size", C.c_int)]

    @staticmethod
    def available():
        return os.path.exists(LIB_REF)

    def __init__(self, lib_path=None):
        """lib_path=None: the compiled reference.  Any other library that exports the reference's symbols with the reference's
        signatures can be driven through this same harness (tests/test_compat_gpu.py passes libcsdr_amd.so)."""
        L = self.L = C.CDLL(lib_path or LIB_REF)
        for name in ("shift_math_cc", "shift_table_cc", "shift_unroll_cc", "shift_addfast_cc", "shift_addition_cc",
                     "shift_addition_fc", "deemphasis_wfm_ff"):
            getattr(L, name).restype = C.c_float
        L.shift_addition_init.restype = _ShiftAdd
        L.decimating_shift_addition_init.restype = _ShiftAdd
        L.decimating_shift_addition_cc.restype = _DsaStatus
        L.fmdemod_quadri_cf.restype = _CF
        L.fmdemod_quadri_novect_cf.restype = _CF
        L.fastddc_inv_cc.restype = _DsaStatus
        L.shift_table_init.restype = Ref.ShiftTable
        L.shift_addfast_init.restype = Ref.ShiftAddfast
        L.shift_unroll_init.restype = Ref.ShiftUnroll
        L.fractional_decimator_ff_init.restype = Ref.FracDec
        L.make_fft_c2c.restype = C.POINTER(Ref.FftPlan)
        for name in ("fmdemod_atan_cf", "fastdcblock_ff", "agc_ff"):
            getattr(L, name).restype = C.c_float
        L.dcblock_ff.restype = _DcBlock
        L.encode_ima_adpcm_i16_u8.restype = _Adpcm
        L.decode_ima_adpcm_u8_i16.restype = _Adpcm
        L.precalculate_window.restype = C.POINTER(C.c_float)
        if lib_path is None:
            L.fftwf_malloc.restype = C.c_void_p


    # ---- f2 blocks
    def amdemod_cf(self, x):
        x = _cf(x); y = np.zeros(x.size, f32); self.L.amdemod_cf(_p(x), _p(y), x.size); return y

    def amdemod_estimator_cf(self, x, alpha=0.0, beta=0.0):
        x = _cf(x); y = np.zeros(x.size, f32)
        self.L.amdemod_estimator_cf(_p(x), _p(y), x.size, C.c_float(alpha), C.c_float(beta)); return y

    def fmdemod_atan_cf(self, x, last_phase=0.0):
        x = _cf(x); y = np.zeros(x.size, f32)
        ph = self.L.fmdemod_atan_cf(_p(x), _p(y), x.size, C.c_float(last_phase)); return y, ph

    def dcblock_ff(self, x, a=0.0, state=(0.0, 0.0)):
        x = np.ascontiguousarray(x, f32); y = np.zeros_like(x)
        st = self.L.dcblock_ff(_p(x), _p(y), x.size, C.c_float(a), _DcBlock(*state)); return y, (st.last_input, st.last_output)

    def fastdcblock_ff(self, x, block=1024, last_dc=0.0):
        x = np.ascontiguousarray(x, f32); nb = x.size // block; y = np.zeros(nb * block, f32)
        for b in range(nb):
            last_dc = self.L.fastdcblock_ff(_p(x[b * block:]), _p(y[b * block:]), block, C.c_float(last_dc))
        return y, last_dc

    def agc_ff(self, x, block=1024, hang_time=200, reference=0.2, attack_rate=0.01, decay_rate=0.0001, max_gain=65536.0,
               attack_wait=0, filter_alpha=0.999, last_gain=1.0):
        x = np.ascontiguousarray(x, f32); y = np.zeros_like(x)
        for at in range(0, x.size, block):
            n = min(block, x.size - at)
            last_gain = self.L.agc_ff(_p(x[at:]), _p(y[at:]), n, C.c_float(reference), C.c_float(attack_rate), C.c_float(decay_rate),
                                      C.c_float(max_gain), C.c_short(hang_time), C.c_short(attack_wait), C.c_float(filter_alpha), C.c_float(last_gain))
        return y, last_gain

    def logpower_cf(self, x, add_db=0.0):
        x = _cf(x); y = np.zeros(x.size, f32); self.L.logpower_cf(_p(x), _p(y), x.size, C.c_float(add_db)); return y

    def precalculate_window(self, size, window="HAMMING"):
        p = self.L.precalculate_window(size, WINDOWS[window])
        return np.ctypeslib.as_array(p, (size,)).copy()

    def apply_precalculated_window_c(self, x, w):
        x = _cf(x); w = np.ascontiguousarray(w, f32); y = np.zeros_like(x)
        self.L.apply_precalculated_window_c(_p(x), _p(y), x.size, _p(w)); return y


    # ---- f3
    def encode_ima_adpcm_i16_u8(self, x, state=(0, 0)):
        x = np.ascontiguousarray(x, np.int16); y = np.zeros(x.size // 2, np.uint8)
        st = self.L.encode_ima_adpcm_i16_u8(_p(x), _p(y), x.size, _Adpcm(*state)); return y, (st.index, st.previousValue)

    def decode_ima_adpcm_u8_i16(self, x, state=(0, 0)):
        x = np.ascontiguousarray(x, np.uint8); y = np.zeros(2 * x.size, np.int16)
        st = self.L.decode_ima_adpcm_u8_i16(_p(x), _p(y), x.size, _Adpcm(*state)); return y, (st.index, st.previousValue)

    def nfm_taps(self, sample_rate):
        n = {48000: 201, 44100: 123, 8000: 81, 11025: 81}[sample_rate]
        arr = (C.c_float * n).in_dll(self.L, "deemphasis_nfm_predefined_fir_%d" % sample_rate)
        return np.array(arr, dtype=f32)

    # ---- design
    def firdes_filter_len(self, tbw): return self.L.firdes_filter_len(C.c_float(tbw))

    def firdes_lowpass_f(self, length, cutoff, window="HAMMING"):
        t = np.zeros(length, f32)
        self.L.firdes_lowpass_f(_p(t), length, C.c_float(cutoff), WINDOWS[window]); return t

    def firdes_bandpass_c(self, length, lo, hi, window="HAMMING"):
        t = np.zeros(length, c64)
        self.L.firdes_bandpass_c(_p(t), length, C.c_float(lo), C.c_float(hi), WINDOWS[window]); return t

    def next_pow2(self, x): return self.L.next_pow2(x)
    def log2n(self, x): return self.L.log2n(x)

    # ---- converters
    def _conv(self, name, x, in_dt, out_dt, n_call=None, n_out=None, extra=()):
        x = np.ascontiguousarray(x, dtype=in_dt)
        y = np.zeros(x.size if n_out is None else n_out, out_dt)
        getattr(self.L, name)(_p(x), _p(y), x.size if n_call is None else n_call, *extra)
        return y

    def convert_u8_f(self, x): return self._conv("convert_u8_f", x, np.uint8, f32)
    def convert_s8_f(self, x): return self._conv("convert_s8_f", x, np.int8, f32)
    def convert_s16_f(self, x): return self._conv("convert_s16_f", x, np.int16, f32)
    def convert_f_u8(self, x): return self._conv("convert_f_u8", x, f32, np.uint8)
    def convert_f_s8(self, x): return self._conv("convert_f_s8", x, f32, np.int8)
    def convert_f_s16(self, x): return self._conv("convert_f_s16", x, f32, np.int16)

    def convert_f_s24(self, x, bigendian=0):
        x = np.ascontiguousarray(x, f32)
        return self._conv("convert_f_s24", x, f32, np.uint8, n_out=3 * x.size, extra=(int(bigendian),))

    def convert_s24_f(self, x, bigendian=0):
        x = np.ascontiguousarray(x, np.uint8)
        return self._conv("convert_s24_f", x, np.uint8, f32, n_call=x.size // 3, n_out=x.size // 3, extra=(int(bigendian),))

    # ---- shifters (with the CLI's chunking)
    def shift_math_cc(self, x, rate, phase=0.0):
        x = _cf(x); y = np.zeros_like(x)
        ph = self.L.shift_math_cc(_p(x), _p(y), x.size, C.c_float(rate), C.c_float(phase)); return y, ph

    def shift_table_cc(self, x, rate, table_size=65536, phase=0.0):
        x = _cf(x); y = np.zeros_like(x)
        td = self.L.shift_table_init(table_size)
        ph = self.L.shift_table_cc(_p(x), _p(y), x.size, C.c_float(rate), td, C.c_float(phase))
        self.L.shift_table_deinit(td)
        return y, ph

    def shift_unroll_cc(self, x, rate, size=1024, phase=0.0):
        x = _cf(x); y = np.zeros_like(x)
        d = self.L.shift_unroll_init(C.c_float(rate), size)
        for pos in range(0, x.size, size):
            n = min(size, x.size - pos)
            phase = self.L.shift_unroll_cc(_p(x[pos:]), _p(y[pos:]), n, C.byref(d), C.c_float(phase))
        return y, phase

    def shift_addfast_cc(self, x, rate, chunk=1024, phase=0.0):
        x = _cf(x); y = np.zeros_like(x)
        d = self.L.shift_addfast_init(C.c_float(rate))
        for pos in range(0, x.size, chunk):
            n = min(chunk, x.size - pos)
            phase = self.L.shift_addfast_cc(_p(x[pos:]), _p(y[pos:]), n, C.byref(d), C.c_float(phase))
        return y, phase

    def shift_addition_cc(self, x, rate, chunk=1024, phase=0.0):
        x = _cf(x); y = np.zeros_like(x)
        d = self.L.shift_addition_init(C.c_float(rate))
        for pos in range(0, x.size, chunk):
            n = min(chunk, x.size - pos)
            phase = self.L.shift_addition_cc(_p(x[pos:]), _p(y[pos:]), n, d, C.c_float(phase))
        return y, phase

    def shift_addition_fc(self, x, rate, chunk=1024, phase=0.0):
        x = np.ascontiguousarray(x, f32); y = np.zeros(x.size, c64)
        d = self.L.shift_addition_init(C.c_float(rate))
        for pos in range(0, x.size, chunk):
            n = min(chunk, x.size - pos)
            phase = self.L.shift_addition_fc(_p(x[pos:]), _p(y[pos:]), n, d, C.c_float(phase))
        return y, phase

    def decimating_shift_addition_cc(self, x, rate, decimation, status=(0, 0.0, 0)):
        x = _cf(x); y = np.zeros(x.size // decimation + 2, c64)
        d = self.L.decimating_shift_addition_init(C.c_float(rate), decimation)
        st = self.L.decimating_shift_addition_cc(_p(x), _p(y), x.size, d, decimation, _DsaStatus(*status))
        return y[:st.output_size].copy(), (st.decimation_remain, st.starting_phase, st.output_size)

    # ---- filters etc.
    def fir_decimate_cc_block(self, x, decimation, taps):
        """One library call (libcsdr.c:528-549)."""
        x = _cf(x); taps = np.ascontiguousarray(taps, f32)
        y = np.zeros(x.size // decimation + 1, c64)
        n = self.L.fir_decimate_cc(_p(x), _p(y), x.size, decimation, _p(taps), taps.size)
        return y[:n].copy()

    def fir_decimate_cc(self, x, decimation, taps, bufsize=16384):
        """CLI block loop with the refeed rule (csdr.c:1160-1176), whole stream in memory."""
        x = _cf(x); taps = np.ascontiguousarray(taps, f32)
        while bufsize < 2 * taps.size:
            bufsize *= 2
        outs = []; pos = 0
        ob = np.zeros(bufsize // decimation + 1, c64)
        while pos + bufsize <= x.size:
            n = self.L.fir_decimate_cc(_p(x[pos:]), _p(ob), bufsize, decimation, _p(taps), taps.size)
            outs.append(ob[:n].copy()); pos += n * decimation
        rem = x.size - pos                       # model of an infinitely patient reader: final partial window
        if rem >= taps.size:
            n = self.L.fir_decimate_cc(_p(x[pos:]), _p(ob), rem, decimation, _p(taps), taps.size)
            outs.append(ob[:n].copy())
        return np.concatenate(outs) if outs else np.zeros(0, c64)

    def fmdemod_quadri_cf(self, x, last=(0.0, 0.0), block=1024):
        x = _cf(x); y = np.zeros(x.size, f32)
        tmp = np.zeros(4 * max(block, 1), f32); last = _CF(*last)
        for pos in range(0, x.size, block):
            n = min(block, x.size - pos)
            last = self.L.fmdemod_quadri_cf(_p(x[pos:]), _p(y[pos:]), n, _p(tmp), last)
        return y, (last.i, last.q)

    def fmdemod_quadri_novect_cf(self, x, last=(0.0, 0.0)):
        x = _cf(x); y = np.zeros(x.size, f32)
        r = self.L.fmdemod_quadri_novect_cf(_p(x), _p(y), x.size, _CF(*last))
        return y, (r.i, r.q)

    def deemphasis_wfm_ff(self, x, tau, sample_rate, last=0.0, block=1024):
        x = np.ascontiguousarray(x, f32); y = np.zeros_like(x)
        for pos in range(0, x.size, block):
            n = min(block, x.size - pos)
            last = self.L.deemphasis_wfm_ff(_p(x[pos:]), _p(y[pos:]), n, C.c_float(tau), int(sample_rate), C.c_float(last))
        return y, last

    def deemphasis_nfm_ff(self, x, sample_rate):
        x = np.ascontiguousarray(x, f32); y = np.zeros(x.size, f32)
        n = self.L.deemphasis_nfm_ff(_p(x), _p(y), x.size, int(sample_rate))
        return y[:max(n, 0)].copy()

    def limit_ff(self, x, m=1.0):
        x = np.ascontiguousarray(x, f32); y = np.zeros_like(x)
        self.L.limit_ff(_p(x), _p(y), x.size, C.c_float(m)); return y

    def gain_ff(self, x, g):
        x = np.ascontiguousarray(x, f32); y = np.zeros_like(x)
        self.L.gain_ff(_p(x), _p(y), x.size, C.c_float(g)); return y

    def fastagc_ff(self, x, block=1024, reference=1.0):
        x = np.ascontiguousarray(x, f32)
        nb = x.size // block
        bufs = [np.zeros(block, f32) for _ in range(3)]
        st = Ref.FastAgc(_p(bufs[0]), _p(bufs[1]), _p(bufs[2]), 0, 0, block, reference, 0)
        y = np.zeros(nb * block, f32); ob = np.zeros(block, f32)
        for b in range(nb):
            C.memmove(st.buffer_input, _p(x[b * block:]), 4 * block)
            self.L.fastagc_ff(C.byref(st), _p(ob))
            y[b * block:(b + 1) * block] = ob
        return y

    def fractional_decimator_ff(self, x, rate, num_poly_points=12, taps=None, bufsize=None):
        """bufsize=None: one call over the whole array; else the CLI loop (csdr.c:1511-1524)."""
        x = np.ascontiguousarray(x, f32)
        if taps is not None:
            taps = np.ascontiguousarray(taps, f32)
            d = self.L.fractional_decimator_ff_init(C.c_float(rate), num_poly_points, _p(taps), taps.size)
        else:
            d = self.L.fractional_decimator_ff_init(C.c_float(rate), num_poly_points, None, 0)
        if bufsize is None:
            y = np.zeros(int(x.size / rate) + 4, f32)
            self.L.fractional_decimator_ff(_p(x), _p(y), x.size, C.byref(d))
            return y[:d.output_size].copy()
        buf = np.zeros(bufsize, f32); ob = np.zeros(bufsize, f32); outs = []; pos = 0
        while True:
            need = bufsize if d.input_processed == 0 else d.input_processed
            if pos + need > x.size:
                break
            if d.input_processed == 0:
                d.input_processed = bufsize
            else:
                buf[:bufsize - d.input_processed] = buf[d.input_processed:].copy()
            buf[bufsize - d.input_processed:] = x[pos:pos + d.input_processed]; pos += d.input_processed
            self.L.fractional_decimator_ff(_p(buf), _p(ob), bufsize, C.byref(d))
            outs.append(ob[:d.output_size].copy())
        return np.concatenate(outs) if outs else np.zeros(0, f32)

    # ---- FFT paths
    def fft_c2c(self, x, forward=True):
        x = _cf(x); y = np.zeros_like(x)
        p = self.L.make_fft_c2c(x.size, _p(x), _p(y), int(forward), 0)
        self.L.fft_execute(p); self.L.fft_destroy(p)
        return y

    def bandpass_fir_fft_cc(self, x, taps, fft_size):
        """apply_fir_fft_cc (libcsdr.c:814-849) driven like csdr.c:1846-1880 at a chosen fft_size."""
        x = _cf(x); taps = _cf(taps)
        Lt = taps.size; inp = fft_size - Lt + 1; ovl = Lt - 1
        tp = np.zeros(fft_size, c64); tp[:Lt] = taps
        tf = self.fft_c2c(tp, True)
        a_in = np.zeros(fft_size, c64); a_spec = np.zeros(fft_size, c64); a_prod = np.zeros(fft_size, c64)
        o = [np.zeros(fft_size, c64), np.zeros(fft_size, c64)]
        pf = self.L.make_fft_c2c(fft_size, _p(a_in), _p(a_spec), 1, 0)
        pi = [self.L.make_fft_c2c(fft_size, _p(a_prod), _p(o[0]), 0, 0), self.L.make_fft_c2c(fft_size, _p(a_prod), _p(o[1]), 0, 0)]
        nb = x.size // inp
        y = np.zeros(nb * inp, c64)
        for b in range(nb):
            a_in[:inp] = x[b * inp:(b + 1) * inp]
            cur, oth = b & 1, (b & 1) ^ 1
            last = o[oth][inp:]
            self.L.apply_fir_fft_cc(pf, pi[cur], _p(tf), _p(last), ovl)
            y[b * inp:(b + 1) * inp] = o[cur][:inp]
        self.L.fft_destroy(pf); self.L.fft_destroy(pi[0]); self.L.fft_destroy(pi[1])
        return y

    def fastddc_init(self, tbw, decimation, shift_rate):
        d = _FastDDC()
        err = self.L.fastddc_init(C.byref(d), C.c_float(tbw), decimation, C.c_float(shift_rate))
        return d, err

    def fastddc_fwd_cc(self, x, ddc):
        x = _cf(x)
        nb = x.size // ddc.input_size
        buf = np.zeros(ddc.fft_size, c64)
        out = np.zeros((nb, ddc.fft_size), c64)
        for b in range(nb):
            buf[:ddc.overlap_length] = buf[ddc.input_size:ddc.input_size + ddc.overlap_length].copy()
            buf[ddc.overlap_length:] = x[b * ddc.input_size:(b + 1) * ddc.input_size]
            out[b] = self.fft_c2c(buf, True)
        return out

    def fastddc_taps_fft(self, ddc, shift_rate, decimation, window="HAMMING"):
        half = np.float32(0.5) / np.float32(decimation)
        lo = np.float32(-np.float32(shift_rate)) - half
        hi = np.float32(-np.float32(shift_rate)) + half
        taps = self.firdes_bandpass_c(ddc.taps_length, float(lo), float(hi), window)
        tp = np.zeros(ddc.fft_size, c64); tp[:ddc.taps_length] = taps
        tf = self.fft_c2c(tp, True)
        self.L.fft_swap_sides(_p(tf), ddc.fft_size)
        return tf

    def fastddc_inv_cc(self, spectra, ddc, taps_fft):
        spectra = np.ascontiguousarray(spectra, c64); taps_fft = _cf(taps_fft)
        M = ddc.fft_inv_size
        a_in = np.zeros(M, c64); a_out = np.zeros(M, c64)
        pinv = self.L.make_fft_c2c(M, _p(a_in), _p(a_out), 0, 0)
        st = _DsaStatus(0, 0.0, 0)
        ob = np.zeros(ddc.post_input_size + 2, c64); outs = []
        for b in range(spectra.shape[0]):
            spec = spectra[b].copy()          # the reference swaps its input in place (fastddc.c:123)
            st = self.L.fastddc_inv_cc(_p(spec), _p(ob), C.byref(ddc), pinv, _p(taps_fft), st)
            outs.append(ob[:st.output_size].copy())
        self.L.fft_destroy(pinv)
        return np.concatenate(outs) if outs else np.zeros(0, c64)

    # ---- chain, stage by stage with the CLI's block sizes
    def wfm_chain(self, iq_u8, shift_rate, decimation, taps, frac_rate=5, tau=50e-6, audio_rate=48000):
        xf = self.convert_u8_f(iq_u8).view(c64)
        sh, _ = self.shift_addition_cc(xf, shift_rate)
        dec = self.fir_decimate_cc(sh, decimation, taps)
        dem, _ = self.fmdemod_quadri_cf(dec)
        aud = self.fractional_decimator_ff(dem, float(frac_rate))
        de, _ = self.deemphasis_wfm_ff(aud, tau, audio_rate)
        return self.convert_f_s16(de), de


_port = None
_ref = None


def port():
    global _port
    if _port is None:
        _port = Port()
    return _port


def ref():
    global _ref
    if _ref is None:
        _ref = Ref()
    return _ref


def relrms(a, b):
    """|a-b|_2 / |b|_2 over whole arrays (SURVEY.md section 8d parity gate)."""
    a = np.asarray(a); b = np.asarray(b)
    den = np.linalg.norm(b.astype(np.complex128 if np.iscomplexobj(b) else np.float64))
    num = np.linalg.norm(a.astype(np.complex128 if np.iscomplexobj(a) else np.float64) - b)
    return float(num / den) if den > 0 else float(num)
