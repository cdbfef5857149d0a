#!/bin/bash
# tools/bench_cli.sh -- PCIe- and pipe-inclusive rate of the CLI on ONE stream: u8 IQ from a file through `csdr wfm_chain_u8_s16` (fused) and
# through `csdr chain "<seven commands>"` (unfused kernels, HBM-resident intermediates) to /dev/null.  Two input lengths separate the process
# start-up (HIP initialisation, table build) from the streaming rate: T(N) = t0 + N / rate.
cd $GRAFT_REPO_ROOT
python - <<PY
import numpy as np
rng = np.random.default_rng(1)
blk = rng.integers(0, 256, 2 * 24000000, dtype=np.uint8).tobytes()
for name, reps in (("/tmp/iq_a.u8", 20), ("/tmp/iq_b.u8", 80)):
    with open(name, "wb") as f:
        for _ in range(reps): f.write(blk)
PY
# best of three runs (the first process on a cold box pays the HIP start-up of the whole image: that is not a streaming rate)
run() { local best=1e9; for i in 1 2 3; do local s=$(date +%s.%N); "$@" > /dev/null 2>/dev/null; local e=$(date +%s.%N); best=$(python -c "print(min($best, $e - $s))"); done; echo $best; }
csdr_amd/csdr wfm_chain_u8_s16 -0.085 < /tmp/iq_a.u8 > /dev/null 2>&1    # warm the box
WFM='convert_u8_f | shift_addition_cc -0.085 | fir_decimate_cc 10 0.05 HAMMING | fmdemod_quadri_cf | fractional_decimator_ff 5.5 | deemphasis_wfm_ff 48000 50e-6 | convert_f_s16'
for b in 262144 1048576 4194304; do
  export CSDR_AMD_BLOCK=$b
  ta=$(run sh -c 'csdr_amd/csdr wfm_chain_u8_s16 -0.085 < /tmp/iq_a.u8'); tb=$(run sh -c 'csdr_amd/csdr wfm_chain_u8_s16 -0.085 < /tmp/iq_b.u8')
  python -c "ta, tb = $ta, $tb; r = (1920e6 - 480e6) / (tb - ta); print('wfm_chain_u8_s16 block=$b: %.0f MS/s streaming, start-up %.2f s (480 M samples %.2f s, 1920 M samples %.2f s)' % (r / 1e6, ta - 480e6 / r, ta, tb))"
done
export CSDR_AMD_BLOCK=4194304
ta=$(run sh -c "csdr_amd/csdr chain '$WFM' < /tmp/iq_a.u8"); tb=$(run sh -c "csdr_amd/csdr chain '$WFM' < /tmp/iq_b.u8")
python -c "ta, tb = $ta, $tb; r = (1920e6 - 480e6) / (tb - ta); print('chain of seven unfused commands block=4194304: %.0f MS/s streaming, start-up %.2f s' % (r / 1e6, ta - 480e6 / r))"
ta=$(run sh -c 'cat /tmp/iq_b.u8')
python -c "print('cat of the same file: %.0f MS/s' % (1920e6 / $ta / 1e6))"
rm -f /tmp/iq_a.u8 /tmp/iq_b.u8
