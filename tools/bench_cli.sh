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
${CSDR_BIN:-csdr_amd/csdr} wfm_chain_u8_s16 -0.085 < /tmp/iq_a.u8 > /dev/null 2>&1    # warm the box
LEGS=${LEGS:-fused chain pipe7 ref cat}     # which parts to run
has() { case " $LEGS " in *" $1 "*) return 0;; esac; return 1; }
WFM='convert_u8_f | shift_addition_cc -0.085 | fir_decimate_cc 10 0.05 HAMMING | fmdemod_quadri_cf | fractional_decimator_ff 5.5 | deemphasis_wfm_ff 48000 50e-6 | convert_f_s16'
for rep in $(seq 1 ${REPS:-1}); do
for b in ${BLOCKS:-262144 1048576 4194304}; do
  has fused || break
  export CSDR_AMD_BLOCK=$b
  ta=$(run sh -c '${CSDR_BIN:-csdr_amd/csdr} wfm_chain_u8_s16 -0.085 < /tmp/iq_a.u8'); tb=$(run sh -c '${CSDR_BIN:-csdr_amd/csdr} wfm_chain_u8_s16 -0.085 < /tmp/iq_b.u8')
  python -c "ta, tb = $ta, $tb; r = (1920e6 - 480e6) / (tb - ta); print('wfm_chain_u8_s16 block=$b: %.0f MS/s streaming, start-up %.2f s (480 M samples %.2f s, 1920 M samples %.2f s)' % (r / 1e6, ta - 480e6 / r, ta, tb))"
done
done
export CSDR_AMD_BLOCK=4194304
if has chain; then
ta=$(run sh -c "${CSDR_BIN:-csdr_amd/csdr} chain '$WFM' < /tmp/iq_a.u8"); tb=$(run sh -c "${CSDR_BIN:-csdr_amd/csdr} chain '$WFM' < /tmp/iq_b.u8")
python -c "ta, tb = $ta, $tb; r = (1920e6 - 480e6) / (tb - ta); print('chain of seven unfused commands block=4194304: %.0f MS/s streaming, start-up %.2f s' % (r / 1e6, ta - 480e6 / r))"
fi
# the LITERAL shell pipeline of README.md:66: seven processes of this csdr, adjacent ones handing blocks over in HBM (csdr_cli.cpp "device hand-off"), the same with
# the hand-off switched off (bytes through every pipe: D2H + pipe + H2D per stage), and the reference's own binary (built by oracle/Makefile) on the same file
PIPE7() { echo "$1 convert_u8_f < $2 | $1 shift_addition_cc -0.085 | $1 fir_decimate_cc 10 0.05 HAMMING | $1 fmdemod_quadri_cf | $1 fractional_decimator_ff 5 | $1 deemphasis_wfm_ff 48000 50e-6 | $1 convert_f_s16"; }
for ipc in ${IPCS:-1 0}; do
  has pipe7 || break
  export CSDR_AMD_IPC=$ipc
  ta=$(run timeout 120 sh -c "$(PIPE7 ${CSDR_BIN:-csdr_amd/csdr} /tmp/iq_a.u8)"); tb=$(run timeout 120 sh -c "$(PIPE7 ${CSDR_BIN:-csdr_amd/csdr} /tmp/iq_b.u8)")
  python -c "ta, tb = $ta, $tb; r = (1920e6 - 480e6) / (tb - ta); print('seven csdr processes in a shell pipeline, CSDR_AMD_IPC=$ipc (1 = device hand-off between the processes, 0 = bytes through the pipes), block=4194304: %.0f MS/s streaming, start-up %.2f s (480 M samples %.2f s, 1920 M samples %.2f s)' % (r / 1e6, ta - 480e6 / r, ta, tb))"
done
unset CSDR_AMD_IPC
if has ref && [ -x oracle/_ref/csdr ]; then
  head -c 192000000 /tmp/iq_a.u8 > /tmp/iq_r.u8
  tr=$(run timeout 120 sh -c "$(PIPE7 oracle/_ref/csdr /tmp/iq_r.u8)")
  python -c "print('the reference csdr (oracle/_ref/csdr, this box, one process per command = 7 host threads): %.1f MS/s (96 M samples in %.2f s)' % (96e6 / $tr / 1e6, $tr))"
  rm -f /tmp/iq_r.u8
fi
has cat && ta=$(run sh -c 'cat /tmp/iq_b.u8') &&
python -c "print('cat of the same file: %.0f MS/s' % (1920e6 / $ta / 1e6))"
rm -f /tmp/iq_a.u8 /tmp/iq_b.u8
