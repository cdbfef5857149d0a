#!/bin/bash
# per-process stage time inside the literal seven-process pipeline (CSDR_AMD_CLI_TIMING=1 in every process; device hand-off on), 240 M samples
cd $GRAFT_REPO_ROOT 2>/dev/null
python -c "
import numpy as np
np.random.default_rng(1).integers(0, 256, 2 * 240000000, dtype=np.uint8).tofile('/tmp/iq_t.u8')"
export CSDR_AMD_BLOCK=4194304 CSDR_AMD_CLI_TIMING=1
C=csdr_amd/csdr
s=$(date +%s.%N)
timeout 100 sh -c "$C convert_u8_f < /tmp/iq_t.u8 2>/tmp/e0 | $C shift_addition_cc -0.085 2>/tmp/e1 | $C fir_decimate_cc 10 0.05 HAMMING 2>/tmp/e2 | $C fmdemod_quadri_cf 2>/tmp/e3 | $C fractional_decimator_ff 5 2>/tmp/e4 | $C deemphasis_wfm_ff 48000 50e-6 2>/tmp/e5 | $C convert_f_s16 2>/tmp/e6 > /dev/null"
e=$(date +%s.%N)
python -c "print('whole run %.2f s for 240 M samples' % ($e - $s))"
grep -h stage /tmp/e0 /tmp/e1 /tmp/e2 /tmp/e3 /tmp/e4 /tmp/e5 /tmp/e6
rm -f /tmp/iq_t.u8 /tmp/e?
