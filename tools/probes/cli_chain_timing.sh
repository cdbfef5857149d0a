#!/bin/bash
# where an unfused in-process chain spends its time (CSDR_AMD_CLI_TIMING=1): the README.md:66 commands with fractional_decimator_ff 5.5 (no fused pattern), 480 M samples
cd $GRAFT_REPO_ROOT 2>/dev/null
python -c "
import numpy as np
np.random.default_rng(1).integers(0, 256, 2 * 240000000, dtype=np.uint8).tofile('/tmp/iq_t.u8')"
export CSDR_AMD_BLOCK=4194304 CSDR_AMD_CLI_TIMING=1
for fd in 5.5 5; do
WFM="convert_u8_f | shift_addition_cc -0.085 | fir_decimate_cc 10 0.05 HAMMING | fmdemod_quadri_cf | fractional_decimator_ff $fd | deemphasis_wfm_ff 48000 50e-6 | convert_f_s16"
[ $fd = 5 ] && export CSDR_AMD_CHAIN_NOFUSE=1
s=$(date +%s.%N); timeout 100 csdr_amd/csdr chain "$WFM" < /tmp/iq_t.u8 2>&1 > /dev/null | grep -E "stage|fused" ; e=$(date +%s.%N)
python -c "print('fractional rate $fd: whole run %.2f s for 240 M samples' % ($e - $s))"
done
rm -f /tmp/iq_t.u8
