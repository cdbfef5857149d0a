#!/bin/bash
# per-process stage time (CSDR_AMD_CLI_TIMING=1) inside the README's NFM (README.md:87), AM (README.md:95) and SSB (README.md:110) pipelines, 240 M samples each
cd $GRAFT_REPO_ROOT 2>/dev/null
python -c "
import numpy as np
np.random.default_rng(1).integers(0, 256, 2 * 240000000, dtype=np.uint8).tofile('/tmp/iq_t.u8')"
export CSDR_AMD_BLOCK=4194304 CSDR_AMD_CLI_TIMING=1
C=csdr_amd/csdr
runpipe() { # name, stages...
  local name=$1; shift; local n=$#; local cmd=""; local i=0
  for st in "$@"; do if [ $i -eq 0 ]; then cmd="$C $st < /tmp/iq_t.u8 2>/tmp/e$i"; else cmd="$cmd | $C $st 2>/tmp/e$i"; fi; i=$((i+1)); done
  local s=$(date +%s.%N); timeout 100 sh -c "$cmd > /dev/null"; local e=$(date +%s.%N)
  python -c "print('$name: whole run %.2f s for 240 M samples' % ($e - $s))"
  for j in $(seq 0 $((n-1))); do grep -h stage /tmp/e$j | cut -c1-150; done
}
runpipe NFM "convert_u8_f" "shift_addition_cc 0.11" "fir_decimate_cc 50 0.005 HAMMING" "fmdemod_quadri_cf" "limit_ff" "deemphasis_nfm_ff 48000" "fastagc_ff" "convert_f_s16"
runpipe AM "convert_u8_f" "shift_addition_cc 0.11" "fir_decimate_cc 50 0.005 HAMMING" "amdemod_cf" "fastdcblock_ff" "agc_ff" "limit_ff" "convert_f_s16"
runpipe SSB "convert_u8_f" "shift_addition_cc 0.11" "fir_decimate_cc 50 0.005 HAMMING" "bandpass_fir_fft_cc 0 0.1 0.05" "realpart_cf" "agc_ff" "limit_ff" "convert_f_s16"
rm -f /tmp/iq_t.u8 /tmp/e?
