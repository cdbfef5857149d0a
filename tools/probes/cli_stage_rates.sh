#!/bin/bash
# tools/probes/cli_stage_rates.sh -- each command of the README.md:66 pipeline ALONE (file -> /dev/null, CSDR_AMD_BLOCK=4194304): which stage bounds the literal pipeline
cd $GRAFT_REPO_ROOT 2>/dev/null
python - <<PY
import numpy as np
rng = np.random.default_rng(1)
rng.integers(0, 256, 2 * 96000000, dtype=np.uint8).tofile("/tmp/s_u8")            # 96 M complex samples of u8 IQ
(rng.random(2 * 48000000, dtype=np.float32) - 0.5).tofile("/tmp/s_cf")             # 48 M complexf
(rng.random(48000000, dtype=np.float32) - 0.5).tofile("/tmp/s_f")                  # 48 M floats
PY
export CSDR_AMD_BLOCK=4194304
run() { local n=$1 unit=$2 f=$3; shift 3; local best=1e9; for i in 1 2 3; do local s=$(date +%s.%N); timeout 120 csdr_amd/csdr "$@" < $f > /dev/null 2>/dev/null; local e=$(date +%s.%N); best=$(python -c "print(min($best, $e - $s))"); done; python -c "print('%-44s %8.0f M %s/s   (%.2f s for %d M incl. ~0.25 s start-up)' % ('$*', $n / max($best - 0.25, 1e-3) / 1e6, '$unit', $best, $n / 1e6))"; }
csdr_amd/csdr convert_u8_f < /tmp/s_u8 > /dev/null 2>&1
run 192000000 floats /tmp/s_u8 convert_u8_f
run 48000000 complex /tmp/s_cf shift_addition_cc -0.085
run 48000000 complex /tmp/s_cf fir_decimate_cc 10 0.05 HAMMING
run 48000000 complex /tmp/s_cf fmdemod_quadri_cf
run 48000000 floats /tmp/s_f fractional_decimator_ff 5
run 48000000 floats /tmp/s_f deemphasis_wfm_ff 48000 50e-6
run 48000000 floats /tmp/s_f convert_f_s16
rm -f /tmp/s_u8 /tmp/s_cf /tmp/s_f
