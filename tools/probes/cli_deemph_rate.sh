#!/bin/bash
# deemphasis_wfm_ff alone, file -> /dev/null (48 M floats), default and CSDR_AMD_DEEMPH_SERIAL=1
cd $GRAFT_REPO_ROOT 2>/dev/null
python -c "
import numpy as np
(np.random.default_rng(1).random(48000000, dtype=np.float32) - 0.5).tofile('/tmp/s_f')"
export CSDR_AMD_BLOCK=4194304
csdr_amd/csdr deemphasis_wfm_ff 48000 50e-6 < /tmp/s_f > /dev/null 2>&1
for ser in "" 1; do
  if [ -n "$ser" ]; then export CSDR_AMD_DEEMPH_SERIAL=1; fi
  s=$(date +%s.%N); timeout 100 csdr_amd/csdr deemphasis_wfm_ff 48000 50e-6 < /tmp/s_f > /tmp/o_$ser.f 2>/dev/null; e=$(date +%s.%N)
  python -c "print('deemphasis_wfm_ff 48000 50e-6, serial=[$ser]: %.0f M floats/s (%.2f s for 48 M incl. ~0.25 s start-up)' % (48e6 / max($e - $s - 0.25, 1e-3) / 1e6, $e - $s))"
done
cmp /tmp/o_.f /tmp/o_1.f && echo "outputs identical (bit for bit)"
rm -f /tmp/s_f /tmp/o_.f /tmp/o_1.f
