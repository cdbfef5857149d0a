#!/bin/bash
# tools/sweep_cli_readers.sh -- the fused CLI command on a regular file against the number of pread() threads and the block size (best of three, two file lengths:
# streaming rate = the difference)
cd $GRAFT_REPO_ROOT 2>/dev/null || cd "$(dirname "$0")/.."
python - <<PY
import numpy as np
rng = np.random.default_rng(1)
blk = rng.integers(0, 256, 2 * 24000000, dtype=np.uint8).tobytes()
for name, reps in (("/tmp/iq_a.u8", 20), ("/tmp/iq_b.u8", 80)):
    with open(name, "wb") as f:
        for _ in range(reps): f.write(blk)
PY
run() { local best=1e9; for i in 1 2 3; do local s=$(date +%s.%N); "$@" > /dev/null 2>/dev/null; local e=$(date +%s.%N); best=$(python -c "print(min($best, $e - $s))"); done; echo $best; }
csdr_amd/csdr wfm_chain_u8_s16 -0.085 < /tmp/iq_a.u8 > /dev/null 2>&1
for b in 4194304 16777216; do
  for n in 1 2 4 8 16; do
    export CSDR_AMD_BLOCK=$b CSDR_AMD_READERS=$n
    ta=$(run sh -c 'csdr_amd/csdr wfm_chain_u8_s16 -0.085 < /tmp/iq_a.u8'); tb=$(run sh -c 'csdr_amd/csdr wfm_chain_u8_s16 -0.085 < /tmp/iq_b.u8')
    python -c "ta, tb = $ta, $tb; r = (1920e6 - 480e6) / (tb - ta); print('block=$b readers=$n: %.0f MS/s streaming (480 M samples %.3f s, 1920 M samples %.3f s)' % (r / 1e6, ta, tb))"
  done
done
ta=$(run sh -c 'cat /tmp/iq_b.u8'); python -c "print('cat of the same file: %.0f MS/s' % (1920e6 / $ta / 1e6))"
rm -f /tmp/iq_a.u8 /tmp/iq_b.u8
