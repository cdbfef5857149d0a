#!/bin/bash
# tools/probes/cli_numa.sh -- the fused CLI command pinned to either socket's cores (taskset), the GPU's NUMA node, and the stage timer: where do 9 GS/s come from
cd $GRAFT_REPO_ROOT 2>/dev/null || cd "$(dirname "$0")/../.."
python - <<PY
import numpy as np
rng = np.random.default_rng(1)
blk = rng.integers(0, 256, 2 * 24000000, dtype=np.uint8).tobytes()
with open("/tmp/iq_b.u8", "wb") as f:
    for _ in range(80): f.write(blk)
PY
echo "gpu numa node: $(cat /sys/class/drm/card*/device/numa_node 2>/dev/null | tr '\n' ' ')"; lscpu | grep -E "NUMA node[0-9]+ CPU" | cut -c1-120
echo "allowed cpus: $(taskset -pc $$ | cut -d: -f2 | cut -c1-80), cgroup quota: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"
export CSDR_AMD_BLOCK=4194304
t() { local best=1e9; for i in 1 2 3; do local s=$(date +%s.%N); "$@" > /dev/null 2>/dev/null; local e=$(date +%s.%N); best=$(python -c "print(min($best, $e - $s))"); done; echo $best; }
csdr_amd/csdr wfm_chain_u8_s16 -0.085 < /tmp/iq_b.u8 > /dev/null 2>&1
for cpus in "" "0-31" "64-95" "128-159" "192-223"; do
  if [ -z "$cpus" ]; then pre=""; else pre="taskset -c $cpus"; fi
  a=$(t sh -c "$pre csdr_amd/csdr wfm_chain_u8_s16 -0.085 < /tmp/iq_b.u8"); c=$(t sh -c "$pre cat /tmp/iq_b.u8")
  echo "cpus [$cpus]: fused 1920 M samples in $a s, cat in $c s"
done
CSDR_AMD_CLI_TIMING=1 csdr_amd/csdr wfm_chain_u8_s16 -0.085 < /tmp/iq_b.u8 2>&1 > /dev/null | tail -3
# the same bytes through the cheapest command there is (u8 -> float of the first 1/..: no): convert_u8_f writes 4 x the bytes; realpart of nothing.  Reader alone: a block size nothing can process
s=$(date +%s.%N); csdr_amd/csdr wfm_chain_u8_s16 -0.085 < /tmp/iq_b.u8 > /dev/null 2>&1; e=$(date +%s.%N); python -c "print('one more plain run: %.3f s' % ($e - $s))"
rm -f /tmp/iq_b.u8
