#!/bin/bash
# tools/pmc_fftfilt.sh <tag> -- SQ issue / LDS counters of the one-pass FFT filter kernel (k_fftfilt_wave / k_fftfilt_team / k_fftfilt_lds) at 1023 and 4095 taps (BASELINE config 3): gpurun_out/<tag>_{1023,4095}/summary.json
tag=${1:-r6_fftfilt}
P="SQ_INSTS_VALU,SQ_INSTS_SALU,SQ_INSTS_LDS,SQ_INSTS_VMEM,SQ_WAVES:SQ_BUSY_CYCLES,SQ_WAIT_INST_ANY,SQ_ACTIVE_INST_VALU,SQ_ACTIVE_INST_ANY,SQ_WAVE_CYCLES:SQ_ACTIVE_INST_LDS,SQ_LDS_BANK_CONFLICT,SQ_LDS_IDX_ACTIVE,SQ_WAIT_INST_LDS:SQ_LDS_ADDR_CONFLICT,SQ_ACTIVE_INST_VMEM,SQ_WAIT_ANY,SQ_BUSY_CU_CYCLES"
for taps in 1023 4095; do
  kern=k_fftfilt_wave; [ $taps -gt 1025 ] && kern=k_fftfilt_team          # (not "k_fftfilt_": that would average the window kernel with k_fftfilt_hist)
  [ -n "$CSDR_AMD_FFTFILT_LDS_MODE" ] && [ "$CSDR_AMD_FFTFILT_LDS_MODE" != 0 ] && kern=k_fftfilt_lds
  bash tools/pmc_sq.sh ${tag}_$taps $kern "$P" -- python bench_fftfilt.py --steps 2 --warmup 1 --no-sweep --taps $taps --no-cpu-baseline
done
