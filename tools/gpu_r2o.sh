#!/bin/bash
cd $GRAFT_REPO_ROOT
CSDR_AMD_FFTFILT_LDS_MODE=2 bash tools/pmc_sq.sh r2o k_fftfilt_lds "SQ_WAVE_CYCLES,SQ_BUSY_CYCLES,SQ_WAIT_ANY,SQ_WAIT_INST_ANY,SQ_ACTIVE_INST_ANY,SQ_ACTIVE_INST_VALU,SQ_ACTIVE_INST_LDS,SQ_INSTS_VALU:SQ_LDS_BANK_CONFLICT,SQ_LDS_IDX_ACTIVE,SQ_WAIT_INST_LDS,SQ_INSTS_LDS,SQ_ACTIVE_INST_VMEM,SQ_INSTS_VMEM_RD,SQ_WAVES,GRBM_GUI_ACTIVE" -- python bench_fftfilt.py --steps 3 --warmup 1 --no-sweep --no-cpu-baseline
tail -3 gpurun_out/r2o/p0.log gpurun_out/r2o/p1.log | cut -c1-300
