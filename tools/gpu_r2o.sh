#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/pmc_sq.sh r2o k_ddc_mfma "SQ_WAVE_CYCLES,SQ_BUSY_CYCLES,SQ_WAIT_ANY,SQ_WAIT_INST_ANY,SQ_ACTIVE_INST_ANY,SQ_ACTIVE_INST_VALU,SQ_VALU_MFMA_BUSY_CYCLES,SQ_INSTS_VALU:SQ_INSTS_VALU_MFMA_I8,SQ_ACTIVE_INST_LDS,SQ_WAIT_INST_LDS,SQ_INSTS_LDS,SQ_LDS_BANK_CONFLICT,SQ_LDS_IDX_ACTIVE,SQ_WAVES,GRBM_GUI_ACTIVE" -- python bench_nfm.py --steps 3 --warmup 1 --no-cpu-baseline
grep -i "error\|invalid\|unknown" gpurun_out/r2o/p1.log | head -3
