#!/bin/bash
# round-2 GPU batch L: the measured lines of every config (verify + CPU baselines) and the rocprof / PMC summaries behind them
cd $GRAFT_REPO_ROOT; out=gpurun_out/r2l; mkdir -p $out
timeout 300 python bench.py --verify > $out/r2l_bench_n1.json 2> $out/bench.err; tail -c 600 $out/r2l_bench_n1.json
timeout 300 python bench_nfm.py --verify > $out/r2l_nfm_n1.json 2> $out/nfm.err; tail -c 300 $out/r2l_nfm_n1.json
timeout 300 python bench_fir.py --verify > $out/r2l_fir_n1.json 2> $out/fir.err; tail -c 200 $out/r2l_fir_n1.json
timeout 300 python bench_fir.py --decimation 50 --tbw 0.005 --streams 64 --verify > $out/r2l_fir50_n1.json 2> $out/fir50.err; tail -c 200 $out/r2l_fir50_n1.json
timeout 300 python bench_fftfilt.py --verify > $out/r2l_fftfilt_n1.json 2> $out/fftfilt.err; tail -c 200 $out/r2l_fftfilt_n1.json
timeout 300 python bench_fastddc.py --verify > $out/r2l_fastddc_n1.json 2> $out/fastddc.err; tail -c 200 $out/r2l_fastddc_n1.json
timeout 600 bash tools/profile_bench.sh r2l_wfm k_wfm_mfma_seq bench.py > $out/prof_wfm.log 2>&1; tail -5 $out/prof_wfm.log | cut -c1-180
timeout 600 bash tools/profile_bench.sh r2l_fastddc k_ddc_gemm bench_fastddc.py > $out/prof_fastddc.log 2>&1; tail -9 $out/prof_fastddc.log | cut -c1-180
timeout 600 bash tools/profile_bench.sh r2l_fir50 k_fir_mfma bench_fir.py --decimation 50 --tbw 0.005 --streams 64 > $out/prof_fir50.log 2>&1; tail -4 $out/prof_fir50.log | cut -c1-180
timeout 600 bash tools/profile_bench.sh r2l_nfm k_ddc_mfma bench_nfm.py > $out/prof_nfm.log 2>&1; tail -12 $out/prof_nfm.log | cut -c1-180
for f in $out/*.err; do [ -s $f ] && { grep -v "amdgpu.ids\|^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" $f | tail -3 | cut -c1-300; }; done
