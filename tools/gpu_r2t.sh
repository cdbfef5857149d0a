#!/bin/bash
# round-2 GPU batch T: the whole GPU suite, then every config's measured line (verify + CPU baselines) and the rocprof / PMC summaries of the kernels that changed
cd $GRAFT_REPO_ROOT; out=gpurun_out/r2t; mkdir -p $out
timeout 1500 python -m pytest tests -q -x -m gpu 2>&1 | grep -E "passed|failed|Error|error" | tail -3
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 300 python bench.py --verify > $out/r2t_bench_n1.json 2> $out/bench.err; cut -c1-250 $out/r2t_bench_n1.json
timeout 300 python bench_nfm.py --verify > $out/r2t_nfm_n1.json 2> $out/nfm.err; cut -c1-200 $out/r2t_nfm_n1.json
timeout 300 python bench_fir.py --verify > $out/r2t_fir_n1.json 2> $out/fir.err; cut -c1-200 $out/r2t_fir_n1.json
timeout 300 python bench_fir.py --decimation 50 --tbw 0.005 --streams 64 --verify > $out/r2t_fir50_n1.json 2> $out/fir50.err; cut -c1-200 $out/r2t_fir50_n1.json
timeout 300 python bench_fftfilt.py --verify > $out/r2t_fftfilt_n1.json 2> $out/fftfilt.err; cut -c1-200 $out/r2t_fftfilt_n1.json
timeout 300 python bench_fastddc.py --verify > $out/r2t_fastddc_n1.json 2> $out/fastddc.err; cut -c1-200 $out/r2t_fastddc_n1.json
timeout 600 bash tools/profile_bench.sh r2t_fastddc k_ddc_gemm3 bench_fastddc.py > $out/prof_fastddc.log 2>&1; tail -9 $out/prof_fastddc.log | cut -c1-160
timeout 600 bash tools/profile_bench.sh r2t_nfm k_ddc_mfma bench_nfm.py > $out/prof_nfm.log 2>&1; tail -12 $out/prof_nfm.log | cut -c1-160
for f in $out/*.err; do [ -s $f ] && { grep -v "amdgpu.ids\|^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" $f | tail -3 | cut -c1-300; }; done
