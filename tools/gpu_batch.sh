#!/bin/bash
# tools/gpu_batch.sh <tag> <step> [<step> ...] -- ONE parameterised GPU batch (replaces the per-batch tools/gpu_r2?.sh scripts of round 2).
# Run through gpurun from the repo root:   gpurun --timeout 1500 -- 'bash tools/gpu_batch.sh r3a tests smoke bench'
# Everything a step writes goes to gpurun_out/<tag>/ (lines as <tag>_<name>.json); profile steps leave their summaries in gpurun_out/profiles_<tag>*/.
# Steps:
#   tests[:<pytest -k expr>]   the GPU suite (optionally a subset)          smoke      __graft_entry__.smoke()
#   bench nfm fir fir50 fftfilt fastddc                                      each config's measured line with --verify (bench = BASELINE configs[1])
#   emu:<world>:<shard>:<blocks>[:local]                                      bench_fastddc.py --emulate-world (one rank's work of a world-N bank, null transport)
#   prof_wfm prof_nfm prof_fir prof_fir50 prof_fftfilt prof_fastddc          kernel trace + PMC traffic passes of that bench (tools/profile_bench.sh)
#   ops                                                                       tools/bench_ops.py (shifters, ADPCM)
#   cmd:<shell command>                                                       anything else (quoted as one argument)
cd $GRAFT_REPO_ROOT 2>/dev/null || cd "$(dirname "$0")/.."
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd - > /dev/null
line() { # <name> <script> [args...]: one measured line with --verify, first 260 characters echoed
    local name=$1; shift
    timeout 400 python "$@" > $out/${tag}_${name}.json 2> $out/${name}.err; echo "[$name rc=$?] $(cut -c1-260 $out/${tag}_${name}.json | tail -1)"
}
for step in "$@"; do
    case $step in
    tests)      timeout 1500 python -m pytest tests -q -x -m gpu 2>&1 | grep -E "passed|failed|rror" | tail -4 ;;
    tests:*)    timeout 1500 python -m pytest tests -q -x -m gpu -k "${step#tests:}" 2>&1 | tail -15 ;;
    smoke)      timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 ;;
    bench)      line bench_n1 bench.py --verify ;;
    nfm)        line nfm_n1 bench_nfm.py --verify ;;
    fir)        line fir_n1 bench_fir.py --verify ;;
    fir50)      line fir50_n1 bench_fir.py --decimation 50 --tbw 0.005 --streams 64 --verify ;;
    fftfilt)    line fftfilt_n1 bench_fftfilt.py --verify ;;
    fastddc)    line fastddc_n1 bench_fastddc.py --verify ;;
    emu:*)      IFS=: read -r _ w sh nb loc <<< "$step"
                line fastddc_rank_w${w}_${sh}_b${nb}${loc:+_local} bench_fastddc.py --emulate-world $w --shard $sh --blocks $nb --steps 100 ${loc:+--local-input} ;;
    prof_wfm)     timeout 900 bash tools/profile_bench.sh ${tag}_wfm k_wfm_mfma_seq bench.py 2>&1 | tail -10 | cut -c1-200 ;;
    prof_nfm)     timeout 900 bash tools/profile_bench.sh ${tag}_nfm k_ddc_mfma bench_nfm.py 2>&1 | tail -12 | cut -c1-200 ;;
    prof_fir)     timeout 900 bash tools/profile_bench.sh ${tag}_fir k_fir_poly bench_fir.py 2>&1 | tail -8 | cut -c1-200 ;;
    prof_fir50)   timeout 900 bash tools/profile_bench.sh ${tag}_fir50 k_fir_mfma bench_fir.py --decimation 50 --tbw 0.005 --streams 64 2>&1 | tail -8 | cut -c1-200 ;;
    prof_fftfilt) timeout 900 bash tools/profile_bench.sh ${tag}_fftfilt k_fftfilt_ bench_fftfilt.py 2>&1 | tail -8 | cut -c1-200 ;;
    prof_fastddc) timeout 900 bash tools/profile_bench.sh ${tag}_fastddc k_ddc_gemm3 bench_fastddc.py 2>&1 | tail -10 | cut -c1-200 ;;
    ops)        timeout 600 python tools/bench_ops.py > $out/${tag}_ops.jsonl 2> $out/ops.err; cut -c1-200 $out/${tag}_ops.jsonl ;;
    cmd:*)      bash -c "${step#cmd:}" ;;
    *)          echo "unknown step $step" ;;
    esac
done
for f in $out/*.err; do [ -s $f ] && { grep -v "amdgpu.ids\|^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" $f | tail -3 | cut -c1-300; }; done
exit 0
