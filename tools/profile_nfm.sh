#!/bin/bash
# tools/profile_nfm.sh <tag> -- on the GPU box: the two PMC traffic passes of bench_nfm.py's default workload for k_ddc_mfma -> profiles/<tag>_pmc_traffic.json
tag=$1
root=$GRAFT_REPO_ROOT; [ -z "$root" ] && root=$PWD
out=$root/gpurun_out/$tag; mkdir -p $out/run/pmc_fetch $out/run/pmc_write
cd /tmp && export TMPDIR=/tmp; cd $root
python bench_nfm.py 2>/dev/null | tail -1 > $out/run/bench.json
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pf -- python bench_nfm.py --steps 2 --warmup 1 > $out/pf.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pw -- python bench_nfm.py --steps 2 --warmup 1 > $out/pw.log 2>&1
cp $(find $out/pf -name "*counter_collection.csv" | head -1) $out/run/pmc_fetch/
cp $(find $out/pw -name "*counter_collection.csv" | head -1) $out/run/pmc_write/
mkdir -p $root/gpurun_out/profiles_$tag
(cd $root && python tools/pmc_summary.py $out/run $tag k_ddc_mfma && cp profiles/${tag}_pmc_traffic.json gpurun_out/profiles_$tag/)
