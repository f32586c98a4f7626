#!/bin/bash
# Everything the judged numbers come from, in one GPU call:  tools/profile_round.sh <tag>
#   1. python3 bench.py                                   -> profiles/<tag>_bench.json
#   2. rocprofv3 --kernel-trace --stats of the same command -> profiles/<tag>_kernel_stats.csv
#   3. SQ counter passes of one k_queue launch            -> profiles/<tag>_pmc_k_queue.json
#   4. FETCH_SIZE / WRITE_SIZE passes                     -> profiles/<tag>_hbm.json
# then bench.py once more so that its line uses this round's instruction count and traffic.
export TMPDIR=/tmp
TAG=${1:?tag}
mkdir -p profiles gpurun_out
OUT=gpurun_out/prof_$TAG; rm -rf $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --no-cpu-baseline > gpurun_out/prof_$TAG.out 2>&1 || exit 1
cp $(ls $OUT/*/*kernel_stats.csv | head -1) profiles/${TAG}_kernel_stats.csv || exit 1
tools/pmc_fused.sh $TAG > gpurun_out/pmc_$TAG.txt 2>&1 || exit 1
cp gpurun_out/pmc_$TAG/summary.json profiles/${TAG}_pmc_k_queue.json || exit 1
tools/pmc_hbm.sh $TAG > gpurun_out/hbm_$TAG.txt 2>&1 || exit 1
cp gpurun_out/pmc_hbm_$TAG/summary.json profiles/${TAG}_hbm.json || exit 1
timeout -k 10 300 python3 bench.py > profiles/${TAG}_bench.json 2> gpurun_out/bench_$TAG.err || exit 1
mkdir -p gpurun_out/profiles_$TAG && cp profiles/${TAG}_* gpurun_out/profiles_$TAG/
cat profiles/${TAG}_bench.json
