#!/bin/bash
# Everything the judged numbers come from, in one GPU call:  tools/profile_round.sh <tag>
#   1. rocprofv3 --kernel-trace --stats of `python3 bench.py --no-extra` (the headline steps alone, so that k_queue's
#      average duration is the one roofline.kernel_ms must agree with)  -> profiles/<tag>_kernel_stats.csv
#      and of the whole `python3 bench.py` (extra_configs too: k_bounce, k_mesh_brute, k_mesh_bvh, k_raytrace1 ...)
#                                                            -> profiles/<tag>_extra_kernel_stats.csv
#   2. SQ counter passes of one k_queue launch               -> profiles/<tag>_pmc_k_queue.json
#   3. FETCH_SIZE / WRITE_SIZE passes, work queue            -> profiles/<tag>_hbm.json
#   4. the same for the compacted wavefront pipeline         -> profiles/<tag>_compacted_hbm_summary.json
#   5. kernel stats + SQ counters of configs[4] (1.31 M faces) -> profiles/<tag>_mesh_brute_stats_pmc.txt
# then bench.py once more, un-profiled, so that its line uses this round's instruction count and traffic
#                                                            -> profiles/<tag>_bench.json
export TMPDIR=/tmp
TAG=${1:?tag}
mkdir -p profiles gpurun_out
OUT=gpurun_out/prof_$TAG; rm -rf $OUT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --no-cpu-baseline --no-extra > gpurun_out/prof_$TAG.out 2>&1 || exit 1
cp $(ls $OUT/*/*kernel_stats.csv | head -1) profiles/${TAG}_kernel_stats.csv || exit 1
rm -rf ${OUT}_extra
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d ${OUT}_extra -- python3 bench.py --no-cpu-baseline > gpurun_out/prof_${TAG}_extra.out 2>&1 || exit 1
cp $(ls ${OUT}_extra/*/*kernel_stats.csv | head -1) profiles/${TAG}_extra_kernel_stats.csv || exit 1
tools/pmc_fused.sh $TAG > gpurun_out/pmc_$TAG.txt 2>&1 || exit 1
cp gpurun_out/pmc_$TAG/summary.json profiles/${TAG}_pmc_k_queue.json || exit 1
tools/pmc_hbm.sh $TAG > gpurun_out/hbm_$TAG.txt 2>&1 || exit 1
cp gpurun_out/pmc_hbm_$TAG/summary.json profiles/${TAG}_hbm.json || exit 1
PIPE=1 tools/pmc_hbm.sh ${TAG}_compacted > gpurun_out/hbm_${TAG}_compacted.txt 2>&1 || exit 1
cp gpurun_out/pmc_hbm_${TAG}_compacted/summary.json profiles/${TAG}_compacted_hbm_summary.json || exit 1
tools/pmc_mesh.sh $TAG > profiles/${TAG}_mesh_brute_stats_pmc.txt 2>&1 || exit 1
cp gpurun_out/mesh_$TAG/summary.json profiles/${TAG}_mesh_pmc.json || exit 1
timeout -k 10 400 python3 bench.py > profiles/${TAG}_bench.json 2> gpurun_out/bench_$TAG.err || exit 1
mkdir -p gpurun_out/profiles_$TAG && cp profiles/${TAG}_* gpurun_out/profiles_$TAG/
python3 -c "
import json; d = json.load(open('profiles/${TAG}_bench.json'))
print({k: d[k] for k in ('value', 'ms_per_step', 'ms_per_step_min')}); print(d['roofline'])"
