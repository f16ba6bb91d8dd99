#!/bin/bash
# dev tool (GPU box): the round's records -- the multi-GPU host path rehearsed on one GPU, refreshed profiles, the full default bench line
R=$(cd "$(dirname "$0")/.." && pwd); cd $R; tag=${1:-r04}
python bench.py --rccl --force-rccl-step --steps 20 --warmup 5 --no-variants --no-cpu-baseline > gpurun_out/${tag}_bench_rccl_rehearsal.json 2> gpurun_out/${tag}_bench_rccl_rehearsal.err || echo "rehearsal failed"
bash tools/profile.sh p2 $tag > gpurun_out/${tag}_profile_p2.log 2>&1 || echo "profile p2 failed"
bash tools/profile.sh p1 $tag > gpurun_out/${tag}_profile_p1.log 2>&1 || echo "profile p1 failed"
bash tools/profile_spatial.sh $tag > gpurun_out/${tag}_profile_spatial.log 2>&1 || echo "profile spatial failed"
python bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench_final.json 2> gpurun_out/${tag}_bench_final.err || echo "bench failed"
tail -c 400 gpurun_out/${tag}_bench_final.json
