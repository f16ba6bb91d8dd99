#!/bin/bash
# dev tool (GPU box): kernel times of a short bench run under development-flag variants
R=$(cd "$(dirname "$0")/.." && pwd)
for f in "$@"; do
  tag=xp_$(echo $f | tr ',' '_'); [ "$f" = "-" ] && tag=xp_base
  if [ "$f" = "-" ]; then bash $R/tools/tl.sh $tag || exit 1; else bash $R/tools/tl.sh $tag --flags $f || exit 1; fi
  echo "== flags=$f"; python3 $R/tools/stats.py $R/gpurun_out/$tag/g_kernel_stats.csv | head -${XP_LINES:-11}
done
