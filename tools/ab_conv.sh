#!/bin/bash
# A/B two builds of the library on the same GPU box:  tools/ab_conv.sh <variants...>
# expects diffusion_models_dsdiff_amd/libdsdiff.so (B, current) and gpurun_out/libdsdiff_a.so (A, baseline); alternates runs.
set -e
cd "$(dirname "$0")/.."
for rep in 1 2; do
  for tag in a b; do
    if [ $tag = a ]; then export DSD_LIBRARY=$PWD/tools/_ab/libdsdiff_a.so; else export DSD_LIBRARY=$PWD/diffusion_models_dsdiff_amd/libdsdiff.so; fi
    echo "== $tag (rep $rep)"
    DSD_SHAPES=${DSD_SHAPES:-} python tools/bench_conv.py "$@" 2>/dev/null | head -${AB_LINES:-4}
  done
done
