#!/bin/bash
# Knock-out builds of conv_bwd96_kernel (csrc/dd_conv_bwd96.hip): which part of a tile's time is what.  Build here (no GPU needed), run on the GPU box:
#   tools/b96_knockouts.sh build ; gpurun -- 'tools/b96_knockouts.sh run'
cd "$(dirname "$0")/.."
VARIANTS=${VARIANTS:-"NO_DMA NO_WROLE NO_DROLE NO_FLUSH NO_SEL NO_DMA+NO_WROLE NO_DMA+NO_DROLE SPAN4 SPAN6"}
if [ "$1" = build ]; then
  for v in $VARIANTS; do
    flags=""
    for part in ${v//+/ }; do
      case $part in SPAN*) flags="$flags -DB96_DMA_SPAN=${part#SPAN}";; RING*) flags="$flags -DB96_RING=${part#RING}";; BA*) flags="$flags -DB96_BA=${part#BA}";; *) flags="$flags -DB96_EXP_$part";; esac
    done
    tools/build_variant.sh b96_$v dd_conv_bwd96.hip $flags > /dev/null 2>&1 &
  done; wait
  ls tools/exp/libdd_b96_*.so
else
  for sh in "96 96 64" "192 96 64"; do
    echo "shipped:   $(python tools/conv_bwd_bench.py $sh 128 bf16 10 2>/dev/null)"
    for v in $VARIANTS; do echo "$v: $(DD_LIB=tools/exp/libdd_b96_$v.so python tools/conv_bwd_bench.py $sh 128 bf16 10 2>/dev/null)"; done
  done
fi
