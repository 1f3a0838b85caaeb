#!/bin/bash
# conv_ks_kernel knock-outs (tools/build_variant.sh ks_<V> dd_conv_ks.hip -DKS_EXP_<V>): a few Tiramisu launches per experiment build, same box
#   tools/ks_knockouts.sh [heavy|light] [grep -E pattern]
cd $GRAFT_REPO_ROOT
cfg=${1:-heavy}
pat=${2:-"k=576 n=64 |k=64 n=576 |k=1088 n=96 |k=96 n=1088 "}
for v in hip NO_DMA NO_W NO_DMA_W NO_BAR NO_LDS NO_DMA_W_LDS NO_MFMA; do
  echo "== $v"
  if [ $v = hip ]; then python tools/cfg3_launches.py $cfg 8 2>/dev/null; else DD_LIB=tools/exp/libdd_ks_$v.so python tools/cfg3_launches.py $cfg 8 2>/dev/null; fi | grep -E "launches|$pat"
done
