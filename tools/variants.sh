#!/bin/bash
# usage: [KSTATS_ARGS='--kind flexible'] tools/variants.sh name1 name2 ... ; times active-gym_amd/lib/libagx_<name>.so variants back to back
for v in "$@"; do
  echo "== $v"; AGX_LIB=$PWD/active-gym_amd/lib/libagx_$v.so tools/kstats.sh gpurun_out/var/$v ${KSTATS_ARGS:-} | grep -v "^$"
done
