#!/bin/bash
# Build experimental variants of libparl_hip.so (only atari_env.o differs) into build_exp/<name>.so.
# Usage: tools/build_exp.sh name1:"-DFLAG ..." name2:"..."   (run tools/exp_variants.sh on the GPU box)
set -e
R=$(cd $(dirname $0)/.. && pwd)
cd $R/parl_amd/csrc
make -s -j8
mkdir -p $R/build_exp
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fvisibility=hidden \
    -mllvm -structurizecfg-skip-uniform-regions=1 $flags -c atari_env.hip -o /tmp/atari_env_$name.o
  objs=$(ls *.o | grep -v atari_env.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/atari_env_$name.o -o $R/build_exp/$name.so
  echo built $name
done
