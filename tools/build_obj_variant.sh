#!/bin/bash
# Build a variant of libparl_hip.so in which ONE object was compiled with extra flags into build_exp/<name>.so
# (PARL_HIP_LIB=build_exp/<name>.so selects it).  Usage: tools/build_obj_variant.sh <name> <file.hip> <hipcc flags...>
# e.g. tools/build_obj_variant.sh heads4 scan_kernels.hip -DPARLHIP_HEADS_MIN_WAVES=4
set -e
R=$(cd $(dirname $0)/.. && pwd)
name=$1; file=$2; shift 2
obj=${file%.hip}.o
mkdir -p $R/build_exp /tmp/objvar_$name
cd $R/parl_amd/csrc
extra=""
if [ "$file" == "atari_env.hip" ]; then extra="-mllvm -structurizecfg-skip-uniform-regions=1 -Wno-unused-label"; fi
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fvisibility=hidden $extra "$@" \
  -c $file -o /tmp/objvar_$name/$obj
objs=$(ls $R/parl_amd/csrc/*.o | grep -v "/$obj$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/objvar_$name/$obj -o $R/build_exp/$name.so
echo built build_exp/$name.so
