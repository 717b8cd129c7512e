#!/bin/bash
# Build a variant of libparl_hip.so whose conv_kernels.o was compiled with extra flags (e.g. -DPARLHIP_BPF=8)
# into build_exp/<name>.so.  Usage: tools/build_conv_variant.sh <name> <hipcc flags...>
set -e
R=$(cd $(dirname $0)/.. && pwd)
name=$1; shift
mkdir -p $R/build_exp /tmp/convvar_$name
cd $R/parl_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fvisibility=hidden "$@" \
  -c conv_kernels.hip -o /tmp/convvar_$name/conv_kernels.o
objs=$(ls $R/parl_amd/csrc/*.o | grep -v conv_kernels.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/convvar_$name/conv_kernels.o -o $R/build_exp/$name.so
echo built build_exp/$name.so
