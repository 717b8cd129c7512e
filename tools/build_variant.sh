#!/bin/bash
# Build a variant of libparl_hip.so whose translated cartridges were GENERATED with different switches
# (environment of gen_cart_native.py) into build_exp/<name>.so; only atari_env.o differs.
# Usage: tools/build_variant.sh <name> [VAR=value ...] [-- extra hipcc flags]
set -e
R=$(cd $(dirname $0)/.. && pwd)
name=$1; shift
envs=(); flags=()
while [ $# -gt 0 ]; do
  if [ "$1" == "--" ]; then shift; flags=("$@"); break; fi
  envs+=("$1"); shift
done
make -s -C $R/parl_amd/csrc -j8
W=/tmp/var_$name
mkdir -p $W/parl_amd/csrc $W/include $R/build_exp
cp $R/include/parl_hip.h $W/include/
cd $R/parl_amd/csrc
cp *.hpp *.hip gen_cart_native.py cart_branch_profile.json $W/parl_amd/csrc/
cd $W/parl_amd/csrc
rm -f cart_native.gen.hpp
env "${envs[@]}" python3 gen_cart_native.py cart_native.gen.hpp pong=$R/roms/pong.bin breakout=$R/roms/breakout.bin
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fvisibility=hidden \
  -mllvm -structurizecfg-skip-uniform-regions=1 "${flags[@]}" -c atari_env.hip -o atari_env.o
objs=$(ls $R/parl_amd/csrc/*.o | grep -v atari_env.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs atari_env.o -o $R/build_exp/$name.so
echo built build_exp/$name.so
