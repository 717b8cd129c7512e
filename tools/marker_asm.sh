#!/bin/bash
# Dev tool (CPU only): gfx950 assembly of the env kernels with a marker at every translated 6507 block
# (PARLHIP_CART_MARKERS=1), for tools/cart_profile.py.  Output: $OUT/parl_amd/csrc/atari_env.s
# Usage: tools/marker_asm.sh [outdir=/tmp/mk]
set -e
R=$(cd $(dirname $0)/.. && pwd)
OUT=${1:-/tmp/mk}
mkdir -p $OUT/parl_amd/csrc $OUT/include
cp $R/include/parl_hip.h $OUT/include/
cd $R/parl_amd/csrc
cp *.hpp *.hip gen_cart_native.py cart_branch_profile.json $OUT/parl_amd/csrc/
cd $OUT/parl_amd/csrc
rm -f cart_native.gen.hpp
PARLHIP_CART_MARKERS=1 python3 gen_cart_native.py cart_native.gen.hpp pong=$R/roms/pong.bin breakout=$R/roms/breakout.bin
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fvisibility=hidden \
  -mllvm -structurizecfg-skip-uniform-regions=1 $EXTRA_FLAGS -S --cuda-device-only atari_env.hip -o atari_env.s
grep -n "sgpr_spill_count\|\.vgpr_count\|\.sgpr_count" atari_env.s | tail -8
