#!/bin/bash
# Dev tool (CPU only): generate the translated cartridges, compile them for the host on top of the oracle
# (tests/tools/cart_host) and compare with the oracle's own atari_frame(), N frames per game.
# Usage: tools/cart_host_check.sh [frames=600]   (env PARLHIP_TRACE_LOOPS / PARLHIP_LOOP_REENTRY are passed on)
set -e
R=$(cd $(dirname $0)/.. && pwd)
D=${CART_HOST_DIR:-/tmp/cart_host_check}
N=${1:-600}
mkdir -p $D
python3 $R/parl_amd/csrc/gen_cart_native.py $D/cart_native.gen.hpp pong=$R/roms/pong.bin breakout=$R/roms/breakout.bin
gcc -O1 -std=c11 -ffp-contract=off -c $R/tests/tools/cart_host/shim.c -o $D/shim.o
g++ -O1 -std=c++17 -I $D -c $R/tests/tools/cart_host/main.cpp -o $D/main.o
g++ $D/shim.o $D/main.o -lm -o $D/cart_host
$D/cart_host --alu
$D/cart_host $R/roms/pong.bin 1 $N
$D/cart_host $R/roms/breakout.bin 2 $N
