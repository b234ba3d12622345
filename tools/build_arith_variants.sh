#!/bin/bash
# A/B builds of the 29-bit-limb arithmetic (csrc/ff29.hip.hpp, csrc/ec29.hip.hpp):
#   head  : whatever lib/libzkmi355_head.so holds (copy the library of the commit to compare against there first)
#   v1    : fused group law, compiler-associated column sums     (-DZK_EC_FUSED=1 -DZK_MAD_CHAIN=0)
#   v2    : fused group law, carry-first column sums (block asm) (-DZK_EC_FUSED=1 -DZK_MAD_CHAIN=2)
#   cMNQ  : v1 objects with the v2 object of a group where its bit is 1 -- M: msm + ecntt, N: ntt, Q: quotient + vec
# The translation units outside these groups do not use the arithmetic.
set -e
cd "$(dirname "$0")/../zkevm-circuits_amd"
make -j4 VARIANT=v1 EXTRA="-DZK_EC_FUSED=1 -DZK_MAD_CHAIN=0" > /dev/null &
make -j4 VARIANT=v2 EXTRA="-DZK_EC_FUSED=1 -DZK_MAD_CHAIN=2" > /dev/null &
wait
OTHERS="api srs params lookup prover comm wire"
for m in 0 1; do for n in 0 1; do for q in 0 1; do
    objs=""
    for o in $OTHERS; do objs="$objs build_v1/$o.o"; done
    for o in msm ecntt; do objs="$objs build_v$((m + 1))/$o.o"; done
    objs="$objs build_v$((n + 1))/ntt.o"
    for o in quotient vec; do objs="$objs build_v$((q + 1))/$o.o"; done
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o lib/libzkmi355_c$m$n$q.so $objs -ldl
done; done; done
ls -la lib/
