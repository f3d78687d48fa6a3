#!/bin/bash
# A library variant for A/B runs (tools/ab_libs.sh): ONE translation unit recompiled with extra flags, everything else -- and the
# link order -- as in the release build.   bash tools/make_variant.sh <name> <tu basename, e.g. scl_lane_kernel> "<extra flags>"
# -> polar-codes_amd/lib/variants/<name>.so   (run `make -C polar-codes_amd` first)
set -e
R=$(cd "$(dirname "$0")/.." && pwd); P=$R/polar-codes_amd; NAME=$1; TU=$2; EXTRA=$3
mkdir -p $P/lib/variants $P/build/var
FLAGS=$(make -C $P -n -W csrc/$TU.hip build/$TU.o | grep -m1 "hipcc.*-c csrc/$TU.hip" | sed "s/.*hipcc //; s/ -c csrc.*//")
/opt/rocm/bin/hipcc $FLAGS $EXTRA -c $P/csrc/$TU.hip -o $P/build/var/${TU}_$NAME.o
OBJS=$(make -C $P -n -W build/$TU.o lib/libpolar_amd.so | grep -m1 "hipcc.*-shared" | sed "s/.*exports.map //; s/ -o lib.*//" | sed "s#build/$TU.o#build/var/${TU}_$NAME.o#")
(cd $P && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=exports.map $OBJS -o lib/variants/$NAME.so)
echo "$P/lib/variants/$NAME.so"
