#!/bin/bash
# build a variant of libaisgpu.so into tools/ab/NAME.so:  tools/build_variant.sh NAME -DFLAG=.. ...   (prints VGPR / scratch of the kernels named in KERNELS)
cd "$(dirname "$0")/../ais-catcher_amd/csrc"
NAME=$1; shift
mkdir -p ../../tools/ab /tmp/bv_$NAME
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC -Wall -Wno-unused-result -Wno-unused-value"
timeout 900 /opt/rocm/bin/hipcc $F "$@" -x hip -shared -o ../../tools/ab/$NAME.so kernels.hip aisgpu.cpp 2>&1 | grep -E "error" 
if [ -n "$KERNELS" ]; then
  (cd /tmp/bv_$NAME && timeout 900 /opt/rocm/bin/hipcc $F "$@" -x hip -c $OLDPWD/kernels.hip --cuda-device-only -S -o k.s 2>/dev/null
   for k in $KERNELS; do echo -n "$NAME $k: "; awk "/\.amdhsa_kernel _ZN4aisk[0-9]*$k/,/\.end_amdhsa_kernel/" k.s < /dev/null | grep -E "next_free_vgpr|private_segment_fixed|group_segment_fixed" | awk '{printf "%s=%s ", $1, $2}'; echo; done)
fi
