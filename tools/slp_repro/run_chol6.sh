#!/bin/bash
# builds the stand-alone 6x6 solve with and without the SLP vectorizer and runs both (on a GPU box)
set -e
cd "$(dirname "$0")"
T=$(mktemp -d)
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 chol6.hip -o $T/chol6_slp
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fno-slp-vectorize chol6.hip -o $T/chol6_noslp
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -mllvm -slp-threshold=8 chol6.hip -o $T/chol6_t8
echo -n "slp:          "; $T/chol6_slp || true
echo -n "slp thresh 8: "; $T/chol6_t8 || true
echo -n "no slp:       "; $T/chol6_noslp || true
