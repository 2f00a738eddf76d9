#!/bin/bash
# tools/micro/tile_clock.sh : in-kernel clock of the persistent tile GEMM + epilogue ablations (diagnostic builds of gemm_tile_p.hip)
cd $GRAFT_REPO_ROOT
for f in "" "-DAHA_ABL_NOSTORE" "-DAHA_ABL_NOEPI"; do
    echo "== build flags: '$f'"
    hipcc --offload-arch=gfx950 -O3 -std=c++17 $f -I aha-_amd/csrc -o /tmp/tile_clock tools/micro/tile_clock.hip 2>/dev/null || exit 1
    timeout -k 10 120 /tmp/tile_clock || exit 1
done
