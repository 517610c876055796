#!/bin/bash
# builds knocked-out / instrumented variants of k_encode into tools/variants/ (timing only; the ABL_* builds produce
# wrong bytes by construction).  Run them on the GPU box with tools/time_variants.sh NAME... (timing) or
# tools/try_variants.sh NAME... (encode parity tests first, then timing); tools/time_kinds.sh NAME KIND... picks the clip.
#   A_NOGATHER  constant entries instead of the table look-ups      A_SMALLLUT  table masked to 2 MiB (always an L2 hit)
#   A_NOCOPY    bytes emitted into LDS, copy-out skipped            PROF        phase stamps printed by agmv_hip_check
#   DENSE       dense cube index (32 MiB table, 12 VALU per look-up)  PIX0      pixel loads with the default cache policy
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/variants
B="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -shared -std=c++17 libagmv_amd/csrc/agmv_hip.hip"
$B -DABL_NOGATHER -o tools/variants/libagmv_hip_A_NOGATHER.so &
$B -DABL_SMALLLUT -o tools/variants/libagmv_hip_A_SMALLLUT.so &
$B -DABL_NOCOPYOUT -o tools/variants/libagmv_hip_A_NOCOPY.so &
$B -DENC_PROF -o tools/variants/libagmv_hip_PROF.so &
$B -DLUT_SPARSE=0 -o tools/variants/libagmv_hip_DENSE.so &
$B -DENC_PIXAUX=0 -o tools/variants/libagmv_hip_PIX0.so &
$B -o tools/variants/libagmv_hip_CUR.so &
wait
ls tools/variants
