#!/bin/bash
# builds knocked-out variants of k_encode (timing only; outputs are wrong by construction)
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/variants
for v in NOGATHER NOCMP NOEMIT NOLOOKBACK "NOGATHER -DABL_NOCMP" "NOGATHER -DABL_NOCMP -DABL_NOEMIT"; do
  name=$(echo "$v" | sed 's/ -DABL_/_/g')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -shared -std=c++17 -DABL_$v libagmv_amd/csrc/agmv_hip.hip -o tools/variants/libagmv_hip_$name.so &
done
wait
ls tools/variants
