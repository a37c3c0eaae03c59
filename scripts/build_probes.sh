#!/bin/bash
# Builds the round-5 / round-6 ubench probes from their sources (the binaries are git-ignored: ADVICE r05).  Runs here (hipcc
# cross-compiles gfx950 without a GPU; the binaries then travel to the GPU box with the snapshot) or on the box itself.
set -eu
cd "$(dirname "$0")/ubench"
LIBF="-O3 -std=c++17 -ffp-contract=off -fno-fast-math -mllvm -amdgpu-mfma-vgpr-form -w -I../../include -I../../hevc-complexity-reduction_amd/csrc"
H="/opt/rocm/bin/hipcc --offload-arch=gfx950"
$H -O2 -mllvm -amdgpu-mfma-vgpr-form conv2_pad_probe.hip -o conv2_pad_probe
$H $LIBF fc1_tile_f16_probe.hip -o fc1_tile_f16_probe
$H -O3 -o mfma16_valu mfma16_valu.hip
$H -O3 -o occupancy_probe occupancy_probe.hip
$H $LIBF -DTRUNK16_STAMPS trunk16_probe.hip -o trunk16_probe
for f in "$@"; do $H $LIBF "$f.hip" -o "$f"; done   # further probes by name
