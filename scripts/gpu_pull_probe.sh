#!/bin/bash
# timeline of the PULL form of the single-launch pass (scripts/ubench/small_probe.hip, -DSMALL_STAMPS) -> gpurun_out/pull_timeline.txt
# and the two transfer probes behind its design (scripts/ubench/row_arrival_probe.hip) -> gpurun_out/row_arrival_probe.txt
set -u
mkdir -p gpurun_out
cd scripts/ubench
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -mllvm -amdgpu-mfma-vgpr-form -w -I../../include -I../../hevc-complexity-reduction_amd/csrc -DSMALL_STAMPS -DETHCNN_EXPERIMENTS small_probe.hip -o /tmp/small_probe || exit 1
{
echo "# scripts/gpu_pull_probe.sh: device-wide 100 MHz stamps of every block of ONE single-launch pass (us after the first block's entry; min / avg / max over"
echo "# the blocks of a role; q1..q4 = quarters of the picture in raster order).  woken = the block's input has landed; computed = its own work is done."
echo "# PULL form: picture in page-locked host memory, read over PCIe by the pull blocks; otherwise the picture is already in HBM (direct gather)."
for sz in "1920 1080" "3840 2160"; do
  echo "=== $sz, picture resident in HBM (direct gather; the launch the round-3 path runs BEHIND its host-to-device copy)"; timeout 60 /tmp/small_probe $sz 0 0
  echo "=== $sz PULL form as shipped (pull blocks: one per group up to 32 groups, else 16; 64 x 16 FC1 tiles)"; timeout 60 /tmp/small_probe $sz 0 1
done
echo "=== A/B, 3840 2160 PULL: pull blocks x FC1 tile shape (launch time by HIP events, best of 20)"
for sh in 0 1; do for k in 8 12 16 24 32 128; do printf "ETHCNN_PULL_BLOCKS=%-3s ETHCNN_SMALL_SHAPE=%s (%s): " $k $sh "$([ $sh = 0 ] && echo "64 x 16 tiles" || echo "64 x 32 tiles")"; ETHCNN_SMALL_SHAPE=$sh ETHCNN_PULL_BLOCKS=$k timeout 60 /tmp/small_probe 3840 2160 0 1 | grep -o "launch [0-9.]* us"; done; done
echo "=== A/B, 1920 1080 PULL: pull blocks"
for k in 8 16 32; do printf "ETHCNN_PULL_BLOCKS=%-3s: " $k; ETHCNN_PULL_BLOCKS=$k timeout 60 /tmp/small_probe 1920 1080 0 1 | grep -o "launch [0-9.]* us"; done
} > ../../gpurun_out/pull_timeline.txt 2>&1
hipcc --offload-arch=gfx950 -O2 row_arrival_probe.hip -o /tmp/row_arrival_probe && timeout 300 /tmp/row_arrival_probe > ../../gpurun_out/row_arrival_probe.txt 2>&1
cd ../..
cut -c1-220 gpurun_out/pull_timeline.txt | tail -40
