#!/bin/bash
# GPU box: the HM encoder built by oracle/build_ref_hm.sh (predictor called in process through
# the C ABI) encodes the small synthetic sequence; outputs land in gpurun_out/hm_inprocess/ for
# scripts/hm_inprocess_check.py (run where the reference's prebuilt HM lives).
set -eu
REPO=$PWD
python scripts/make_hm_case.py | tail -1
D=$REPO/gpurun_out/hm_inprocess
rm -rf $D; mkdir -p $D
cp gpurun_out/hm/seq.yuv gpurun_out/hm/Thr_info.txt $D/
cd $D
ETHCNN_HM_DUMP=cu_depth.dat ETHCNN_SYNTHETIC_SEED=9 ETHCNN_HEAD_GAIN=8.0 $REPO/oracle/_ref/hm_ai/TAppEncoderInProcess -c $REPO/scripts/hm_intra_test.cfg \
    -i seq.yuv -wdt 416 -hgt 240 -fr 30 -f 4 -q 32 -b str.bin -o "" > encode.log 2>&1 || { tail -5 encode.log; exit 1; }
grep -E "ethcnn|Total Time|Bytes written" encode.log
# the in-process build writes no cu_depth.dat of its own; ETHCNN_HM_DUMP makes the hook append each picture's probabilities
cmp cu_depth.dat $REPO/gpurun_out/hm/cu_depth_gpu.dat && echo "per-picture probabilities (in-process dump) == cu_depth.dat (python launcher)"
md5sum str.bin
# the reference's encoder UNCHANGED: its hook spawns `python video_to_cu_depth.py <yuv> <w> <h> <qp>` in the cwd
U=$REPO/gpurun_out/hm_unchanged
rm -rf $U; mkdir -p $U; cp seq.yuv Thr_info.txt $U/; cd $U
ln -s $REPO/video_to_cu_depth.py video_to_cu_depth.py
ETHCNN_SYNTHETIC_SEED=9 ETHCNN_HEAD_GAIN=8.0 $REPO/oracle/_ref/hm_ai/TAppEncoderUnchanged -c $REPO/scripts/hm_intra_test.cfg \
    -i seq.yuv -wdt 416 -hgt 240 -fr 30 -f 4 -q 32 -b str.bin -o "" > encode.log 2>&1 || { tail -5 encode.log; exit 1; }
grep -E "Predicting Time|Total Time|Bytes written" encode.log
cmp str.bin $D/str.bin && echo "unchanged HM + drop-in launcher: bitstream == in-process build"
md5sum str.bin
rm -f seq.yuv $D/seq.yuv
