#!/usr/bin/env python
"""Runs HERE (build container).  The reference's prebuilt, UNCHANGED HM encoder replays the
cu_depth.dat that the in-process build produced on the GPU box (gpurun_out/hm_inprocess/) with the
same configuration file, and the two bitstreams are compared: the in-process hook
drives HM to the same bitstream as the reference's file hand-off."""
import hashlib
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HM_BIN = "/root/reference/HM-16.5_Test_AI/bin"
src = os.path.join(ROOT, "gpurun_out", "hm_inprocess")
d = tempfile.mkdtemp(prefix="hm_")
shutil.copy(os.path.join(src, "cu_depth.dat"), os.path.join(d, "replay_cu_depth.dat"))
open(os.path.join(d, "Thr_info.txt"), "w").write("0.5 0.5 0.5 0.5 0.5 0.5\n")
open(os.path.join(d, "video_to_cu_depth.py"), "w").write(
    "import shutil,sys\nassert len(sys.argv)==5\nshutil.copy('replay_cu_depth.dat','cu_depth.dat')\n")
exe = os.path.join(d, "TAppEncoderStatic")
shutil.copy(os.path.join(HM_BIN, "TAppEncoderStatic"), exe)
os.chmod(exe, 0o755)
yuv = os.path.join(ROOT, "gpurun_out", "hm", "seq.yuv")
r = subprocess.run([exe, "-c", os.path.join(ROOT, "scripts", "hm_intra_test.cfg"), "-i", yuv, "-wdt", "416", "-hgt", "240",
                    "-fr", "30", "-f", "4", "-q", "32", "-b", "str.bin", "-o", ""], cwd=d, capture_output=True, text=True)
if r.returncode != 0:
    raise SystemExit("HM failed:\n" + r.stdout[-1500:] + r.stderr[-1500:])
a = hashlib.md5(open(os.path.join(d, "str.bin"), "rb").read()).hexdigest()
b = hashlib.md5(open(os.path.join(src, "str.bin"), "rb").read()).hexdigest()
shutil.rmtree(d)
print("reference prebuilt HM + replayed cu_depth.dat : md5 %s" % a)
print("in-process HM build on the MI355X            : md5 %s  (%d bytes)" % (b, os.path.getsize(os.path.join(src, "str.bin"))))
print("IDENTICAL" if a == b else "DIFFERENT")
sys.exit(0 if a == b else 1)
