#!/usr/bin/env python
"""Runs HERE (build container: HM binary present, no GPU).  Feeds the reference's prebuilt,
UNCHANGED HM encoder (/root/reference/HM-16.5_Test_AI/bin/TAppEncoderStatic) a cu_depth.dat
and reports the bitstream md5.  HM's hook (TAppEncCfg.cpp:2317-2321) runs
`python video_to_cu_depth.py <yuv> <w> <h> <qp>` in its cwd; with no GPU here that command is
satisfied by a replay shim that installs a cu_depth.dat produced earlier (on the GPU box by the
real launcher -> gpurun_out/hm/, or by the CPU oracle) -- what is checked is that HM consumes
the file layout and that GPU-made and oracle-made files drive identical bitstreams.

usage: hm_bitstream_check.py <seq.yuv> <w> <h> <qp> <frames> <cu_depth_a.dat> [<cu_depth_b.dat> ...]
"""
import hashlib
import os
import shutil
import subprocess
import sys
import tempfile

HM_BIN = "/root/reference/HM-16.5_Test_AI/bin"


def encode(yuv, w, h, qp, frames, cu_depth):
    d = tempfile.mkdtemp(prefix="hm_")
    shutil.copy(cu_depth, os.path.join(d, "replay_cu_depth.dat"))
    open(os.path.join(d, "Thr_info.txt"), "w").write("0.5 0.5 0.5 0.5 0.5 0.5\n")
    open(os.path.join(d, "video_to_cu_depth.py"), "w").write(
        "import shutil,sys\nassert len(sys.argv)==5\nshutil.copy('replay_cu_depth.dat','cu_depth.dat')\n")
    exe = os.path.join(d, "TAppEncoderStatic")  # the mount is read-only and not executable: run a temp copy
    shutil.copy(os.path.join(HM_BIN, "TAppEncoderStatic"), exe)
    os.chmod(exe, 0o755)
    cmd = [exe, "-c", os.path.join(HM_BIN, "encoder_intra_main.cfg"),
           "-i", os.path.abspath(yuv), "-wdt", str(w), "-hgt", str(h), "-fr", "30", "-f", str(frames), "-q", str(qp),
           "-b", "str.bin", "-o", ""]
    r = subprocess.run(cmd, cwd=d, capture_output=True, text=True)
    if r.returncode != 0:
        raise SystemExit("HM failed:\n" + r.stdout[-2000:] + r.stderr[-2000:])
    data = open(os.path.join(d, "str.bin"), "rb").read()
    tot = [l for l in r.stdout.splitlines() if "Total Time" in l]
    shutil.rmtree(d)
    return hashlib.md5(data).hexdigest(), len(data), (tot[0].strip() if tot else "")


def main():
    yuv, w, h, qp, frames = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    res = []
    for f in sys.argv[6:]:
        md5, n, tot = encode(yuv, w, h, qp, frames, f)
        print("%-40s bitstream %d bytes md5 %s  %s" % (f, n, md5, tot))
        res.append(md5)
    print("IDENTICAL" if len(set(res)) == 1 else "DIFFERENT")
    return 0 if len(set(res)) == 1 else 1


if __name__ == "__main__":
    sys.exit(main())
