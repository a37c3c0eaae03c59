#!/usr/bin/env python
"""Drop-in launcher for the LDP predictor daemon: copy or symlink into HM-LDP's bin/ directory
(next to TAppEncoderStatic, Thr_info.txt and the model_LDP_* files) and start it before the
encoder, exactly like the reference's resi_to_cu_depth_LDP.py.  Set ETHCNN_HOME to the
repository root if this file is copied rather than symlinked."""
import importlib
import os
import sys


def _main():
    home = os.environ.get("ETHCNN_HOME") or os.path.dirname(os.path.realpath(__file__))
    # Default: the same protocol served by the C daemon (tools/resi_to_cu_depth_ldp.c over the C ABI; byte-identical
    # cu_depth.dat / state.dat per frame: tests/test_gpu_ldp_native.py; ~2.3x sooner from the encoder's side) whenever its
    # binary is built.  --python or ETHCNN_LDP_NATIVE=0 keeps the Python daemon; --native / ETHCNN_LDP_NATIVE=1 insists on
    # the C one (an error when it is not built instead of a silent fall-back to the slower daemon).
    args = sys.argv[1:]
    env = os.environ.get("ETHCNN_LDP_NATIVE", "")
    exe = os.path.join(home, "hevc-complexity-reduction_amd", "bin", "resi_to_cu_depth_ldp")
    want_python = "--python" in args or env == "0"
    insist = "--native" in args or env not in ("", "0")
    if not want_python and (insist or os.path.isfile(exe)):
        if not os.path.isfile(exe):
            sys.stderr.write("resi_to_cu_depth_LDP.py: %s is not built (python __graft_entry__.py build)\n" % exe)
            return 1
        os.execv(exe, [exe] + [a for a in args if a not in ("--native", "--python")])
    sys.argv = [sys.argv[0]] + [a for a in args if a not in ("--native", "--python")]
    sys.path.insert(0, home)
    pkg = importlib.import_module("hevc-complexity-reduction_amd")
    mod = importlib.import_module("hevc-complexity-reduction_amd.resi_to_cu_depth_LDP")
    return mod.main(sys.argv)


if __name__ == "__main__":
    sys.exit(_main())
