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
    sys.path.insert(0, home)
    pkg = importlib.import_module("hevc-complexity-reduction_amd")
    mod = importlib.import_module("hevc-complexity-reduction_amd.resi_to_cu_depth_LDP")
    return mod.main(sys.argv)


if __name__ == "__main__":
    sys.exit(_main())
