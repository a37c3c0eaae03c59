#!/usr/bin/env python
"""Drop-in launcher: copy or symlink this file into HM's bin/ directory (next to
TAppEncoderStatic, Thr_info.txt and the model files).  HM's unchanged hook runs
`python video_to_cu_depth.py <yuv> <w> <h> <qp>` there (TAppEncCfg.cpp:2317-2321).
Set ETHCNN_HOME to the repository root if this file is copied rather than symlinked."""
import importlib
import os
import sys
import time

_T_UP = time.perf_counter()  # interpreter up (ETHCNN_TIMING=1: where the command's wall time goes, scripts/cold_start.py)


def _main():
    os.environ.setdefault("ETHCNN_T_UP_MS", "%.3f" % (_T_UP * 1e3))
    home = os.environ.get("ETHCNN_HOME") or os.path.dirname(os.path.realpath(__file__))
    sys.path.insert(0, home)
    try:
        pkg = importlib.import_module("hevc-complexity-reduction_amd")
        return pkg.video_to_cu_depth.main(sys.argv)
    except SystemExit:
        raise
    except BaseException as exc:  # any failure must reach HM as a non-zero exit status
        sys.stderr.write("video_to_cu_depth: %s: %s\n" % (type(exc).__name__, exc))
        return 1


if __name__ == "__main__":  # guard: multi-GPU mode spawns worker processes that re-import this file
    _rc = _main()
    if _rc == 0 and os.environ.get("ETHCNN_FAST_EXIT", "1") not in ("", "0"):
        # cu_depth.dat is complete and renamed into place; the interpreter's and the HIP runtime's teardown (~60 ms) would only keep the
        # encoder waiting (it blocks on this command, TAppEncCfg.cpp:2317-2321).  ETHCNN_FAST_EXIT=0 keeps the orderly exit.
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(0)
    sys.exit(_rc)
