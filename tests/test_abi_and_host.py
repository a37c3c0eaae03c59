"""CPU: the C-ABI library loads and exports every symbol include/ethcnn.h declares; host
logic that needs no GPU (thresholds, model bands, sharding ranges, CLI failure contract)."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "ethcnn.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ethcnn_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound(pkg):
    import ctypes
    names = _declared()
    assert len(names) >= 30
    lib = ctypes.CDLL(pkg.ethcnn.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), "libethcnn.so lacks %s" % n
    assert set(names) == set(pkg.ethcnn.SIGNATURES), "python binding and header disagree"


def test_library_does_not_depend_on_torch_or_oracle(pkg):
    out = subprocess.check_output(["readelf", "-d", pkg.ethcnn.LIB_PATH]).decode()
    needed = re.findall(r"NEEDED.*\[(.*?)\]", out)
    assert any("amdhip64" in n for n in needed)
    assert not any(("torch" in n) or ("c10" in n) or ("oracle" in n) for n in needed)


def test_product_sources_never_reference_the_oracle():
    """No include / import / dlopen / link of anything under oracle/ (comments may cite it)."""
    bad = re.compile(r"(#\s*include[^\n]*oracle|^\s*(import|from)\s[^\n]*oracle|CDLL[^\n]*oracle|dlopen[^\n]*oracle)", re.M)
    for dirpath, _, files in os.walk(os.path.join(ROOT, "hevc-complexity-reduction_amd")):
        for f in files:
            path = os.path.join(dirpath, f)
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                assert not bad.search(open(path, errors="ignore").read()), path
            if f == "Makefile":
                code = "\n".join(l for l in open(path).read().splitlines() if not l.lstrip().startswith("#"))
                assert "oracle" not in code, path
    assert not bad.search(open(os.path.join(ROOT, "include", "ethcnn.h")).read())


def test_no_gpu_is_a_loud_error(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(pkg.EthCnnError, match="no CPU fallback"):
        pkg.EthCnn()


def test_model_bands(pkg):
    f = pkg.ethcnn.model_name_for_qp  # video_to_cu_depth.py:126-133
    assert [f(q) for q in (0, 22, 24)] == ["model_2000000_qp20~25.dat"] * 3
    assert [f(q) for q in (25, 27, 29)] == ["model_2000000_qp25~30.dat"] * 3
    assert [f(q) for q in (30, 32, 34)] == ["model_2000000_qp30~35.dat"] * 3
    assert [f(q) for q in (35, 37, 51)] == ["model_2000000_qp35~40.dat"] * 3


def test_thresholds(pkg, tmp_path):
    g = pkg.net_CNN.get_thresholds
    p = tmp_path / "Thr_info.txt"
    p.write_text("0.5 0.5 0.5 0.5 0.5 0.5\n")          # shipped AI file
    assert g(str(p)) == (0.5, 0.5)
    p.write_text("0.4 0.6 0.3 0.7 0.2 0.8")             # shipped LDP file: tokens [1],[3]
    assert g(str(p)) == pytest.approx((0.6, 0.7))
    p.write_text("1 2 3 4\n9 9 9 9\n")                   # only the first line counts; [3] carries '\n'
    assert g(str(p)) == (2.0, 4.0)
    for bad in ("0.5 0.5 0.5", "", "a b c d", "0.5  0.5 0.5 0.5"):  # double space -> empty token [1]
        p.write_text(bad)
        with pytest.raises(pkg.EthCnnError):
            g(str(p))
    with pytest.raises(pkg.EthCnnError):
        g(str(tmp_path / "absent.txt"))


def test_constants_mirror(pkg):
    nt = pkg.net_CNN
    assert (nt.IMAGE_SIZE, nt.NUM_CHANNELS, nt.NUM_EXT_FEATURES, nt.NUM_LABEL_BYTES) == (64, 1, 1, 16)
    assert nt.NUM_CONVLAYER_FLAT_FILTERS == 2688
    assert pkg.video_to_cu_depth.SAVE_FILE == "cu_depth.dat"
    assert pkg.ethcnn.ctus_per_frame(1920, 1080) == 510 and pkg.ethcnn.ctus_per_frame(4928, 3264) == 3927


def test_thread_sharded_entry_splits_frames_like_the_process_form(pkg):
    """ethcnn_shard_range (the split ethcnn_predict_yuv_file_sharded hands its worker threads) == sharding.frame_range (the split of the
    process-per-GPU form, world-size-2 gloo test): the two sharded forms cut a file at the same frames, so their outputs can only be
    byte-identical (GPU: tests/test_gpu_cli.py).  Pure host arithmetic."""
    import ctypes
    lib = pkg.load_library()
    for nframes in (0, 1, 7, 8, 9, 50, 425, 10 ** 9 + 7):
        for world in (1, 2, 3, 8, 64):
            for k in range(world):
                a, b = ctypes.c_int64(), ctypes.c_int64()
                assert lib.ethcnn_shard_range(nframes, world, k, ctypes.byref(a), ctypes.byref(b)) == 0
                assert (a.value, b.value) == pkg.sharding.frame_range(nframes, world, k)
    a, b = ctypes.c_int64(), ctypes.c_int64()
    assert lib.ethcnn_shard_range(10, 0, 0, ctypes.byref(a), ctypes.byref(b)) != 0
    assert lib.ethcnn_shard_range(10, 2, 2, ctypes.byref(a), ctypes.byref(b)) != 0


def test_frame_ranges_partition(pkg):
    fr = pkg.sharding.frame_range
    for F in (0, 1, 7, 50, 425):
        for G in (1, 2, 3, 4, 8):
            r = [fr(F, G, g) for g in range(G)]
            assert r[0][0] == 0 and r[-1][1] == F
            assert all(a[1] == b[0] for a, b in zip(r, r[1:]))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1
    assert [fr(425, 8, g) for g in (0, 7)] == [(0, 53), (371, 425)]
    with pytest.raises(ValueError):
        fr(10, 2, 2)


def test_cli_exit_status_contract(tmp_path):
    """HM asserts system(cmd) == 0 (TAppEncCfg.cpp:2321): every failure must be a non-zero
    exit with no cu_depth.dat left behind.  (No GPU here / no Thr_info.txt / bad argv.)"""
    launcher = os.path.join(ROOT, "video_to_cu_depth.py")
    yuv = tmp_path / "x.yuv"
    yuv.write_bytes(bytes(64 * 64 * 3 // 2))
    for argv in (["x.yuv", "64", "64", "32"], ["x.yuv", "64", "64"], ["x.yuv", "sixty", "64", "32"]):
        r = subprocess.run([sys.executable, launcher] + argv, cwd=str(tmp_path), capture_output=True)
        assert r.returncode != 0
        assert not (tmp_path / "cu_depth.dat").exists()


def test_native_c_tool_builds_and_fails_loudly(pkg, tmp_path):
    """include/ethcnn.h is consumable from strict C99 (the tool is compiled with gcc -std=c99
    -pedantic -Werror by the csrc Makefile); without arguments / without a device it exits 1."""
    import subprocess
    tool = os.path.join(ROOT, "hevc-complexity-reduction_amd", "bin", "video_to_cu_depth")
    assert os.path.exists(tool), "run __graft_entry__.build()"
    r = subprocess.run([tool], capture_output=True, text=True, cwd=str(tmp_path))
    assert r.returncode == 1 and "usage" in r.stderr
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if not has_gpu:
        r = subprocess.run([tool, "x.yuv", "64", "64", "32"], capture_output=True, text=True, cwd=str(tmp_path))
        assert r.returncode == 1 and "no CPU fallback" in r.stderr


def test_host_thread_budget_is_per_node_not_per_worker(pkg):
    """one predictor process per GPU, each with a staging-fill pool.  The node's CPU budget is DIVIDED by the
    local worker count: 8 workers under the GPU boxes' 16-core quota start <= 16 fill threads in total (they started
    8 x 16 = 128 before), every worker keeps at least one, and a lone worker never takes more than 16."""
    b = pkg.ethcnn.host_thread_budget
    for usable in (1, 2, 8, 16, 24, 64, 256):
        for workers in (1, 2, 3, 4, 8):
            per = b(workers, usable)
            assert 1 <= per <= 16
            assert workers * per <= max(usable, workers), (workers, usable, per)
    assert [b(w, 16) for w in (1, 2, 4, 8)] == [16, 8, 4, 2]
    assert b(8, 16) * 8 <= 16
    assert b(1, 256) == 16 and b(8, 256) == 16
    assert b(1) >= 1  # usable <= 0: probes this box (cgroup quota, logical CPUs / 2)


def test_shard_workers_are_told_how_many_share_the_node(pkg, monkeypatch):
    """predict_sharded hands every worker the number of workers it started (-> ETHCNN_LOCAL_WORKERS -> the budget above)."""
    import inspect
    src = inspect.getsource(pkg.video_to_cu_depth.predict_sharded)
    assert "len(ranges)" in src
    seen = {}

    class Fake(object):
        def __init__(self, device=0):
            seen["workers"] = os.environ.get("ETHCNN_LOCAL_WORKERS")
            raise RuntimeError("stop here")

    monkeypatch.setattr(pkg.video_to_cu_depth._e, "EthCnn", Fake)
    monkeypatch.delenv("ETHCNN_LOCAL_WORKERS", raising=False)
    with pytest.raises(RuntimeError, match="stop here"):
        pkg.video_to_cu_depth._shard_worker(0, "x.yuv", 64, 64, 32, "o.dat", 0, 1, (0.5, 0.5), 8)
    assert seen["workers"] == "8"
    monkeypatch.delenv("ETHCNN_LOCAL_WORKERS", raising=False)


def test_max_ctus_per_pass_is_bounded_by_the_kernels_32_bit_offsets():
    """the kernels address a pass with 32-bit byte offsets; the bound is a compile-time contract."""
    spec = open(os.path.join(ROOT, "hevc-complexity-reduction_amd", "csrc", "ethcnn_spec.h")).read()
    m = re.search(r"constexpr int kMaxCtusPerPass = (\d+);", spec)
    assert m and int(m.group(1)) == 131072
    n = int(m.group(1))
    assert (n // 16) * 2688 * 16 * 4 < 2 ** 31 and n * 448 * 4 < 2 ** 31 and 2 * n < 2 ** 24
    api = open(os.path.join(ROOT, "hevc-complexity-reduction_amd", "csrc", "ethcnn_context.cpp")).read()
    assert "std::min(kMaxCtusPerPass" in api


def test_bench_refuses_without_gpu_or_with_mismatched_world(tmp_path):
    """bench.py never prints a line for GPUs that were not measured: no GPU -> non-zero; WORLD_SIZE != --gpus -> non-zero."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by tests/test_gpu_bench_contract.py")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True)
    assert r.returncode != 0 and not any(l.startswith("{") for l in r.stdout.splitlines())


def test_development_knobs_are_compiled_out_of_the_shipped_library():
    """A/B switches and forced-steal test modes are read from the environment by the EXPERIMENTS build only
    (csrc `make exp` -> lib_exp/libethcnn.so, ethcnn_spec.h::dev_env).  The shipped library does not even contain their names;
    it keeps the user-facing variables."""
    dev = [b"ETHCNN_SMALL_STEAL_TEST", b"ETHCNN_LSTM_STEAL_TEST", b"ETHCNN_FC1_VARIANT", b"ETHCNN_OVERLAP",
           b"ETHCNN_DONE_WORD", b"ETHCNN_LSTM_ONE_LAUNCH", b"ETHCNN_TILE_BLOCKS", b"ETHCNN_FILE_IO", b"ETHCNN_LDP_INPLACE", b"ETHCNN_SMALL_SHAPE",
           b"ETHCNN_TRUNK_BLOCKS_PER_CU", b"ETHCNN_SMALL_EXP"]
    keep = [b"ETHCNN_FC1_PLAN", b"ETHCNN_NUMA_BIND", b"ETHCNN_HOST_THREADS", b"ETHCNN_LOCAL_WORKERS"]
    shipped = open(os.path.join(ROOT, "hevc-complexity-reduction_amd", "lib", "libethcnn.so"), "rb").read()
    for k in dev:
        assert k + b"\0" not in shipped, k
    for k in keep:
        assert k + b"\0" in shipped, k
    exp = os.path.join(ROOT, "hevc-complexity-reduction_amd", "lib_exp", "libethcnn.so")
    if not os.path.exists(exp):
        import __graft_entry__ as ge
        ge.build()
    blob = open(exp, "rb").read()
    for k in dev:
        assert k + b"\0" in blob, k


def test_hook_row_conversion_sse2_equals_scalar(tmp_path):
    """tools/hm_inprocess_hook.c converts HM's 16-bit picture to 8 bits sixteen samples per step (shift, then the pack instruction's
    saturation as the clamp): equal to the scalar `v >> shift` clamped to [0, 255] for EVERY 16-bit value, shifts 0..8, and row
    widths that leave a scalar tail.  Compiled as the encoder build compiles it (gcc -std=c99 -O2) with the hook's source included."""
    import subprocess
    src = tmp_path / "conv.c"
    src.write_text("""
#include "%s"
int main(void) {
    static short row[65536 + 64];
    static unsigned char got[65536 + 64];
    int shift, w, i;
    for (i = 0; i < 65536 + 64; ++i) row[i] = (short)(i - 32768);
    for (shift = 0; shift <= 8; ++shift)
        for (w = 65536; w <= 65536 + 33; w += 11) {
            convert_row(row, got, w, shift);
            for (i = 0; i < w; ++i) {
                int v = row[i] >> shift;
                unsigned char want = (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
                if (got[i] != want) { printf("mismatch: sample %%d shift %%d width %%d: %%d != %%d\\n", row[i], shift, w, got[i], want); return 1; }
            }
        }
    puts("rows equal");
    return 0;
}
""" % os.path.join(ROOT, "tools", "hm_inprocess_hook.c"))
    lib = os.path.join(ROOT, "hevc-complexity-reduction_amd", "lib")
    exe = str(tmp_path / "conv")
    subprocess.check_call(["gcc", "-std=c99", "-O2", "-D_POSIX_C_SOURCE=200809L", "-I" + os.path.join(ROOT, "include"), str(src), "-o", exe,
                           "-L" + lib, "-lethcnn", "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib"])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "rows equal" in r.stdout, r.stdout[-400:] + r.stderr[-400:]


def test_fast_plan_a_priori_bound_is_host_only_and_sees_looseness(oracle):
    """ethcnn_fast_plan_bound (no context, no device): the rigorous worst case of what the fp16 floors of plans 2 / 3 can move a
    probability by, from the weights alone.  It accepts plan 2 on well-conditioned weights on its own (<= 2.5e-5: nothing has to be
    measured), grows with the head gain, is far larger for plan 3 (more split layers behind each other) and explodes when one conv weight
    per tensor is an outlier (the guaranteed |feature| bound -- and with it the activation scale -- moves by the same factor)."""
    import importlib
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import adversarial_blobs as ab
    e = importlib.import_module("hevc-complexity-reduction_amd").ethcnn
    ok2, b2, f2 = e.fast_plan_bound(oracle.synth_blob(1, 1.0), 2)
    ok3, b3, f3 = e.fast_plan_bound(oracle.synth_blob(1, 1.0), 3)
    assert ok2 and 0.0 < b2 <= 2.5e-5 and not ok3 and b3 > 100 * b2 and f2 == f3 and 50.0 < f2 < 2000.0
    _, b2g, _ = e.fast_plan_bound(oracle.synth_blob(1, 8.0), 2)
    assert b2g > 10 * b2
    okx, bx, fx = e.fast_plan_bound(ab.outlier_in(oracle, 1, 1.0, 1e3, ("Variable",)), 2)
    assert not okx and bx > 100 * b2 and fx > 100 * f2
    with pytest.raises(e.EthCnnError):
        e.fast_plan_bound(oracle.synth_blob(1, 1.0), 1)
    with pytest.raises(e.EthCnnError):
        e.fast_plan_bound(np.zeros(10, np.float32), 2)
