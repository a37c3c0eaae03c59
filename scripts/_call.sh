cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us /sys/fs/cgroup/cpu/cpu.cfs_period_us 2>/dev/null; cat /proc/self/cgroup | head -3
python - <<'PY'
import os, time, sys
sys.path.insert(0, "oracle"); sys.path.insert(0, ".")
import ethcnn_np as o, bench
print("affinity cpus:", len(os.sched_getaffinity(0)))
luma = bench.synth_luma(3840, 2160, 8, 1); blob = o.synth_blob(1, 8.0)
for t in (1, 2, 4, 8, 16, 32, 64, 128, 256):
    o.set_threads(t); o.predict_frames(blob, luma, 3840, 2160, 8, 32)
    t0 = time.time(); o.predict_frames(blob, luma, 3840, 2160, 8, 32); dt = time.time() - t0
    print(t, "threads:", int(8 * 2040 / dt), "CTU/s")
PY
uptime
