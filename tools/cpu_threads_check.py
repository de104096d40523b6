import os, time, sys
sys.path.insert(0, '.')
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try: print(f, open(f).read().strip())
    except Exception as e: print(f, "n/a")
import numpy as np
from oracle import c_oracle
rng = np.random.default_rng(0)
X = rng.standard_normal((3706, 128)).astype(np.float32); X /= np.linalg.norm(X, axis=1, keepdims=True)
Q = rng.standard_normal((2048, 128)); Q /= np.linalg.norm(Q, axis=1, keepdims=True)
for th in (1, 8, 16, 32, 64, 128, 256):
    n = 64 if th == 1 else 2048
    c_oracle.find_closest_f64(X, Q[:n], Q[:n], 200, threads=th)
    t = time.perf_counter(); c_oracle.find_closest_f64(X, Q[:n], Q[:n], 200, threads=th); dt = time.perf_counter() - t
    print(th, "threads:", n * 3706 / dt / 1e6, "M pairs/s")
