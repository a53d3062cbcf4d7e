"""Host facts for the measurement and test harness (not on the compute path)."""
import os


def usable_cores():
    """CPU cores this process may really use: min(affinity mask, cgroup v2/v1 CPU quota).
    (The GPU boxes report 256 hardware threads through nproc but run under a 16-CPU quota; sizing an
    OpenMP team from os.cpu_count() there oversubscribes 16x and gets throttled.)"""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, p = f.read().split()[:2]
            if q != "max":
                quota = float(q) / float(p)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                q = float(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                p = float(f.read())
            if q > 0:
                quota = q / p
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = min(n, max(1, int(quota + 0.5)))
    return max(1, n)


def pin_openmp_team():
    """Default OMP_NUM_THREADS to usable_cores() (before any OpenMP library is loaded)."""
    os.environ.setdefault("OMP_NUM_THREADS", str(usable_cores()))
    return int(os.environ["OMP_NUM_THREADS"])
