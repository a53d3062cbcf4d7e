"""diag: run a command and report the kernel driver's queue-eviction time while it ran (KFD: /sys/class/kfd/kfd/proc/<pid>/stats_<gpu>/evicted_ms,
all processes the driver lists - the pid namespace of a container differs from the driver's). An eviction stops EVERY queue of a process; a
freed host buffer that the runtime had registered for DMA is one way to cause it.   usage: python diag/evicted.py <command...>"""
import glob, subprocess, sys, time


def sample():
    out = {}
    for f in glob.glob("/sys/class/kfd/kfd/proc/*/stats_*/evicted_ms"):
        try:
            out[f] = int(open(f).read().strip())
        except (OSError, ValueError):
            pass
    return out


before = set(f.split("/")[-3] for f in sample())
p = subprocess.Popen(sys.argv[1:])
seen = {}
while p.poll() is None:
    for f, v in sample().items():
        a = seen.setdefault(f, [v, v])
        a[1] = v
    time.sleep(0.02)
tot = {f.split("/")[-3] + "/" + f.split("/")[-2]: b for f, (a, b) in seen.items() if f.split("/")[-3] not in before and b}  # processes that started under this command
print("[evicted] %s: evicted_ms of the processes that started under it (KFD process / GPU): %s" % (" ".join(sys.argv[1:3]), tot), flush=True)
sys.exit(p.returncode)
