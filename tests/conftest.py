import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


from muscle_amd.hostinfo import pin_openmp_team  # noqa: E402

pin_openmp_team()  # the oracle / compiled reference are OpenMP: size their team from the CPU quota


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "ref: needs oracle/_ref/libmuscle_ref.so (the compiled reference; "
                            "present only where /root/reference was available at build time)")
