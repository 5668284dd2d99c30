import os, sys
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu)")
    config.addinivalue_line("markers", "ref: test needs oracle/_ref/libezref.so (reference build)")


# Host-only assertions (grid table, set-up math, options: SURVEY 8 rows a2 - a7, a24) need no device, but the round-end driver only runs `-m gpu` on the GPU box:
# a test decorated with @both_legs is collected twice -- once unmarked (the CPU suite here) and once marked gpu (the driver's suite there) -- and takes `leg`.
both_legs = pytest.mark.parametrize("leg", ["cpu_suite", pytest.param("gpu_suite", marks=pytest.mark.gpu)])
