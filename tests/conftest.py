import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "fortran-library_amd"))


# The parity suite was written against the geometries of n alone (fl_reduction_geometry / _for): it runs under the THROUGHPUT
# policy, whose results do not depend on the batch size.  tests/test_gpu_geometry.py covers the latency geometries and the
# process default FL_GEOMETRY_AUTO (options and fl_set_geometry_policy override this initial policy); __graft_entry__.smoke()
# and bench.py run under the default.  Set before libFL.so reads it (its first geometry decision).
os.environ.setdefault("FL_GEOMETRY", "throughput")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
