"""pytest configuration: registers the `gpu` marker, puts the repo root on sys.path and makes sure the
in-tree HIP library exists (it is a build artefact, not a tracked file: a fresh checkout compiles it here
with hipcc, which cross-compiles for gfx950 without a GPU)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def pytest_sessionstart(session):
    import __graft_entry__ as entry
    if not os.path.exists(entry.LIB):
        entry.build_native(verbose=False)
