import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "ref: needs the reference build oracle/_ref (only in the build container)")


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """Which bar the GPU comparisons applied on this machine: printed with the result line."""
    try:
        from tests import util
        terminalreporter.write_line("parity bar on this host: " + util.TOLERANCE_BRANCH)
    except Exception:
        pass
