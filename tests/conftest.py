import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """One line per golden fixture: how many gradient tensors passed the primary 2e-4 criterion, how many needed the
    fp64/perturbation band, and the worst errors observed (tests/golden_util.py::AUDIT)."""
    try:
        from tests import golden_util as gu
    except Exception:
        return
    lines = gu.audit_lines()
    if not lines:
        return
    terminalreporter.write_sep('=', 'parity audit (golden fixtures generated from the reference)')
    for ln in lines:
        terminalreporter.write_line(ln)
    out = os.path.join(ROOT, 'gpurun_out')
    if os.path.isdir(out):
        import json
        with open(os.path.join(out, 'parity_audit.json'), 'w') as f:
            json.dump({k: {kk: vv for kk, vv in v.items()} for k, v in gu.AUDIT.items()}, f, indent=1, default=str)
