import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(autouse=True)
def _library_knobs_back_to_defaults():
    """Run-time knobs (ps_set_option) are process-wide: a test that switches a kernel family off must not decide which
    kernel the next test of the same worker runs."""
    yield
    import sys
    mod = sys.modules.get("probly_search_amd")
    if mod is None:
        return
    try:
        L = mod.load()
    except Exception:  # noqa: BLE001 (library not built: nothing to reset)
        return
    for name, v in ((b"PS_DAAT", 1), (b"PS_DAAT_MULTI", 1), (b"PS_DAAT_Z", 1), (b"PS_DAAT_Z_SPLIT", 1), (b"PS_DEVICE_PLAN", 1), (b"PS_WORK_COUNTERS", 1),
                    (b"PS_KERNEL_TIMERS", 1), (b"PS_DAAT_CHUNK", 4096), (b"PS_DAAT_SPLIT", 1), (b"PS_SCORE_ALT", 1), (b"PS_DAAT_SMALL_NL", 1)):
        L.ps_set_option(name, v)
