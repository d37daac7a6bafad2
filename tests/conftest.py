import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_addoption(parser):
    parser.addoption("--runslow", action="store_true", default=False,
                     help="also run the full parametric grids (marker `slow`): the host-side D-pass truths of the "
                          "64-wide MADE spline presets take 10-30 s each")


def pytest_collection_modifyitems(config, items):
    if config.getoption("--runslow"):
        return
    skip = pytest.mark.skip(reason="full grid: run with --runslow")
    for item in items:
        if "slow" in item.keywords:
            item.add_marker(skip)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: full parametric grids of the GPU suite (deselected unless --runslow)")


# One summary line at the end of a run (so that it lands in the tail the driver records): every golden fixture whose
# log_prob on the HIP path is further than 1e-5 from the reference's fp32 value, with the reference's own fp32-vs-fp64
# distance beside it (tests/test_gpu_flows.py admits those through the floor-relative bar).
PARITY_NOTES = []


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    if PARITY_NOTES:
        terminalreporter.write_line("parity: fixtures above 1e-5 on log_prob vs the reference's fp32 value "
                                    "(error / reference's own fp32-vs-fp64 floor): " + "; ".join(PARITY_NOTES))


def set_debug(monkeypatch, **switches) -> None:
    """Set (value) or clear (None) route / tuning switches inside TORCHFLOWS_AMD_DEBUG="key=value,..." (torchflows_amd/
    utils.py: debug_switch), keeping the switches other calls of the same test have set."""
    cur = {}
    for item in os.environ.get("TORCHFLOWS_AMD_DEBUG", "").split(","):
        k, _, v = item.partition("=")
        if k.strip():
            cur[k.strip().lower()] = v.strip()
    for k, v in switches.items():
        if v is None:
            cur.pop(k.lower(), None)
        else:
            cur[k.lower()] = str(v)
    if cur:
        monkeypatch.setenv("TORCHFLOWS_AMD_DEBUG", ",".join(f"{k}={v}" for k, v in cur.items()))
    else:
        monkeypatch.delenv("TORCHFLOWS_AMD_DEBUG", raising=False)


def free_port() -> str:
    """A TCP port that is free on 127.0.0.1 right now (rendezvous of the multi-process tests: no hard-coded ports)."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return str(s.getsockname()[1])


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def state_dict_of(fx, variant):
    pre = f"sd_{variant}/"
    return {k[len(pre):]: fx[k] for k in fx.files if k.startswith(pre)}


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure). Built on demand with gcc."""
    from oracle import oracle as orc
    orc.build()
    return orc
