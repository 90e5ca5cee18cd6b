import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def free_port():
    """A TCP port nobody listens on right now, for the rendezvous of a multi-process test (fixed port numbers collide when two suites
    share a host)."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pins():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "ref_pins.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (oracle/) -- checker only."""
    from oracle import oracle_ffi
    oracle_ffi.lib()
    return oracle_ffi
