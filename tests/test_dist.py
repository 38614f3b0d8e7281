"""CPU suite, part 3: the N > 1 path.  The collectives the library issues all go through one hook
(rsseg_allreduce_fn); this runs the Python side of that hook with 2 and 3 gloo ranks on CPU buffers and
checks the reduction protocol the C++ host code relies on."""
import os
import socket
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return str(p)


@pytest.mark.parametrize("world", [2, 3])
def test_allreduce_hook_protocol_gloo(world, tmp_path):
    port = _free_port()
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "_dist_worker.py"), "hook_cpu", str(r), str(world), port, str(tmp_path)])
             for r in range(world)]
    for p in procs:
        assert p.wait(timeout=180) == 0
    for r in range(world):
        assert os.path.exists(tmp_path / f"ok_{r}")
