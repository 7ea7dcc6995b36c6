"""launch helper: run tests/_mp_worker.py as `world` processes over gloo on 127.0.0.1 and load their results"""
import json
import os
import socket
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run_world(case, args=None, world=2, timeout=600, extra_env=None):
    tmp = tempfile.mkdtemp(prefix="nla_mp_")
    out = os.path.join(tmp, "res")
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   GLOO_SOCKET_IFNAME="lo", HSA_ENABLE_IPC_MODE_LEGACY="0", **(extra_env or {}))
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "_mp_worker.py"), case, out, json.dumps(args or {})],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o.decode(errors="replace"))
    for r, p in enumerate(procs):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, logs[r][-4000:])
    res = [dict(np.load(out + ".rank%d.npz" % r)) for r in range(world)]
    shutil.rmtree(tmp, ignore_errors=True)
    return res
