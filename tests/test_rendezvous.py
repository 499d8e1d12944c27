"""epnn_amd.rendezvous: the torch-free rank rendezvous train.py uses (RCCL id broadcast, metric gathers).  CPU only."""
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

from conftest import ROOT


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_broadcast_gather_barrier_between_three_ranks():
    from epnn_amd.rendezvous import Rendezvous
    port, world = _free_port(), 3
    out, errs = [None] * world, []

    def run(rank):
        try:
            r = Rendezvous(rank, world, "127.0.0.1", port, timeout=20)
            ident = r.broadcast(bytes(range(128)) if rank == 0 else None, name="rccl_id")
            g1 = r.all_gather(np.full(4, rank, np.float32), name="grad")
            g2 = r.all_gather({"rank": rank}, name="grad")             # second call with the same name: its own slot
            r.barrier()
            r.close()
            out[rank] = (ident, g1, g2)
        except Exception as exc:                                         # noqa: BLE001
            errs.append(exc)

    ts = [threading.Thread(target=run, args=(k,)) for k in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(60)
    assert not errs, errs
    for rank in range(world):
        ident, g1, g2 = out[rank]
        assert ident == bytes(range(128))
        assert [float(a[0]) for a in g1] == [0.0, 1.0, 2.0] and [d["rank"] for d in g2] == [0, 1, 2]


def test_missing_peer_is_an_error_not_a_hang():
    from epnn_amd.rendezvous import Rendezvous, RendezvousError
    r = Rendezvous(0, 2, "127.0.0.1", _free_port(), timeout=1.0)
    with pytest.raises(RendezvousError, match="did not arrive"):
        r.all_gather(1, name="x")                                        # rank 1 never shows up
    r._server.shutdown()
    with pytest.raises(RendezvousError, match="no rendezvous server"):
        Rendezvous(1, 2, "127.0.0.1", _free_port(), timeout=0.5)


_WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
from epnn_amd.rendezvous import Rendezvous
r = Rendezvous()
vals = r.all_gather(int(os.environ["RANK"]) * 10, name="v")
r.barrier(); r.close()
if r.rank == 0:
    print("GATHERED", vals, os.environ["LOCAL_RANK"], os.environ["WORLD_SIZE"])
if len(sys.argv) > 2 and r.rank == 1:
    sys.exit(7)
'''


def test_launch_ranks_starts_fresh_children_and_reports_failure(tmp_path):
    """What `bench.py --gpus N` / `train.py --gpus N` do without a launcher: N fresh child processes with the rank
    environment; rank 0's stdout passes through; a failing rank makes the launcher's exit code non-zero."""
    script = tmp_path / "w.py"
    script.write_text(_WORKER)
    drv = ("import sys; sys.path.insert(0, sys.argv[1]); from epnn_amd.rendezvous import launch_ranks; "
           "sys.exit(launch_ranks(sys.argv[2], sys.argv[3:], 2))")
    ok = subprocess.run([sys.executable, "-c", drv, ROOT, str(script), ROOT], capture_output=True, text=True, timeout=120)
    assert ok.returncode == 0, ok.stderr[-2000:]
    assert "GATHERED [0, 10] 0 2" in ok.stdout
    bad = subprocess.run([sys.executable, "-c", drv, ROOT, str(script), ROOT, "fail"], capture_output=True, text=True, timeout=120)
    assert bad.returncode == 7
