"""Rank rendezvous for the multi-process entry points (train.py, tools) without torch: a tiny TCP key-value store.

The data path of this library has exactly one collective -- the RCCL all-reduce of the flat gradient inside
``epnn_train_apply`` (``epnn_comm_init``).  All it needs from the launcher is that the 128-byte RCCL unique id created
on rank 0 reaches the other ranks; the training script additionally gathers a few small arrays (metrics, predictions)
for the artefacts rank 0 writes.  Rank 0 serves a dictionary on ``MASTER_ADDR:EPNN_RDZV_PORT`` (default MASTER_PORT + 17:
under ``torch.distributed.run`` MASTER_PORT itself belongs to torch's store); every rank, rank 0 included, is a client.

Protocol: one JSON object per line.  ``{"op": "set", "key": k, "val": <base64>}`` -> ``{"ok": true}``;
``{"op": "get", "key": k, "timeout": s}`` blocks on the server until the key exists -> ``{"ok": true, "val": ...}`` or
``{"ok": false}``.  Failures are loud: a key that never arrives raises ``RendezvousError`` after the timeout, so a rank whose
peer died exits non-zero instead of waiting for ever.
"""
from __future__ import annotations

import base64
import json
import os
import pickle
import socket
import socketserver
import threading
import time


class RendezvousError(RuntimeError):
    pass


class _Store:
    def __init__(self):
        self.data = {}
        self.cv = threading.Condition()


class _Handler(socketserver.StreamRequestHandler):
    def handle(self):
        store = self.server.store
        for line in self.rfile:
            try:
                req = json.loads(line)
            except ValueError:
                return
            if req.get("op") == "set":
                with store.cv:
                    store.data[req["key"]] = req["val"]
                    store.cv.notify_all()
                rep = {"ok": True}
            elif req.get("op") == "get":
                deadline = time.monotonic() + float(req.get("timeout", 60.0))
                with store.cv:
                    while req["key"] not in store.data and time.monotonic() < deadline:
                        store.cv.wait(timeout=max(0.0, min(1.0, deadline - time.monotonic())))
                    val = store.data.get(req["key"])
                rep = {"ok": val is not None, "val": val}
            else:
                rep = {"ok": False}
            self.wfile.write((json.dumps(rep) + "\n").encode())
            self.wfile.flush()


class _Server(socketserver.ThreadingTCPServer):
    allow_reuse_address = True
    daemon_threads = True


class Rendezvous:
    """One per rank.  ``rank``/``world``/address default to the launcher's environment (RANK, WORLD_SIZE, MASTER_ADDR,
    EPNN_RDZV_PORT or MASTER_PORT + 17)."""

    def __init__(self, rank=None, world=None, addr=None, port=None, timeout=120.0):
        self.rank = int(os.environ.get("RANK", "0")) if rank is None else int(rank)
        self.world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else int(world)
        self.addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
        if port is None:
            port = os.environ.get("EPNN_RDZV_PORT")
            port = int(port) if port else int(os.environ.get("MASTER_PORT", "29500")) + 17
        self.port = int(port)
        self.timeout = float(timeout)
        self._server = None
        self._seq = {}
        if self.rank == 0:
            self._server = _Server((self.addr, self.port), _Handler)
            self._server.store = _Store()
            threading.Thread(target=self._server.serve_forever, daemon=True).start()
        deadline = time.monotonic() + self.timeout
        while True:
            try:
                self._sock = socket.create_connection((self.addr, self.port), timeout=self.timeout)
                break
            except OSError as exc:
                if time.monotonic() > deadline:
                    raise RendezvousError(f"rank {self.rank}: no rendezvous server at {self.addr}:{self.port}: {exc}") from exc
                time.sleep(0.05)
        self._file = self._sock.makefile("rwb")

    def _call(self, req):
        self._sock.settimeout(float(req.get("timeout", self.timeout)) + 10.0)
        try:
            self._file.write((json.dumps(req) + "\n").encode())
            self._file.flush()
            line = self._file.readline()
        except OSError as exc:
            raise RendezvousError(f"rank {self.rank}: rendezvous connection failed: {exc}") from exc
        if not line:
            raise RendezvousError(f"rank {self.rank}: rendezvous server closed the connection")
        return json.loads(line)

    def set(self, key, obj):
        self._call({"op": "set", "key": key, "val": base64.b64encode(pickle.dumps(obj)).decode()})

    def get(self, key, timeout=None):
        rep = self._call({"op": "get", "key": key, "timeout": self.timeout if timeout is None else timeout})
        if not rep.get("ok"):
            raise RendezvousError(f"rank {self.rank}: '{key}' did not arrive within the timeout (a peer failed?)")
        return pickle.loads(base64.b64decode(rep["val"]))

    def _tag(self, name):
        """Collectives are matched by call order per name, like every collective API."""
        n = self._seq.get(name, 0)
        self._seq[name] = n + 1
        return f"{name}#{n}"

    def broadcast(self, obj, src=0, name="bcast"):
        tag = self._tag(name)
        if self.rank == src:
            self.set(tag, obj)
            return obj
        return self.get(tag)

    def all_gather(self, obj, name="gather"):
        tag = self._tag(name)
        self.set(f"{tag}/{self.rank}", obj)
        return [self.get(f"{tag}/{r}") for r in range(self.world)]

    def barrier(self, name="barrier"):
        self.all_gather(None, name)

    def close(self):
        """Rank 0 keeps serving until every rank has said goodbye (or the timeout passes)."""
        try:
            self.set(f"bye/{self.rank}", True)
            if self.rank == 0:
                for r in range(self.world):
                    try:
                        self.get(f"bye/{r}", timeout=min(self.timeout, 30.0))
                    except RendezvousError:
                        break
        finally:
            try:
                self._file.close()
                self._sock.close()
            except OSError:
                pass
            if self._server is not None:
                self._server.shutdown()
                self._server.server_close()
                self._server = None


def launch_ranks(script, argv, n):
    """Start `n` rank processes of `script` as fresh children of the calling (GPU-free) process -- RANK, LOCAL_RANK,
    WORLD_SIZE, MASTER_ADDR, MASTER_PORT in their environment --, pass rank 0's stdout through, and return the first
    non-zero exit code (0 if every rank succeeded)."""
    import subprocess
    import sys
    with socket.socket() as s, socket.socket() as s2:
        s.bind(("127.0.0.1", 0))
        s2.bind(("127.0.0.1", 0))
        port, rport = s.getsockname()[1], s2.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), EPNN_RDZV_PORT=str(rport))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(script)] + list(argv), env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rcs = [p.wait() for p in procs]
    bad = [rc for rc in rcs if rc != 0]
    return bad[0] if bad else 0
