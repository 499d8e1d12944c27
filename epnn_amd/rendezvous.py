"""Rank rendezvous for the multi-process entry points (train.py, tools) without torch: a tiny TCP key-value store.

The data path of this library has exactly one collective -- the RCCL all-reduce of the flat gradient inside
``epnn_train_apply`` (``epnn_comm_init``).  All it needs from the launcher is that the 128-byte RCCL unique id created
on rank 0 reaches the other ranks; the training script additionally gathers a few small arrays (metrics, predictions)
for the artefacts rank 0 writes.  Rank 0 serves a dictionary on ``MASTER_ADDR:EPNN_RDZV_PORT`` (default MASTER_PORT + 17:
under ``torch.distributed.run`` MASTER_PORT itself belongs to torch's store); every rank, rank 0 included, is a client.

Protocol: one JSON object per line.  ``{"op": "set", "key": k, "val": <encoded>}`` -> ``{"ok": true}``;
``{"op": "get", "key": k, "timeout": s, "readers": n}`` blocks on the server until the key exists -> ``{"ok": true,
"val": ...}`` or ``{"ok": false}``; a key is dropped from the store once ``n`` gets have fetched it (every collective
knows how many ranks read each of its keys), so a long training run does not accumulate its per-step gathers on rank 0.
Failures are loud: a key that never arrives raises ``RendezvousError`` after the timeout, so a rank whose peer died
exits non-zero instead of waiting for ever.

Values are DATA, never code: bytes, NumPy arrays (``np.save`` / ``np.load(allow_pickle=False)``), numbers, strings,
None and lists / tuples / string-keyed dicts of those, as tagged JSON -- nothing a peer sends is ever unpickled.  When
the launcher passes a secret (``EPNN_RDZV_SECRET``; ``launch_ranks`` always does) every request carries an HMAC-SHA256 of
its key and value and the server drops requests without a valid one; the server binds to loopback when MASTER_ADDR is a
loopback address.
"""
from __future__ import annotations

import base64
import hashlib
import hmac
import io
import json
import os
import socket
import socketserver
import threading
import time


class RendezvousError(RuntimeError):
    pass


def encode(obj):
    """Python value -> JSON-able tree (see the module docstring for what may travel)."""
    import numpy as np
    if obj is None or isinstance(obj, (bool, int, float, str)):
        return obj
    if isinstance(obj, (bytes, bytearray)):
        return {"__b": base64.b64encode(bytes(obj)).decode()}
    if isinstance(obj, np.generic):
        return obj.item()
    if isinstance(obj, np.ndarray):
        if obj.dtype == object:
            raise TypeError("rendezvous: object arrays do not travel")
        buf = io.BytesIO()
        np.save(buf, obj, allow_pickle=False)
        return {"__nd": base64.b64encode(buf.getvalue()).decode()}
    if isinstance(obj, tuple):
        return {"__t": [encode(v) for v in obj]}
    if isinstance(obj, list):
        return [encode(v) for v in obj]
    if isinstance(obj, dict):
        if not all(isinstance(k, str) for k in obj):
            raise TypeError("rendezvous: dict keys must be strings")
        return {"__d": {k: encode(v) for k, v in obj.items()}}
    raise TypeError(f"rendezvous: a {type(obj).__name__} does not travel (bytes, arrays, numbers, strings, lists, tuples, dicts)")


def decode(tree):
    import numpy as np
    if isinstance(tree, list):
        return [decode(v) for v in tree]
    if isinstance(tree, dict):
        if "__b" in tree:
            return base64.b64decode(tree["__b"])
        if "__nd" in tree:
            return np.load(io.BytesIO(base64.b64decode(tree["__nd"])), allow_pickle=False)
        if "__t" in tree:
            return tuple(decode(v) for v in tree["__t"])
        if "__d" in tree:
            return {k: decode(v) for k, v in tree["__d"].items()}
        raise RendezvousError("rendezvous: malformed value")
    return tree


def _mac(secret, key, val):
    return hmac.new(secret.encode(), (key + "\n" + json.dumps(val, sort_keys=True)).encode(), hashlib.sha256).hexdigest()


class _Store:
    def __init__(self, secret=None):
        self.data = {}
        self.reads = {}
        self.cv = threading.Condition()
        self.secret = secret


class _Handler(socketserver.StreamRequestHandler):
    def handle(self):
        store = self.server.store
        for line in self.rfile:
            try:
                req = json.loads(line)
            except ValueError:
                return
            key = req.get("key")
            if not isinstance(key, str):
                return
            if store.secret is not None and not hmac.compare_digest(str(req.get("mac", "")), _mac(store.secret, key, req.get("val"))):
                return                                            # not one of this job's ranks: drop the connection
            if req.get("op") == "set":
                with store.cv:
                    store.data[key] = req["val"]
                    store.reads[key] = 0
                    store.cv.notify_all()
                rep = {"ok": True}
            elif req.get("op") == "get":
                deadline = time.monotonic() + float(req.get("timeout", 60.0))
                with store.cv:
                    while key not in store.data and time.monotonic() < deadline:
                        store.cv.wait(timeout=max(0.0, min(1.0, deadline - time.monotonic())))
                    found = key in store.data
                    val = store.data.get(key)
                    if found:
                        store.reads[key] += 1
                        if 0 < int(req.get("readers", 0)) <= store.reads[key]:     # its last reader: forget it
                            del store.data[key], store.reads[key]
                rep = {"ok": found, "val": val}
            elif req.get("op") == "size":
                with store.cv:
                    rep = {"ok": True, "val": len(store.data)}
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
        self._secret = os.environ.get("EPNN_RDZV_SECRET") or None
        if self.rank == 0:
            self._server = _Server((self.addr, self.port), _Handler)
            self._server.store = _Store(self._secret)
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
        if self._secret is not None:
            req["mac"] = _mac(self._secret, req["key"], req.get("val"))
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
        self._call({"op": "set", "key": key, "val": encode(obj)})

    def get(self, key, timeout=None, readers=0):
        """`readers` > 0: the store forgets the key once that many gets have fetched it."""
        rep = self._call({"op": "get", "key": key, "timeout": self.timeout if timeout is None else timeout, "readers": int(readers)})
        if not rep.get("ok"):
            raise RendezvousError(f"rank {self.rank}: '{key}' did not arrive within the timeout (a peer failed?)")
        return decode(rep["val"])

    def store_size(self):
        """Number of keys the store holds (diagnostics / tests)."""
        return int(self._call({"op": "size", "key": ""})["val"])

    def _tag(self, name):
        """Collectives are matched by call order per name, like every collective API."""
        n = self._seq.get(name, 0)
        self._seq[name] = n + 1
        return f"{name}#{n}"

    def broadcast(self, obj, src=0, name="bcast"):
        tag = self._tag(name)
        if self.rank == src:
            if self.world > 1:
                self.set(tag, obj)
            return obj
        return self.get(tag, readers=self.world - 1)

    def all_gather(self, obj, name="gather"):
        tag = self._tag(name)
        self.set(f"{tag}/{self.rank}", obj)
        return [self.get(f"{tag}/{r}", readers=self.world) for r in range(self.world)]

    def all_reduce_max(self, values, name="max"):
        """element-wise maximum of a short list of floats over the ranks (host-side: timing exchanges)"""
        parts = self.all_gather([float(v) for v in values], name)
        return [max(p[k] for p in parts) for k in range(len(parts[0]))]

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


def launch_ranks(script, argv, n, timeout=None, grace=5.0):
    """Start `n` rank processes of `script` as fresh children of the calling (GPU-free) process -- RANK, LOCAL_RANK,
    WORLD_SIZE, MASTER_ADDR, MASTER_PORT, a fresh rendezvous port and secret in their environment --, pass rank 0's stdout
    through, and return the first non-zero exit code (0 if every rank succeeded).

    The ranks are watched together: as soon as one exits non-zero the others get `grace` seconds to finish by themselves
    (a peer blocked in an RCCL collective or in hipStreamSynchronize never will: RCCL has no timeout), then SIGTERM, then
    SIGKILL, so a failed rank ends the job instead of leaving its peers holding their GPUs.  `timeout` (seconds, or
    EPNN_LAUNCH_TIMEOUT) bounds the whole job the same way; exit code 124 when it strikes."""
    import secrets
    import subprocess
    import sys
    if timeout is None and os.environ.get("EPNN_LAUNCH_TIMEOUT"):
        timeout = float(os.environ["EPNN_LAUNCH_TIMEOUT"])
    with socket.socket() as s, socket.socket() as s2:
        s.bind(("127.0.0.1", 0))
        s2.bind(("127.0.0.1", 0))
        port, rport = s.getsockname()[1], s2.getsockname()[1]
    secret = secrets.token_hex(16)
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), EPNN_RDZV_PORT=str(rport), EPNN_RDZV_SECRET=secret)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(script)] + list(argv), env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))

    def stop_all():
        for p in procs:
            if p.poll() is None:
                p.terminate()
        t_end = time.monotonic() + 3.0
        for p in procs:
            try:
                p.wait(max(0.0, t_end - time.monotonic()))
            except subprocess.TimeoutExpired:
                p.kill()
        for p in procs:
            p.wait()

    t_start = time.monotonic()
    first_bad, t_bad = 0, None
    try:
        while True:
            rcs = [p.poll() for p in procs]
            if first_bad == 0:
                bad = [rc for rc in rcs if rc not in (None, 0)]
                if bad:
                    first_bad, t_bad = bad[0], time.monotonic()
            if all(rc is not None for rc in rcs):
                break
            if t_bad is not None and time.monotonic() - t_bad > grace:
                print(f"[launch_ranks] a rank exited with code {first_bad}; stopping the others", file=sys.stderr, flush=True)
                stop_all()
                break
            if timeout is not None and time.monotonic() - t_start > timeout:
                print(f"[launch_ranks] job exceeded {timeout:.0f} s; stopping every rank", file=sys.stderr, flush=True)
                stop_all()
                return first_bad or 124
            time.sleep(0.05)
    except BaseException:
        stop_all()
        raise
    return first_bad
