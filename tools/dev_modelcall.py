import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from epnn_amd import checkpoint, synth, charge_gn
w = checkpoint.load_epnn_weights(os.path.join(ROOT, "models/decay_model_weights"))
rng = np.random.default_rng(2)
for n, N in ((18, 29), (29, 29), (38, 41)):
    span = 1.6 * n ** (1 / 3.0) * 1.3
    while True:
        pts = rng.uniform(0, span, size=(n, 3))
        d = np.linalg.norm(pts[:, None] - pts[None], axis=-1) + np.eye(n) * 10
        if d.min() > 0.8: break
    h = np.zeros((1, N, N, 48), np.float32); e = np.zeros((1, N, N, 48), np.float32)
    x = np.zeros((1, N, N, 9), np.float32); q = np.zeros((1, N, N, 1), np.float32); m = np.zeros((1, N, N, 1), np.float32)
    e[0, :n, :n] = charge_gn.get_init_edges(pts.astype(np.float32), np.array([]), num=48)[0]
    x[0, :n, :n] = synth.features(rng.choice(["H", "C", "N", "O"], size=n))[None]
    m[0, :n, :n, 0] = 1
    model = charge_gn.make_model([32, 32], 48, 5, 9, N); model.set_weights_dict(w)
    for _ in range(5): model([h, e, x, q, m])
    t0 = time.perf_counter()
    for _ in range(200): model([h, e, x, q, m])
    dt = (time.perf_counter() - t0) / 200
    eng = model.engine()
    d = [eng.to_device(a) for a in (h, e, x, q, m)]; out = eng.alloc(N * 4)
    fn = eng.lib.epnn_model_forward_dense_dev
    for _ in range(5): fn(eng.h, 1, N, d[0].ptr, d[1].ptr, d[2].ptr, d[3].ptr, d[4].ptr, out.ptr)
    eng.sync(); t0 = time.perf_counter()
    for _ in range(200):
        fn(eng.h, 1, N, d[0].ptr, d[1].ptr, d[2].ptr, d[3].ptr, d[4].ptr, out.ptr); eng.sync()
    dd = (time.perf_counter() - t0) / 200
    print(f"n={n} N={N}: model([h,e,x,q,mask]) with host arrays {dt*1e3:.3f} ms per call; the same with the tensors resident in HBM {dd*1e3:.3f} ms", flush=True)
