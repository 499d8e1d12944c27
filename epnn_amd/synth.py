"""Synthetic molecule batches of the shapes BASELINE.json names.

There is no network for datasets, so bench.py runs on molecules generated here:
  * QM9-like: atom counts from the empirical histogram of the 1338 QM9 files of the reference's `mixed` set,
    element mix H/C/O/N/F, coordinates by random sequential addition (min separation 0.95 A, each new atom
    within 1.0-1.6 A of an existing one) which gives ~8 partners within the 3 A cutoff per atom like the real set.
  * periodic-like box: uniform atoms at density 0.1 / A^3 with min separation 0.9 A and the protein's element mix.
"""
from __future__ import annotations

import numpy as np

# infer.py:13-30 element table (nx = 9): Z + one-hot over H C N O F S Cl Br
ATOM_NUM_8 = {'H': 1, 'C': 6, 'N': 7, 'O': 8, 'F': 9, 'S': 16, 'Cl': 17, 'Br': 35}
ELEM_8 = {'H': 0, 'C': 1, 'N': 2, 'O': 3, 'F': 4, 'S': 5, 'Cl': 6, 'Br': 7}

QM9_N_HIST = {7: 1, 9: 2, 10: 4, 11: 13, 12: 21, 13: 48, 14: 62, 15: 106, 16: 148, 17: 193, 18: 164, 19: 187,
              20: 122, 21: 124, 22: 51, 23: 66, 24: 5, 25: 18, 27: 2, 29: 1}
QM9_ELEMS = (("H", 0.5101), ("C", 0.3521), ("O", 0.0775), ("N", 0.0589), ("F", 0.0015))
PROTEIN_ELEMS = (("C", 0.318), ("H", 0.503), ("N", 0.090), ("O", 0.089), ("S", 0.001))


def features(symbols):
    """x rows = [Z, one-hot(8)] (charge_gn.py:325-327 with infer.py's table)."""
    x = np.zeros((len(symbols), 9), dtype=np.float32)
    for k, s in enumerate(symbols):
        x[k, 0] = ATOM_NUM_8[s]
        x[k, 1 + ELEM_8[s]] = 1
    return x


def _grow_molecule(rng, n, min_sep=0.95, lo=1.0, hi=1.6, max_links=2):
    """Random sequential addition: each new atom sits 1.0-1.6 A from an existing atom that has fewer than
    `max_links` attachments yet, and at least `min_sep` from every atom.  max_links=2 (chain-like growth) is
    calibrated to the real QM9 subset of the reference's `mixed` set: ~8 partners within 3 A per atom, near
    fraction ~0.45 of all ordered pairs (unrestricted attachment gives compact blobs with 10.6 partners)."""
    pts = np.zeros((n, 3))
    links = np.zeros(n, dtype=np.int64)
    k = 1
    while k < n:
        free = np.flatnonzero(links[:k] < max_links)
        bi = int(free[rng.integers(len(free))]) if len(free) else int(rng.integers(k))
        v = rng.normal(size=3)
        v /= np.linalg.norm(v)
        cand = pts[bi] + v * rng.uniform(lo, hi)
        if np.min(np.linalg.norm(pts[:k] - cand, axis=1)) >= min_sep:
            pts[k] = cand
            links[bi] += 1
            links[k] += 1
            k += 1
    return pts


def qm9_like_batch(B=1024, seed=0, N=29):
    """Returns offsets (B+1,), xyz (A,3) f32, x (A,9) f32, Q (B,) f32, N."""
    rng = np.random.default_rng(seed)
    sizes = np.array(sorted(QM9_N_HIST), dtype=np.int64)
    probs = np.array([QM9_N_HIST[s] for s in sizes], dtype=np.float64)
    probs /= probs.sum()
    ns = rng.choice(sizes, size=B, p=probs)
    ns[0] = min(N, 29)                      # the directory maximum is present, as in the real set
    names = [e for e, _ in QM9_ELEMS]
    ep = np.array([p for _, p in QM9_ELEMS])
    ep /= ep.sum()
    offsets = np.zeros(B + 1, dtype=np.int32)
    offsets[1:] = np.cumsum(ns)
    xyz = np.concatenate([_grow_molecule(rng, int(n)) for n in ns]).astype(np.float32)
    symbols = rng.choice(names, size=int(offsets[-1]), p=ep)
    return offsets, xyz, features(symbols), np.zeros(B, dtype=np.float32), N


def box_system(n_atoms=100_000, seed=0, density=0.1, min_sep=0.9):
    """One large non-periodic system: uniform atoms in a cube, minimum separation enforced with a cell grid."""
    rng = np.random.default_rng(seed)
    side = (n_atoms / density) ** (1.0 / 3.0)
    cell = min_sep
    ncell = int(np.ceil(side / cell))
    grid = {}
    pts = np.empty((n_atoms, 3))
    k = 0
    while k < n_atoms:
        cand = rng.uniform(0, side, size=(4096, 3))
        for p in cand:
            key = tuple((p // cell).astype(int))
            ok = True
            for dx in (-1, 0, 1):
                for dy in (-1, 0, 1):
                    for dz in (-1, 0, 1):
                        for q in grid.get((key[0] + dx, key[1] + dy, key[2] + dz), ()):
                            if np.sum((pts[q] - p) ** 2) < min_sep * min_sep:
                                ok = False
                                break
                        if not ok:
                            break
                    if not ok:
                        break
                if not ok:
                    break
            if ok:
                pts[k] = p
                grid.setdefault(key, []).append(k)
                k += 1
                if k == n_atoms:
                    break
    names = [e for e, _ in PROTEIN_ELEMS]
    ep = np.array([p for _, p in PROTEIN_ELEMS])
    ep /= ep.sum()
    symbols = rng.choice(names, size=n_atoms, p=ep)
    offsets = np.array([0, n_atoms], dtype=np.int32)
    return offsets, pts.astype(np.float32), features(symbols), np.zeros(1, dtype=np.float32), n_atoms


def algorithmic_flops(ns, near_unordered_pairs, nx=9, T=5, E=48, H=32, parts=False, chains_bf16=False, edges_bf16=True):
    """Forward flop count of the factorised exact algorithm (flop = 2*MAC).

    ns: atom count per molecule; near_unordered_pairs: total number of unordered pairs with D < cutoff."""
    ns = np.asarray(ns, dtype=np.float64)
    F = nx + 49
    nnz = 2.0 * float(near_unordered_pairs)
    n1, n2 = ns.sum(), (ns * ns).sum()
    gnn = n1 * 2 * F * H + nnz * E * H + n2 * H * H + n1 * H * H + n1 * H * H + n1 * (80 * H + H * H + H * 48)
    epn = n1 * 2 * F * H + (nnz / 2) * (E * H + 2 * H * H + 2 * H)
    if parts == "pipes":
        # (f32-MFMA flops, flops on the bf16 matrix pipe as six bf16 products of exact three-piece splits per f32-grade product --
        #  DESIGN.md section 4): the pair MLPs' second Dense always; with chains_bf16 (the fused kernels) also the per-atom chains --
        #  the first Dense's atom blocks, the update MLP -- and, since round 5's last change (edges_bf16), the edge products G = We^T e
        #  of the in-kernel front-end, which leaves the EPN read-out (32 MACs per pair and direction: vector instructions, priced
        #  with the f32 part); the tiled kernels keep their per-atom chains and correction tiles on the f32 pipe
        bf = n2 * H * H + (nnz / 2) * 2 * H * H
        if chains_bf16:
            bf += 2 * (n1 * 2 * F * H) + n1 * H * H + n1 * H * H + n1 * (80 * H + H * H + H * 48)
            if edges_bf16:      # ... and the edge products of both stacks (K = 16 in the kernel's own basis, counted with all E channels)
                bf += nnz * E * H + (nnz / 2) * E * H
        return 2.0 * T * (gnn + epn - bf), 2.0 * T * bf
    if parts:
        return 2.0 * T * gnn, 2.0 * T * epn
    return 2.0 * T * (gnn + epn)


# Dense matrix-pipe peaks of MI355X (MI355X_MICROARCH.md): v_mfma_f32_16x16x4_f32 / 32x32x2; v_mfma_f32_16x16x32_bf16 / 32x32x16
FP32_MFMA_PEAK_TFLOPS = 157.3
BF16_MFMA_PEAK_TFLOPS = 2500.0
BF16X6_PEAK_TFLOPS = BF16_MFMA_PEAK_TFLOPS / 6.0     # an f32-grade product = six bf16 products


def mixed_pipe_peak(ns, near_unordered_pairs, nx=9, T=5, chains_bf16=False, edges_bf16=True):
    """The matrix pipes' bound on the forward, in ALGORITHMIC TFLOP/s: the flops that run as f32 MFMAs at 157.3 TFLOP/s, those on
    the bf16 pipe at 2500 / 6 (six bf16 MFMAs per f32-grade product): peak = total / (f32 part / 157.3 + bf16 part / 416.7).
    chains_bf16: the fused kernels (per-atom chains and, with edges_bf16, the edge products on the bf16 pipe too); False: the tiled
    kernels (the all-pairs Dense only).  Returns (peak, share of the algorithmic flops on the bf16 pipe)."""
    f32, d2 = algorithmic_flops(ns, near_unordered_pairs, nx=nx, T=T, parts="pipes", chains_bf16=chains_bf16, edges_bf16=edges_bf16)
    return (f32 + d2) / (f32 / FP32_MFMA_PEAK_TFLOPS + d2 / BF16X6_PEAK_TFLOPS), d2 / (f32 + d2)
