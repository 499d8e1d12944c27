import sys, os, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
os.chdir('/root/repo')
from conftest import load_molecules, random_weights
import tarfile, tempfile
d = tempfile.mkdtemp(); tarfile.open('tests/golden/mixed_val.tar.gz').extractall(d); vd = os.path.join(d, 'mixed_val')
names = [str(n) for n in np.load('tests/golden/val_names.npy', allow_pickle=True)]
from epnn_amd.engine import Engine
from oracle import epnn_oracle as orc
nx,T,N = 9,5,41
w = random_weights(nx, T, seed=nx+T, scale=0.35)
eng = Engine(nx=nx, T=T); eng.set_weights(w)
sel = [nm for nm in names if nm.startswith("dsgdb9nsd")][:24] + names[:8]
mols, offsets, xyz, x, Q = load_molecules(vd, sel, nx)
keep=[k for k,m in enumerate(mols) if m[1].shape[0] <= 32]; mols=[mols[k] for k in keep]
off=np.zeros(len(mols)+1,dtype=np.int32); off[1:]=np.cumsum([m[1].shape[0] for m in mols])
q = eng.forward_xyz(off, np.concatenate([m[0] for m in mols]), np.concatenate([m[1] for m in mols]), np.array([m[2] for m in mols],dtype=np.float32), N=N)
for k,m in enumerate(mols):
    r = orc.forward_xyz(m[0], m[1], m[2], w, N=N, dtype=np.float64)
    n=m[1].shape[0]
    err=np.abs(q[off[k]:off[k+1]]-r[:n])
    print(k, 'n=',n, 'err=%.2e'%err.max(), 'bad atoms', np.flatnonzero(err>1e-5)[:10])
