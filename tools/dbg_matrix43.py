import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import biem_helmholtz_sphere_amd as amd
from oracle import biem_oracle as O
_dev = lambda a, dt=torch.float64: torch.as_tensor(np.array(a), device="cuda").to(dt).contiguous()
tree, n_end, k, eta = "ba", int(sys.argv[1]) if len(sys.argv) > 1 else 43, 1.3, 0.7
tr = O.tree(tree)
cen = np.array([[0.0, 1.3, 0.2], [0.3, -1.4, -0.1]]); rad = np.array([1.0, 0.8])
alpha, beta = 1.0 + 0.25j, 0.4 - 0.1j
c = amd.create_from_branching_types(tree)
calc = amd.biem(c, centers=_dev(cen), radii=_dev(rad), k=_dev(k), eta=_dev(eta), n_end=n_end, alpha=alpha, beta=beta)
M = calc.matrix.cpu().numpy()
A, _ = O.assemble(tr, n_end, k, eta, cen, rad, np.full(2, alpha), np.full(2, beta))
nz = np.abs(A) > 1e-200
rel = np.where(nz, np.abs(M - A) / np.maximum(np.abs(A), 1e-300), 0)
print("max rel", rel.max(), "count > 5e-11:", (rel > 5e-11).sum(), "of", nz.sum())
deg = tr.degrees(n_end)
idx = np.argwhere(rel > 5e-11)
for b, h, bp, hp in idx[:: max(1, len(idx) // 12)][:12]:
    print((b, h, bp, hp), "deg", deg[h], deg[hp], "M", M[b, h, bp, hp], "A", A[b, h, bp, hp], "rel", rel[b, h, bp, hp])
# normwise per block
for b in range(2):
    for bp in range(2):
        print("block", b, bp, "normwise", np.abs(M[b, :, bp] - A[b, :, bp]).max() / np.abs(A[b, :, bp]).max())
if len(idx):
    hs = deg[idx[:, 1]]; hps = deg[idx[:, 3]]
    print("failing entries: min/max row degree", hs.min(), hs.max(), "col degree", hps.min(), hps.max())
