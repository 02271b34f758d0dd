"""CSR pattern of the CG2 x CG1 stiffness matrix and the destination-sorted contribution map used by
``k_csr_segmented`` (host side, mesh only)."""
import numpy as np


def build_csr_map(mesh):
    cd = mesh.cell_dofs().astype(np.int64)                       # (nel, ld)
    nel, ld = cd.shape
    n = mesh.ndof
    rows = np.repeat(cd, ld, axis=1).ravel()                     # contribution k = e*ld*ld + i*ld + j -> (cd[e,i], cd[e,j])
    cols = np.tile(cd, (1, ld)).ravel()
    key = rows * n + cols
    perm = np.argsort(key, kind="stable")
    ks = key[perm]
    head = np.ones(ks.size, dtype=bool)
    head[1:] = ks[1:] != ks[:-1]
    dest = np.cumsum(head) - 1
    ukey = ks[head]
    nnz = int(ukey.size)
    urow = ukey // n
    rowptr = np.zeros(n + 1, dtype=np.int64)
    np.add.at(rowptr, urow + 1, 1)
    rowptr = np.cumsum(rowptr)
    return dict(nnz=nnz, rowptr=rowptr.astype(np.int32), colidx=(ukey % n).astype(np.int32),
                perm=np.ascontiguousarray(perm.astype(np.int32)), dest=np.ascontiguousarray(dest.astype(np.int32)))
