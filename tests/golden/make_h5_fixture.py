"""Writes the XDMF + HDF5 mesh fixtures of tests/test_mesh_io.py in the layout dolfinx's XDMFFile.write_mesh uses
(an XML file whose DataItems point at /Mesh/Grid/topology [int64] and /Mesh/Grid/geometry [float64] of an .h5 file) --
the format of every mesh the reference ships, e.g. examples/advanced_examples/simple_shell_opt/plate_meshes/
plate_2_10_quad_4_20.xdmf + .h5 (git-LFS stubs in the reference tree; regenerable from their names:
a 2 x 10 plate, 4 x 20 quads, ex_simple_shell_opt.py:27-30).

Needs h5py, which the image only has under /opt/conda:    /opt/conda/bin/python3.9 tests/golden/make_h5_fixture.py

Three HDF5 flavours of the same mesh, so that the library-free reader (femo_alpha_amd/hdf5_min.py) is exercised on every
structure it claims to parse:
  plate_2_10_quad_4_20.h5          libver earliest (what HDF5 1.10/1.12 + MPI-IO write by default): superblock 0,
                                   version 1 object headers, symbol-table groups, contiguous datasets
  plate_2_10_quad_4_20_chunked.h5  the same with chunked (unfiltered) datasets -> version 1 chunk B-trees
  plate_2_10_quad_4_20_latest.h5   libver latest: superblock 3, version 2 object headers, link messages
"""
import os

import h5py
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
XDMF = """<?xml version="1.0"?>
<!DOCTYPE Xdmf SYSTEM "Xdmf.dtd" []>
<Xdmf Version="3.0" xmlns:xi="http://www.w3.org/2001/XInclude">
  <Domain>
    <Grid Name="Grid" GridType="Uniform">
      <Topology TopologyType="Quadrilateral" NumberOfElements="{nel}" NodesPerElement="4">
        <DataItem Dimensions="{nel} 4" NumberType="Int" Format="HDF">{h5}:/Mesh/Grid/topology</DataItem>
      </Topology>
      <Geometry GeometryType="XYZ">
        <DataItem Dimensions="{nn} 3" Format="HDF">{h5}:/Mesh/Grid/geometry</DataItem>
      </Geometry>
    </Grid>
  </Domain>
</Xdmf>
"""


def plate(width, length, nw, nl):
    xs, ys = np.linspace(0.0, length, nl + 1), np.linspace(0.0, width, nw + 1)
    X, Y = np.meshgrid(xs, ys, indexing="ij")
    nodes = np.stack([X.ravel(), Y.ravel(), np.zeros(X.size)], axis=1)
    idx = np.arange((nl + 1) * (nw + 1)).reshape(nl + 1, nw + 1)
    cells = np.stack([idx[:-1, :-1].ravel(), idx[1:, :-1].ravel(), idx[1:, 1:].ravel(), idx[:-1, 1:].ravel()], axis=1)
    return nodes, cells.astype(np.int64)


def main():
    nodes, cells = plate(2.0, 10.0, 4, 20)
    # dolfinx writes the vertices in its own (reordered) numbering: shuffle them so that the fixture is not trivially ordered
    perm = np.random.default_rng(5).permutation(nodes.shape[0])
    inv = np.empty_like(perm); inv[perm] = np.arange(perm.size)
    nodes, cells = nodes[perm], inv[cells]
    for suffix, kw, ds in (("", dict(libver="earliest"), {}), ("_chunked", dict(libver="earliest"), dict(chunks=True)),
                           ("_latest", dict(libver="latest"), {})):
        stem = "plate_2_10_quad_4_20" + suffix
        with h5py.File(os.path.join(HERE, stem + ".h5"), "w", **kw) as f:
            g = f.create_group("Mesh").create_group("Grid")
            if ds:
                g.create_dataset("topology", data=cells, chunks=(16, 4))
                g.create_dataset("geometry", data=nodes, chunks=(32, 3))
            else:
                g.create_dataset("topology", data=cells)
                g.create_dataset("geometry", data=nodes)
        with open(os.path.join(HERE, stem + ".xdmf"), "w") as fh:
            fh.write(XDMF.format(nel=cells.shape[0], nn=nodes.shape[0], h5=stem + ".h5"))
    np.savez(os.path.join(HERE, "plate_2_10_quad_4_20_expected.npz"), nodes=nodes, cells=cells)


if __name__ == "__main__":
    main()
