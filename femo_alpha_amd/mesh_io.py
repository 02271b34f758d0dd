"""Mesh files in and out of :class:`ShellMesh` — the reference's mesh entry points without dolfinx/meshio.

* ``readFEAMesh(meshFile, format)``  — reference ``fea/utils_dolfinx.py:34-50`` (XDMF grid named "Grid");
* ``reconstructFEAMesh(filename, nodes, connectivity)`` — reference ``fea/utils_dolfinx.py:652-668`` (the CADDEE
  callers hand over node coordinates and a connectivity table; the reference writes them with meshio and reads
  them back through dolfinx — here the file is written for whoever wants it and the mesh is built directly);
* ``read_msh`` — Gmsh ASCII 2.2 / 4.1 surface meshes (what the reference's advanced examples feed to CADDEE's
  ``import_shell_mesh``), with the physical groups returned as ``mesh_tags`` for ``RMShellModel``.

XDMF heavy data: inline XML (``Format="XML"``) is read and written natively; ``Format="HDF"`` -- the format of every
mesh the reference ships (fea/utils_dolfinx.py:34-50: ``XDMFFile(...).read_mesh(name="Grid")``) -- is read through the
library-free HDF5 parser in ``hdf5_min.py`` (h5py is not part of this image).
"""
from __future__ import annotations

import os
import xml.etree.ElementTree as ET

import numpy as np

from .mesh import ShellMesh

_XDMF_CELLS = {"quadrilateral": 4, "triangle": 3}


def _read_dataitem(item, base_dir):
    dims = tuple(int(d) for d in item.get("Dimensions").split())
    fmt = (item.get("Format") or "XML").upper()
    is_int = (item.get("NumberType", item.get("DataType", "Float")) or "Float").lower() in ("int", "uint")
    if fmt == "XML":
        data = np.array(item.text.split(), dtype=np.int64 if is_int else np.float64)
    elif fmt == "HDF":
        # dolfinx keeps the arrays of every mesh it writes in an HDF5 file next to the XML ("mesh.h5:/Mesh/Grid/topology");
        # they are plain contiguous datasets, read here without h5py / libhdf5 (femo_alpha_amd/hdf5_min.py)
        from .hdf5_min import read_dataset
        fname, _, path = item.text.strip().partition(":")
        data = read_dataset(os.path.join(base_dir, fname), path)
    else:
        raise ValueError(f"unsupported XDMF DataItem format '{fmt}'")
    return data.reshape(dims)


def read_xdmf(path, name="Grid"):
    """ShellMesh from the uniform grid ``name`` of an XDMF file (triangle or quadrilateral topology)."""
    root = ET.parse(path).getroot()
    grid = None
    for g in root.iter("Grid"):
        if g.get("Name") == name or grid is None:
            grid = g
            if g.get("Name") == name:
                break
    if grid is None:
        raise ValueError(f"no Grid in {path}")
    topo, geom = grid.find("Topology"), grid.find("Geometry")
    ttype = (topo.get("TopologyType") or topo.get("Type") or "").lower()
    if ttype not in _XDMF_CELLS:
        raise ValueError("Invalid cell shape--should be either triangular or quadrilateral")
    base = os.path.dirname(os.path.abspath(path))
    cells = _read_dataitem(topo.find("DataItem"), base).astype(np.int64).reshape(-1, _XDMF_CELLS[ttype])
    xyz = _read_dataitem(geom.find("DataItem"), base).astype(np.float64)
    if xyz.shape[1] == 2:
        xyz = np.hstack([xyz, np.zeros((xyz.shape[0], 1))])
    return ShellMesh(xyz, cells)


def write_xdmf(path, nodes, cells, name="Grid"):
    """Self-contained XDMF (inline XML arrays) of a triangle / quadrilateral surface mesh."""
    nodes = np.asarray(nodes, dtype=np.float64)
    cells = np.asarray(cells, dtype=np.int64)
    ttype = {4: "Quadrilateral", 3: "Triangle"}[cells.shape[1]]
    root = ET.Element("Xdmf", Version="3.0")
    grid = ET.SubElement(ET.SubElement(root, "Domain"), "Grid", Name=name, GridType="Uniform")
    topo = ET.SubElement(grid, "Topology", TopologyType=ttype, NumberOfElements=str(cells.shape[0]),
                         NodesPerElement=str(cells.shape[1]))
    ti = ET.SubElement(topo, "DataItem", Dimensions=f"{cells.shape[0]} {cells.shape[1]}", NumberType="Int", Format="XML")
    ti.text = "\n" + "\n".join(" ".join(str(v) for v in row) for row in cells) + "\n"
    geom = ET.SubElement(grid, "Geometry", GeometryType="XYZ")
    gi = ET.SubElement(geom, "DataItem", Dimensions=f"{nodes.shape[0]} 3", NumberType="Float", Precision="8", Format="XML")
    gi.text = "\n" + "\n".join(" ".join(repr(float(v)) for v in row) for row in nodes) + "\n"
    ET.ElementTree(root).write(path, xml_declaration=True, encoding="utf-8")


def readFEAMesh(meshFile, format="HDF"):
    """Reads the surface mesh of an XDMF file (reference fea/utils_dolfinx.py:34-50; same format switch)."""
    if format not in ("HDF", "XML"):
        raise ValueError("Invalid mesh file type. Must be 'HDF' or 'XML'")
    return read_xdmf(meshFile, name="Grid")


def reconstructFEAMesh(filename, nodes, connectivity):
    """ShellMesh from caller-side node coordinates and connectivity; also written to ``filename`` as XDMF
    (reference fea/utils_dolfinx.py:652-668)."""
    connectivity = np.asarray(connectivity)
    if connectivity.ndim != 2 or connectivity.shape[1] not in (3, 4):
        raise ValueError("Invalid cell shape--should be either triangular or quadrilateral")
    nodes = np.asarray(nodes, dtype=np.float64)
    if filename:
        write_xdmf(filename, nodes, connectivity)
    return ShellMesh(nodes, connectivity.astype(np.int64))


# ------------------------------------------------------------------------------------------------ Gmsh
_GMSH_SURFACE = {2: 3, 3: 4}           # element type -> vertices (3-node triangle, 4-node quadrangle)


def _sections(path):
    out, cur, name = {}, None, None
    with open(path) as fh:
        for line in fh:
            s = line.strip()
            if s.startswith("$End"):
                out[name] = cur
                cur = name = None
            elif s.startswith("$"):
                name, cur = s[1:], []
            elif cur is not None:
                cur.append(s)
    return out


def read_msh(path, rescale=None):
    """(ShellMesh, mesh_tags) from a Gmsh ASCII file, format 2.2 or 4.1.  Triangles and quadrangles may not be
    mixed (the element is one or the other, as in the reference).  ``mesh_tags`` maps the physical-group name
    (or number) to the list of cell indices carrying it; nodes not used by any surface cell are dropped."""
    sec = _sections(path)
    version = float(sec["MeshFormat"][0].split()[0])
    names = {}
    for line in sec.get("PhysicalNames", [])[1:]:
        dim, tag, nm = line.split(maxsplit=2)
        if int(dim) == 2:
            names[int(tag)] = nm.strip('"')
    node_id, xyz, cells, phys = [], [], [], []
    if version < 4.0:
        for line in sec["Nodes"][1:]:
            p = line.split()
            node_id.append(int(p[0])); xyz.append([float(v) for v in p[1:4]])
        for line in sec["Elements"][1:]:
            p = [int(v) for v in line.split()]
            etype, ntag = p[1], p[2]
            if etype in _GMSH_SURFACE:
                cells.append(p[3 + ntag:3 + ntag + _GMSH_SURFACE[etype]])
                phys.append(p[3] if ntag > 0 else 0)
    else:
        ent_phys = {}
        if "Entities" in sec:
            ent = sec["Entities"]
            npnt, ncur, nsur, _ = (int(v) for v in ent[0].split())
            for line in ent[1 + npnt + ncur:1 + npnt + ncur + nsur]:
                p = line.split()
                nph = int(p[7])
                ent_phys[int(p[0])] = int(p[8]) if nph > 0 else 0
        lines = sec["Nodes"]
        nblocks = int(lines[0].split()[0])
        i = 1
        for _ in range(nblocks):
            _, _, parametric, nb = (int(v) for v in lines[i].split())
            ids = [int(lines[i + 1 + k]) for k in range(nb)]
            for k in range(nb):
                xyz.append([float(v) for v in lines[i + 1 + nb + k].split()[:3]])
            node_id.extend(ids)
            i += 1 + 2 * nb
        lines = sec["Elements"]
        nblocks = int(lines[0].split()[0])
        i = 1
        for _ in range(nblocks):
            edim, etag, etype, nb = (int(v) for v in lines[i].split())
            if etype in _GMSH_SURFACE:
                for k in range(nb):
                    p = [int(v) for v in lines[i + 1 + k].split()]
                    cells.append(p[1:1 + _GMSH_SURFACE[etype]])
                    phys.append(ent_phys.get(etag, 0))
            i += 1 + nb
    if not cells:
        raise ValueError("no triangle or quadrangle cells in " + str(path))
    if len({len(c) for c in cells}) != 1:
        raise ValueError("Invalid cell shape--should be either triangular or quadrilateral")
    cells = np.asarray(cells, dtype=np.int64)
    node_id = np.asarray(node_id, dtype=np.int64)
    xyz = np.asarray(xyz, dtype=np.float64)
    used = np.unique(cells)
    lookup = -np.ones(node_id.max() + 1, dtype=np.int64)
    pos = -np.ones(node_id.max() + 1, dtype=np.int64)
    pos[node_id] = np.arange(node_id.size)
    lookup[used] = np.arange(used.size)
    nodes = xyz[pos[used]]
    if rescale is not None:
        nodes = nodes * np.asarray(rescale, dtype=np.float64)
    mesh = ShellMesh(nodes, lookup[cells])
    phys = np.asarray(phys)
    mesh_tags = {names.get(int(t), int(t)): np.nonzero(phys == t)[0].tolist() for t in np.unique(phys) if t != 0}
    return mesh, mesh_tags
