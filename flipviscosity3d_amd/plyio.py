"""Binary little-endian PLY triangle-mesh reader/writer.

Mirrors the file format of the reference's TriangleMesh::loadPLY / writeMeshToPLY
(reference trianglemesh.cpp:39-63, 190-343, 426-615): header lines
``ply / format binary_little_endian 1.0 / element vertex N / property float x,y,z /
[property uchar red,green,blue] / element face M / property list uchar int vertex_index /
end_header`` followed by N vertex records and M x (u8 count=3, 3 x i32).

Unlike the reference loader, files shorter than 2048 bytes load fine (the reference reads a
fixed 2048-byte header block and fails on cube.ply / sheet.ply / cone.ply).
"""
import numpy as np


def load_ply(path):
    """Return (vertices float32 (N,3), triangles int32 (M,3))."""
    with open(path, "rb") as f:
        data = f.read()
    end = data.find(b"end_header\n")
    if end < 0:
        raise ValueError("%s: no PLY header" % path)
    header = data[:end].decode("ascii", "replace").split("\n")
    body = data[end + len(b"end_header\n"):]
    if not any(h.strip() == "format binary_little_endian 1.0" for h in header):
        raise ValueError("%s: only binary_little_endian 1.0 is supported" % path)
    nv = nf = None
    has_color = False
    for h in header:
        t = h.split()
        if len(t) == 3 and t[0] == "element" and t[1] == "vertex":
            nv = int(t[2])
        elif len(t) == 3 and t[0] == "element" and t[1] == "face":
            nf = int(t[2])
        elif len(t) == 3 and t[0] == "property" and t[1] == "uchar" and t[2] == "red":
            has_color = True
    if nv is None or nf is None:
        raise ValueError("%s: missing element counts" % path)
    vdt = [("x", "<f4"), ("y", "<f4"), ("z", "<f4")]
    if has_color:
        vdt += [("r", "u1"), ("g", "u1"), ("b", "u1")]
    vdt = np.dtype(vdt)
    fdt = np.dtype([("n", "u1"), ("a", "<i4"), ("b", "<i4"), ("c", "<i4")])
    need = nv * vdt.itemsize + nf * fdt.itemsize
    if len(body) < need:
        raise ValueError("%s: truncated PLY body" % path)
    v = np.frombuffer(body, vdt, nv, 0)
    fc = np.frombuffer(body, fdt, nf, nv * vdt.itemsize)
    if nf and not np.all(fc["n"] == 3):
        raise ValueError("%s: non-triangle faces" % path)
    verts = np.stack([v["x"], v["y"], v["z"]], 1).astype(np.float32)
    tris = np.stack([fc["a"], fc["b"], fc["c"]], 1).astype(np.int32)
    return np.ascontiguousarray(verts), np.ascontiguousarray(tris)


def save_ply(path, verts, tris):
    verts = np.ascontiguousarray(verts, "<f4").reshape(-1, 3)
    tris = np.ascontiguousarray(tris, "<i4").reshape(-1, 3)
    hdr = ("ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty float x\n"
           "property float y\nproperty float z\nelement face %d\n"
           "property list uchar int vertex_index\nend_header\n" % (len(verts), len(tris)))
    fdt = np.dtype([("n", "u1"), ("a", "<i4"), ("b", "<i4"), ("c", "<i4")])
    f = np.empty(len(tris), fdt)
    f["n"] = 3
    f["a"], f["b"], f["c"] = tris[:, 0], tris[:, 1], tris[:, 2]
    with open(path, "wb") as fh:
        fh.write(hdr.encode("ascii"))
        fh.write(verts.tobytes())
        fh.write(f.tobytes())
