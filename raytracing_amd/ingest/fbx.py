"""Binary FBX reader: the meshes of a model file, as Unity's importer hands them to
`MeshFilter.sharedMesh` (what BVH.InitMesh reads, Assets/Scripts/BVH/BVH.cs:229-241).

Only what that path consumes is produced: positions, normals, UVs and triangles.
Tangents are left zero (the shaders never read them).

Format (Kaydara FBX Binary, versions 7100-7700): 27-byte header, then node
records {end offset, property count, property bytes, name, properties, nested
records, 13/25-byte null record}.  Array properties may be zlib-deflated.

Conversions restated from the importer's documented behaviour (the importer is
not available here, so they are not pinned by any output of the reference):
  * FBX is right-handed, Unity left-handed: x is negated and the triangle winding
    reversed (`bakeAxisConversion: 0`, Y-up files);
  * scale = `globalScale` of the .meta times the file's unit scale
    (UnitScaleFactor / 100, centimetres -> metres) when `useFileScale: 1`;
  * polygons are triangulated as fans; vertices are split per polygon corner
    whenever normals or UVs are stored per corner, and welded again when
    position, normal and UV coincide (`weldVertices: 1`).
"""
import struct
import zlib
from dataclasses import dataclass, field
from typing import Any, Dict, List

import numpy as np

MAGIC = b"Kaydara FBX Binary  \x00\x1a\x00"


@dataclass
class Node:
    name: str
    props: List[Any] = field(default_factory=list)
    children: List["Node"] = field(default_factory=list)

    def find(self, name):
        return [c for c in self.children if c.name == name]

    def first(self, name):
        for c in self.children:
            if c.name == name:
                return c
        return None


_SCALAR = {"Y": ("<h", 2), "C": ("<b", 1), "I": ("<i", 4), "F": ("<f", 4), "D": ("<d", 8), "L": ("<q", 8)}
_ARRAY = {"f": "<f4", "d": "<f8", "l": "<i8", "i": "<i4", "b": "i1"}


def parse(data: bytes) -> Node:
    """The whole record tree under a synthetic root."""
    if data[:23] != MAGIC:
        raise ValueError("not a binary FBX file")
    version = struct.unpack_from("<I", data, 23)[0]
    wide = version >= 7500
    head = struct.Struct("<QQQ" if wide else "<III")
    root = Node("")

    def read(pos, parent):
        end, nprops, plen = head.unpack_from(data, pos)
        pos += head.size
        nlen = data[pos]
        name = data[pos + 1:pos + 1 + nlen].decode("ascii", "replace")
        pos += 1 + nlen
        if end == 0:
            return None, pos
        node = Node(name)
        p = pos
        for _ in range(nprops):
            t = chr(data[p])
            p += 1
            if t in _SCALAR:
                fmt, size = _SCALAR[t]
                node.props.append(struct.unpack_from(fmt, data, p)[0])
                p += size
            elif t in _ARRAY:
                count, enc, clen = struct.unpack_from("<III", data, p)
                p += 12
                raw = data[p:p + clen]
                if enc == 1:
                    raw = zlib.decompress(raw)
                node.props.append(np.frombuffer(raw, _ARRAY[t], count).copy())
                p += clen
            elif t in "SR":
                n = struct.unpack_from("<I", data, p)[0]
                p += 4
                raw = data[p:p + n]
                node.props.append(raw.decode("utf-8", "replace") if t == "S" else bytes(raw))
                p += n
            else:
                raise ValueError(f"unknown FBX property type {t!r}")
        pos += plen
        while pos < end:
            child, pos = read(pos, node)
            if child is None:
                break
        parent.children.append(node)
        return node, end

    pos = 27
    while pos + head.size + 1 <= len(data):
        node, pos = read(pos, root)
        if node is None:
            break
    return root


def _properties70(node) -> Dict[str, list]:
    out = {}
    p70 = node.first("Properties70") if node else None
    if p70:
        for p in p70.find("P"):
            out[p.props[0]] = p.props[4:]
    return out


@dataclass
class FbxMesh:
    name: str
    positions: np.ndarray        # f32[V,3]   Unity mesh space
    normals: np.ndarray          # f32[V,3]
    uvs: np.ndarray              # f32[V,2]
    triangles: np.ndarray        # u32[T,3]
    # the model node's own transform (Unity convention), used when the file is placed as a prefab
    local_position: tuple = (0.0, 0.0, 0.0)
    local_euler: tuple = (0.0, 0.0, 0.0)
    local_scale: tuple = (1.0, 1.0, 1.0)


def _layer(geom, layer_name, data_name, index_name, n_corners, corner_vertex, width):
    """One layer element expanded to per-corner values (or None)."""
    le = geom.first(layer_name)
    if le is None or le.first(data_name) is None:
        return None
    mapping = le.first("MappingInformationType").props[0]
    ref = le.first("ReferenceInformationType").props[0]
    vals = np.asarray(le.first(data_name).props[0], np.float64).reshape(-1, width)
    if mapping == "ByPolygonVertex":
        which = np.arange(n_corners)
    elif mapping in ("ByVertice", "ByVertex"):
        which = corner_vertex
    elif mapping == "AllSame":
        which = np.zeros(n_corners, np.int64)
    else:
        raise ValueError(f"{layer_name}: mapping {mapping} not supported")
    if ref == "IndexToDirect" and mapping != "AllSame":
        idx = np.asarray(le.first(index_name).props[0], np.int64)
        which = idx[which]
    elif ref not in ("Direct", "IndexToDirect"):
        raise ValueError(f"{layer_name}: reference {ref} not supported")
    return vals[which]


def load_meshes(data: bytes, global_scale=1.0, use_file_scale=False, weld=True) -> List[FbxMesh]:
    root = parse(data)
    settings = _properties70(root.first("GlobalSettings"))
    unit = float(settings.get("UnitScaleFactor", [1.0])[0])
    scale = float(global_scale) * (unit / 100.0 if use_file_scale else 1.0)
    objects = root.first("Objects")
    if objects is None:
        return []
    models = {m.props[0]: m for m in objects.find("Model")}
    owner = {}
    conns = root.first("Connections")
    for c in conns.find("C") if conns else []:
        if c.props[0] == "OO" and c.props[2] in models:
            owner[c.props[1]] = models[c.props[2]]
    out = []
    for geom in objects.find("Geometry"):
        if geom.first("Vertices") is None or geom.first("PolygonVertexIndex") is None:
            continue
        ctrl = np.asarray(geom.first("Vertices").props[0], np.float64).reshape(-1, 3)
        pvi = np.asarray(geom.first("PolygonVertexIndex").props[0], np.int64)
        ends = pvi < 0
        corner_vertex = np.where(ends, ~pvi, pvi)            # the last corner of a polygon is stored as ~index
        n_corners = len(pvi)
        normals = _layer(geom, "LayerElementNormal", "Normals", "NormalsIndex", n_corners, corner_vertex, 3)
        uvs = _layer(geom, "LayerElementUV", "UV", "UVIndex", n_corners, corner_vertex, 2)
        pos = ctrl[corner_vertex] * scale
        if normals is None:
            normals = np.zeros_like(pos)
        if uvs is None:
            uvs = np.zeros((n_corners, 2))
        # right-handed -> left-handed
        pos = pos * np.array([-1.0, 1.0, 1.0])
        normals = normals * np.array([-1.0, 1.0, 1.0])
        # polygons -> triangle fans, winding reversed with the mirror
        starts = np.concatenate([[0], np.nonzero(ends)[0][:-1] + 1])
        sizes = np.nonzero(ends)[0] + 1 - starts
        tris = []
        for s, n in zip(starts, sizes):
            for k in range(1, n - 1):
                tris.append((s, s + k + 1, s + k))
        tris = np.asarray(tris, np.int64).reshape(-1, 3)
        if weld:
            key = np.concatenate([pos, normals, uvs], 1).astype(np.float32)
            _, first, inverse = np.unique(key, axis=0, return_index=True, return_inverse=True)
            order = np.argsort(first)                         # keep first-appearance order
            rank = np.empty_like(order)
            rank[order] = np.arange(len(order))
            remap = rank[inverse.reshape(-1)]
            keep = first[order]
            pos, normals, uvs = pos[keep], normals[keep], uvs[keep]
            tris = remap[tris]
        model = owner.get(geom.props[0])
        name = (model.props[1].split("\x00")[0] if model else geom.props[1].split("\x00")[0]) or "mesh"
        mesh = FbxMesh(name, pos.astype(np.float32), normals.astype(np.float32), uvs.astype(np.float32), tris.astype(np.uint32))
        if model is not None:
            p = _properties70(model)
            t = p.get("Lcl Translation", [0.0, 0.0, 0.0])
            r = p.get("Lcl Rotation", [0.0, 0.0, 0.0])
            s = p.get("Lcl Scaling", [1.0, 1.0, 1.0])
            mesh.local_position = (-float(t[0]) * scale, float(t[1]) * scale, float(t[2]) * scale)
            mesh.local_euler = (float(r[0]), -float(r[1]), -float(r[2]))     # mirrored x: rotations about y and z flip
            mesh.local_scale = tuple(float(v) for v in s)
        out.append(mesh)
    return out


def write_minimal(meshes, version=7500, unit_scale=1.0, compress=True) -> bytes:
    """A binary FBX holding polygon meshes -- the inverse of what `load_meshes` needs, for tests.
    meshes: list of (name, control_points[V,3], polygons (list of index lists), per-corner normals or None)."""
    wide = version >= 7500
    head = struct.Struct("<QQQ" if wide else "<III")

    def prop(v):
        if isinstance(v, bool):
            return b"C" + struct.pack("<b", int(v))
        if isinstance(v, int):
            return (b"L" + struct.pack("<q", v)) if abs(v) > 0x7FFFFFFF else (b"I" + struct.pack("<i", v))
        if isinstance(v, float):
            return b"D" + struct.pack("<d", v)
        if isinstance(v, str):
            raw = v.encode()
            return b"S" + struct.pack("<I", len(raw)) + raw
        if isinstance(v, np.ndarray):
            code = {np.dtype("<f8"): b"d", np.dtype("<i4"): b"i", np.dtype("<f4"): b"f", np.dtype("<i8"): b"l"}[v.dtype]
            raw = v.tobytes()
            body = zlib.compress(raw) if compress else raw
            return code + struct.pack("<III", v.size, 1 if compress else 0, len(body)) + body
        raise TypeError(type(v))

    def node(name, props=(), children=()):
        return (name, list(props), list(children))

    def emit(buf, n):
        name, props, children = n
        pbytes = b"".join(prop(p) for p in props)
        start = len(buf)
        buf += b"\0" * head.size
        buf += bytes([len(name)]) + name.encode()
        buf += pbytes
        for c in children:
            emit(buf, c)
        if children:
            buf += b"\0" * (head.size + 1)
        buf[start:start + head.size] = head.pack(len(buf), len(props), len(pbytes))

    objs, conns = [], []
    for k, (name, ctrl, polys, normals) in enumerate(meshes):
        gid, mid = 1000 + 2 * k, 1001 + 2 * k
        pvi = []
        for poly in polys:
            pvi += list(poly[:-1]) + [~poly[-1]]
        children = [node("Vertices", [np.asarray(ctrl, "<f8").reshape(-1)]),
                    node("PolygonVertexIndex", [np.asarray(pvi, "<i4")])]
        if normals is not None:
            children.append(node("LayerElementNormal", [0], [node("MappingInformationType", ["ByPolygonVertex"]),
                                                             node("ReferenceInformationType", ["Direct"]),
                                                             node("Normals", [np.asarray(normals, "<f8").reshape(-1)])]))
        objs.append(node("Geometry", [gid, "\x00\x01Geometry", "Mesh"], children))
        objs.append(node("Model", [mid, name + "\x00\x01Model", "Mesh"], [node("Version", [232])]))
        conns += [node("C", ["OO", mid, 0]), node("C", ["OO", gid, mid])]
    top = [node("GlobalSettings", [], [node("Properties70", [], [node("P", ["UnitScaleFactor", "double", "Number", "", float(unit_scale)])])]),
           node("Objects", [], objs), node("Connections", [], conns)]
    buf = bytearray(MAGIC + struct.pack("<I", version))
    for t in top:
        emit(buf, t)
    buf += b"\0" * (head.size + 1)
    return bytes(buf)
