"""Result writers and wire formats of the reference's inference path.

Host-side file I/O, as in the reference (it is Python there too); nothing here touches the GPU.

* `HighPrecisionLabeler` -- reference inference.py:14-41: 24-bit label <-> RGB in [0,1].  Same
  four methods; the per-element Python loops of the reference are vectorised (identical values).
* `write_point_cloud_ply` / `read_point_cloud_ply` -- the PLY dialect Open3D writes for a point
  cloud with colours (what `o3d.io.write_point_cloud` produces in reference inference.py:207,212
  and what the files under the reference's postInferenceTest/ look like): binary little endian,
  `comment Created by Open3D`, double x/y/z, uchar red/green/blue.  Open3D is not installed in
  this image, so the byte layout follows the headers of the reference's own output files and
  Open3D's documented colour rounding (clamp to [0,1], times 255, round half away from zero);
  parity with Open3D itself is unpinned.
* `save_results` -- reference inference.py:190-219: `<out>.ply` (labels as colours),
  `<stem>_original.ply` (original colours), `<stem>_labels.npy`.
* `save_segmented_ply`, `save_as_json`, `find_instance_labels` -- reference
  saveVertexGroup.py:13-103: ASCII PLY with `comment vertex_group instance_<id> <indices>` lines
  plus a summary `.txt`, or a JSON index map.
"""
import json
from pathlib import Path
from typing import Optional, Tuple

import numpy as np


class HighPrecisionLabeler:
    """reference inference.py:14-41."""

    @staticmethod
    def encode_label_to_color(label: int) -> np.ndarray:
        label = int(label)
        return np.array([(label & 0xFF0000) >> 16, (label & 0x00FF00) >> 8, label & 0x0000FF]) / 255.0

    @staticmethod
    def decode_color_to_label(color: np.ndarray) -> int:
        c = np.round(np.asarray(color) * 255).astype(int)
        return int((c[0] << 16) | (c[1] << 8) | c[2])

    def labels_to_colors(self, labels: np.ndarray) -> np.ndarray:
        lab = np.asarray(labels).astype(np.int64)
        out = np.empty((len(lab), 3), dtype=np.float64)
        out[:, 0] = (lab & 0xFF0000) >> 16
        out[:, 1] = (lab & 0x00FF00) >> 8
        out[:, 2] = lab & 0x0000FF
        return out / 255.0

    def colors_to_labels(self, colors: np.ndarray) -> np.ndarray:
        c = np.round(np.asarray(colors, dtype=np.float64) * 255).astype(np.int64)
        if c.size == 0:
            return np.zeros(0, dtype=np.int32)
        return ((c[:, 0] << 16) | (c[:, 1] << 8) | c[:, 2]).astype(np.int32)


def _colors_to_uchar(colors: np.ndarray) -> np.ndarray:
    c = np.clip(np.asarray(colors, dtype=np.float64), 0.0, 1.0) * 255.0
    return np.floor(c + 0.5).astype(np.uint8)          # round half away from zero (values are >= 0)


_PLY_DTYPE = np.dtype([("x", "<f8"), ("y", "<f8"), ("z", "<f8"), ("red", "u1"), ("green", "u1"), ("blue", "u1")])
_PLY_DTYPE_XYZ = np.dtype([("x", "<f8"), ("y", "<f8"), ("z", "<f8")])


def write_point_cloud_ply(path, points: np.ndarray, colors: Optional[np.ndarray] = None) -> None:
    """Binary little-endian PLY as Open3D writes a coloured point cloud."""
    pts = np.asarray(points, dtype=np.float64).reshape(-1, 3)
    n = len(pts)
    head = ["ply", "format binary_little_endian 1.0", "comment Created by Open3D", f"element vertex {n}",
            "property double x", "property double y", "property double z"]
    if colors is not None:
        col = _colors_to_uchar(np.asarray(colors).reshape(-1, 3))
        if len(col) != n:
            raise ValueError(f"write_point_cloud_ply: {n} points but {len(col)} colours")
        head += ["property uchar red", "property uchar green", "property uchar blue"]
        rec = np.empty(n, dtype=_PLY_DTYPE)
        rec["red"], rec["green"], rec["blue"] = col[:, 0], col[:, 1], col[:, 2]
    else:
        rec = np.empty(n, dtype=_PLY_DTYPE_XYZ)
    rec["x"], rec["y"], rec["z"] = pts[:, 0], pts[:, 1], pts[:, 2]
    head.append("end_header")
    with open(path, "wb") as f:
        f.write(("\n".join(head) + "\n").encode("ascii"))
        f.write(rec.tobytes())


_PLY_TYPES = {"char": "i1", "uchar": "u1", "short": "i2", "ushort": "u2", "int": "i4", "uint": "u4", "float": "f4",
              "double": "f8", "int8": "i1", "uint8": "u1", "int16": "i2", "uint16": "u2", "int32": "i4",
              "uint32": "u4", "float32": "f4", "float64": "f8"}


def read_point_cloud_ply(path) -> Tuple[np.ndarray, Optional[np.ndarray], list]:
    """Vertex element of a PLY file (ascii or binary, either endianness) -> points (N,3) float64,
    colors (N,3) float64 in [0,1] or None, and the header's comment lines.  Enough of the format
    for the files this module and the reference's tools write (scalar vertex properties; other
    elements after the vertices are ignored)."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, n, props, comments, in_vertex = None, 0, [], [], False
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: header without end_header")
            tok = line.decode("ascii", "replace").strip()
            if tok == "end_header":
                break
            parts = tok.split()
            if not parts:
                continue
            if parts[0] == "format":
                fmt = parts[1]
            elif parts[0] == "comment":
                comments.append(tok[len("comment"):].strip())
            elif parts[0] == "element":
                in_vertex = parts[1] == "vertex"
                if in_vertex:
                    n = int(parts[2])
            elif parts[0] == "property" and in_vertex:
                if parts[1] == "list":
                    raise ValueError(f"{path}: list property on the vertex element")
                props.append((parts[2], _PLY_TYPES[parts[1]]))
        if fmt == "ascii":
            rows = [f.readline().split() for _ in range(n)]
            data = {name: np.array([r[i] for r in rows], dtype=np.float64) for i, (name, _) in enumerate(props)}
        else:
            end = "<" if fmt == "binary_little_endian" else ">"
            dt = np.dtype([(name, end + t if t[1] != "1" else t) for name, t in props])
            rec = np.frombuffer(f.read(n * dt.itemsize), dtype=dt, count=n)
            data = {name: rec[name].astype(np.float64) for name, _ in props}
    points = np.stack([data["x"], data["y"], data["z"]], axis=1) if n else np.zeros((0, 3))
    colors = None
    if all(k in data for k in ("red", "green", "blue")):
        colors = np.stack([data["red"], data["green"], data["blue"]], axis=1) / 255.0 if n else np.zeros((0, 3))
    return points, colors, comments


def save_results(points: np.ndarray, colors: Optional[np.ndarray], instance_labels: np.ndarray, output_path) -> None:
    """reference inference.py:190-219 (PointCloudInference.save_results)."""
    output_path = Path(output_path)
    output_path.parent.mkdir(parents=True, exist_ok=True)
    labels = np.asarray(instance_labels)
    write_point_cloud_ply(output_path, points, HighPrecisionLabeler().labels_to_colors(labels))
    if colors is not None:
        write_point_cloud_ply(output_path.parent / f"{output_path.stem}_original.ply", points, colors)
    np.save(str(output_path.parent / f"{output_path.stem}_labels.npy"), labels)


def find_instance_labels(ply_path) -> Path:
    """reference saveVertexGroup.py:13-22."""
    ply_path = Path(ply_path)
    base = ply_path.stem
    if base.endswith("_labeled"):
        base = base[:-8]
    npy = ply_path.parent / f"{base}_labeled_labels.npy"
    if not npy.exists():
        raise FileNotFoundError(f"Instance labels not found at {npy}")
    return npy


def save_segmented_ply(points, instance_labels, output_path, colors=None) -> None:
    """reference saveVertexGroup.py:24-82: ASCII PLY whose header carries one
    `comment vertex_group instance_<id> <indices...>` line per instance (label -1 skipped), and a
    summary text file next to it.  Number formatting is Python's str() of the array scalars and
    int(c*255) truncation for colours, as there."""
    output_path = Path(output_path).with_suffix(".ply")
    points = np.asarray(points)
    labels = np.asarray(instance_labels)
    uniq = np.unique(labels)
    head = ["ply", "format ascii 1.0", f"element vertex {len(points)}", "property float x", "property float y",
            "property float z"]
    if colors is not None:
        head += ["property uchar red", "property uchar green", "property uchar blue"]
    for u in uniq:
        if u == -1:
            continue
        head.append(f"comment vertex_group instance_{u} {' '.join(map(str, np.where(labels == u)[0]))}")
    head.append("end_header")
    with open(output_path, "w") as f:
        f.write("\n".join(head) + "\n")
        for i in range(len(points)):
            line = f"{points[i][0]} {points[i][1]} {points[i][2]}"
            if colors is not None:
                line += f" {int(colors[i][0] * 255)} {int(colors[i][1] * 255)} {int(colors[i][2] * 255)}"
            f.write(line + "\n")
    with open(output_path.with_suffix(".txt"), "w") as f:
        f.write(f"Total points: {len(points)}\n")
        f.write(f"Number of instances: {len(uniq) - 1}\n\n")
        for u in uniq:
            if u == -1:
                continue
            f.write(f"Instance {u}:\n")
            f.write(f"  Number of points: {np.sum(labels == u)}\n")


def save_as_json(points, instance_labels, output_path, colors=None) -> None:
    """reference saveVertexGroup.py:84-103."""
    labels = np.asarray(instance_labels)
    groups = {}
    for u in np.unique(labels):
        if u == -1:
            continue
        groups[f"instance_{u}"] = {"indices": np.where(labels == u)[0].tolist()}
    with open(Path(output_path).with_suffix(".json"), "w") as f:
        json.dump(groups, f, indent=2)
