"""Mask -> boundary edge segments (drop-in for ``qpsim.geometry.extract_edge_segments``).

The solver consumes ``edges`` + ``edge_conditions`` keyed by ``edge_id``; an existing
``edge_conditions`` dict only applies if ids and ordering are identical to the reference's
(``qpsim/geometry.py:150-242``): horizontal faces grouped by (y, normal) with "down" before
"up", then vertical faces grouped by (x, normal) with "left" before "right", each group split
into maximal contiguous runs, ids ``edge_0001`` ... in that order.  This version finds the
runs with array operations per grid line instead of a per-cell Python walk, so a 4096^2 mask
takes seconds, not minutes.  GDS import / rasterisation (geometry.py:23-128, 265-295) is out of
scope (needs gdstk, host-side preprocessing).
"""
from __future__ import annotations

import numpy as np

from .models import BoundaryFace, EdgeSegment, GeometryData


def boundary_face_masks(mask: np.ndarray) -> dict[str, np.ndarray]:
    """Boolean [ny, nx] arrays: cell is interior and its neighbour in that direction is not."""
    m = np.asarray(mask, dtype=bool)
    pad = np.zeros((m.shape[0] + 2, m.shape[1] + 2), dtype=bool)
    pad[1:-1, 1:-1] = m
    return {
        "up": m & ~pad[:-2, 1:-1],
        "down": m & ~pad[2:, 1:-1],
        "left": m & ~pad[1:-1, :-2],
        "right": m & ~pad[1:-1, 2:],
    }


def _runs(positions: np.ndarray) -> list[tuple[int, int]]:
    """Maximal runs of consecutive integers in a sorted array, as (start, stop_exclusive) index pairs."""
    if positions.size == 0:
        return []
    breaks = np.nonzero(np.diff(positions) != 1)[0] + 1
    starts = np.concatenate([[0], breaks])
    stops = np.concatenate([breaks, [positions.size]])
    return list(zip(starts.tolist(), stops.tolist()))


def extract_edge_segments(mask: np.ndarray) -> list[EdgeSegment]:
    mask = np.asarray(mask, dtype=bool)
    ny, nx = mask.shape
    faces = boundary_face_masks(mask)
    segments: list[EdgeSegment] = []

    def emit(normal: str, line: int, cells: np.ndarray, horizontal: bool, cell_line: int) -> None:
        for a, b in _runs(cells):
            lo, hi = int(cells[a]), int(cells[b - 1]) + 1
            if horizontal:
                fl = [BoundaryFace(row=cell_line, col=int(c), direction=normal) for c in cells[a:b]]
                seg = EdgeSegment(f"edge_{len(segments) + 1:04d}", float(lo), float(line), float(hi), float(line),
                                  normal, fl)
            else:
                fl = [BoundaryFace(row=int(r), col=cell_line, direction=normal) for r in cells[a:b]]
                seg = EdgeSegment(f"edge_{len(segments) + 1:04d}", float(line), float(lo), float(line), float(hi),
                                  normal, fl)
            segments.append(seg)

    # horizontal faces: the "down" face of row y-1 and the "up" face of row y share grid line y
    for y in range(ny + 1):
        if y >= 1:
            emit("down", y, np.nonzero(faces["down"][y - 1])[0], True, y - 1)
        if y < ny:
            emit("up", y, np.nonzero(faces["up"][y])[0], True, y)
    for x in range(nx + 1):
        if x < nx:
            emit("left", x, np.nonzero(faces["left"][:, x])[0], False, x)
        if x >= 1:
            emit("right", x, np.nonzero(faces["right"][:, x - 1])[0], False, x - 1)
    return segments


def create_intrinsic_geometry(mesh_size: float = 1.0, width: int = 120, height: int = 64) -> GeometryData:
    """Padded rectangle used as the reference's built-in geometry (geometry.py:245-262)."""
    mask = np.zeros((height, width), dtype=bool)
    pad_y = max(1, min(8, max(1, height // 4)))
    pad_x = max(1, min(8, max(1, width // 4)))
    if height - 2 * pad_y <= 0 or width - 2 * pad_x <= 0:
        mask[:, :] = True
    else:
        mask[pad_y:-pad_y, pad_x:-pad_x] = True
    return GeometryData(
        name="IntrinsicRectangle", source_path="intrinsic", layer=0, mesh_size=mesh_size,
        mask=mask.astype(int).tolist(), edges=extract_edge_segments(mask),
        bounds=[0.0, 0.0, float(width), float(height)],
    )
