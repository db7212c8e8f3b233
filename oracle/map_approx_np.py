"""map_approx_np.py -- restatement of PCL's OctreePointCloudSearch::approxNearestSearch as the reference's mapper uses it
(/root/reference/src/icpslam/octree_mapper.cpp:72-90), to QUANTIFY the one deliberate deviation of the map row (SURVEY.md
8(f4), EXPERIMENTS.md section 9-f4): libicpgpu returns the EXACT nearest map point, PCL a heuristic one.  TEST INFRASTRUCTURE ONLY.

       ***  PARITY UNPINNED  ***  (PCL is not under /root/reference; restated from PCL 1.8's octree_pointcloud.hpp /
       octree_search.hpp as published: adoptBoundingBoxToPoint, genOctreeKeyforPoint, approxNearestSearchRecursive)

What PCL does:
  * the bounding box starts as the first point +- resolution / 2, which getKeyBitSize() at once turns into a tree ONE level
    deep (max_voxels = max(ceil(extent / resolution), 2)) whose 2-voxel side is centred on that box: first point +- resolution;
    it then doubles (all three axes at once) whenever a point falls outside, towards the side the point is on; the lattice of
    leaf voxels never moves (the minimum moves by whole side lengths);
  * one point per leaf here (octree_mapper.cpp:62-69 only adds a point whose leaf is empty);
  * approxNearestSearch descends from the root: at every level it goes to the EXISTING child whose voxel centre is nearest to
    the query (float squared distance, first child on ties in child-index order x*4 + y*2 + z), and returns the point of the
    leaf it ends in -- never looking at any other leaf.
"""
from __future__ import annotations

import numpy as np

f32 = np.float32


class ApproxOctreeMap:
    """One-point-per-voxel map with PCL's octree geometry; `add_points` takes points already in the map frame."""

    def __init__(self, resolution: float = 0.5):
        self.res = float(resolution)
        self.defined = False
        self.min = np.zeros(3)          # min_x_, min_y_, min_z_ (double)
        self.max = np.zeros(3)
        self.depth = 0                  # octree_depth_: number of levels below the root; side = 2**depth voxels
        self.points: list[np.ndarray] = []
        self.leaf: dict[tuple[int, int, int], int] = {}   # leaf key -> point index

    # OctreePointCloud::adoptBoundingBoxToPoint
    def _adopt(self, p):
        eps = float(np.finfo(f32).eps)
        while True:
            if not self.defined:
                self.min = p.astype(np.float64) - self.res / 2
                self.max = p.astype(np.float64) + self.res / 2
                self._key_bit_size()
                self.defined = True
                continue
            lower = p < self.min
            upper = p >= self.max
            if not (lower.any() or upper.any()):
                return
            side = float(1 << self.depth) * self.res
            shift = np.where(~upper, side, 0.0)        # the old root becomes the UPPER child on the axes not violated above
            # every stored key moves with the minimum (the lattice itself does not)
            dk = (shift / self.res).round().astype(np.int64)
            if dk.any() and self.leaf:
                self.leaf = {(k[0] + int(dk[0]), k[1] + int(dk[1]), k[2] + int(dk[2])): v for k, v in self.leaf.items()}
            self.min = self.min - shift
            self.depth += 1
            self.max = self.min + (float(1 << self.depth) * self.res - eps)

    # OctreePointCloud::getKeyBitSize on an octree without leaves (the only call the mapper's usage reaches)
    def _key_bit_size(self):
        eps = float(np.finfo(f32).eps)
        max_key = int(np.ceil((self.max - self.min - eps) / self.res).max())
        max_voxels = max(max_key, 2)
        self.depth = int(np.ceil(np.log(float(max_voxels)) / np.log(2.0) - eps))
        side = float(1 << self.depth) * self.res
        oversize = (side - (self.max - self.min)) / 2.0
        grow = oversize > eps
        self.min = np.where(grow, self.min - oversize, self.min)
        self.max = np.where(grow, self.max + oversize, self.max)

    def _key(self, p):
        return tuple(((p.astype(np.float64) - self.min) / self.res).astype(np.int64))     # genOctreeKeyforPoint (truncation)

    def add_points(self, pts) -> int:
        added = 0
        for p in np.asarray(pts, f32)[:, :3]:
            if not np.isfinite(p).all():
                continue
            inside = self.defined and not ((p < self.min).any() or (p >= self.max).any())
            if inside and self._key(p) in self.leaf:      # isVoxelOccupiedAtPoint
                continue
            self._adopt(p)
            self.leaf[self._key(p)] = len(self.points)
            self.points.append(p.copy())
            added += 1
        self._levels = None
        return added

    def _prefix_sets(self):
        """occupied node keys per level (level d: keys >> (depth - d)), d = 1 .. depth"""
        if getattr(self, "_levels", None) is None:
            keys = np.array(list(self.leaf.keys()), np.int64).reshape(-1, 3)
            self._levels = [None] + [set(map(tuple, (keys >> (self.depth - d)).tolist())) for d in range(1, self.depth + 1)]
        return self._levels

    # OctreePointCloudSearch::approxNearestSearch
    def approx_nearest(self, q):
        q = np.asarray(q, f32)[:3]
        levels = self._prefix_sets()
        key = (0, 0, 0)
        for d in range(1, self.depth + 1):
            best, best_key = None, None
            size = self.res * float(1 << (self.depth - d))
            for child in range(8):
                nk = (2 * key[0] + ((child >> 2) & 1), 2 * key[1] + ((child >> 1) & 1), 2 * key[2] + (child & 1))
                if nk not in levels[d]:
                    continue
                centre = ((np.array(nk, np.float64) + 0.5) * size + self.min).astype(f32)   # genVoxelCenterFromOctreeKey
                diff = centre - q
                dist = f32(diff[0] * diff[0] + diff[1] * diff[1] + diff[2] * diff[2])
                if best is None or dist < best:
                    best, best_key = dist, nk
            key = best_key
        return self.leaf[key]

    def nn_indices_approx(self, queries) -> np.ndarray:
        return np.array([self.approx_nearest(q) for q in np.asarray(queries, f32)], np.int64)

    def map_points(self) -> np.ndarray:
        out = np.ones((len(self.points), 4), f32)
        if self.points:
            out[:, :3] = np.array(self.points, f32)
        return out
