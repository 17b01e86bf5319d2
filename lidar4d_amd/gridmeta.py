"""Host-side geometry of tiny-cuda-nn's multi-resolution hash grid: per-level scale,
resolution, entry count, offset and dense/hashed addressing, derived in float32 the way tiny-cuda-nn's
host code does, and packed into the C ABI's ``l4d_grid_desc``."""
import numpy as np

from ._lib import GridDesc, L4D_MAX_LEVELS


class GridMeta:
    def __init__(self, n_dims, n_levels, n_features, log2_hashmap_size, base_resolution, per_level_scale):
        if n_levels > L4D_MAX_LEVELS:
            raise ValueError(f"at most {L4D_MAX_LEVELS} hash levels are supported")
        f32 = np.float32
        self.n_dims, self.n_levels, self.n_features = int(n_dims), int(n_levels), int(n_features)
        log2_pls = f32(np.log2(f32(per_level_scale)))
        self.scale, self.res, self.size, self.offset, self.hashed = [], [], [], [], []
        off = 0
        for lvl in range(self.n_levels):
            scale = f32(f32(np.exp2(f32(lvl) * log2_pls)) * f32(base_resolution) - f32(1.0))
            res = int(np.ceil(scale)) + 1
            max_params = (2 ** 32 - 1) // 2
            dense = res ** self.n_dims
            n = max_params if float(dense) > float(max_params) else dense
            n = (n + 7) // 8 * 8
            n = min(n, 1 << int(log2_hashmap_size))
            stride = 1
            for _ in range(self.n_dims):  # tiny-cuda-nn grid_index(): uint32 stride walk
                if stride > n:
                    break
                stride = (stride * res) & 0xFFFFFFFF
            self.scale.append(float(scale))
            self.res.append(res)
            self.size.append(n)
            self.offset.append(off)
            self.hashed.append(bool(n < stride))
            off += n
        self.n_entries = off
        self.n_params = off * self.n_features
        self.n_output_dims = self.n_levels * self.n_features

    def desc(self):
        d = GridDesc()
        d.n_dims, d.n_features, d.n_levels = self.n_dims, self.n_features, self.n_levels
        mask = 0
        for l in range(self.n_levels):
            d.scale[l], d.res[l], d.size[l], d.offset[l] = self.scale[l], self.res[l], self.size[l], self.offset[l]
            if self.hashed[l]:
                mask |= 1 << l
        d.hashed_mask = mask
        return d
