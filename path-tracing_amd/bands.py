"""Partitioning of a frame's rows over ranks and the single gather of accumulator bands.

The split is INTERLEAVED: rank k of n takes every n-th tile row of 8 image rows, from rows 8 k on (pt_render_params::row_stride;
its buffers hold those tile rows packed).  Contiguous bands cost unequal amounts -- the rows through the torus are the expensive
ones: 67 ... 90 ms for the four bands of the 3840 x 2160 frame -- and a frame is as slow as its slowest band.  Images with fewer tile
rows than ranks keep contiguous bands.

Host-side Python used by bench.py (RCCL, device tensors) and by the CPU gloo tests: everything here is independent of
which integrator filled the bands.
"""
import numpy as np

BASE_W, BASE_H = 1920, 1080   # BASELINE.json configs[1]: one GPU's share under weak scaling


def frame_for(n_ranks):
    """Weak scaling: every rank keeps a 1920x1080-pixel band.  N = 4 gives the 3840x2160 frame of configs[3]."""
    if n_ranks % 4 == 0:
        return BASE_W * 2, BASE_H * 2 * (n_ranks // 4)
    return BASE_W, BASE_H * n_ranks


def band_rows(height, n_ranks, rank):
    """Contiguous rows [r0, r1) of rank `rank`; bands differ by at most one row and cover the image exactly."""
    base, extra = divmod(height, n_ranks)
    r0 = rank * base + min(rank, extra)
    return r0, r0 + base + (1 if rank < extra else 0)


TILE_ROWS = 8   # image rows of a tile row (csrc/pt_kernels.hip: kTileH)


def split(height, n_ranks, rank):
    """(row_begin, row_end, row_stride, buffer_rows) of rank `rank`: the arguments of its render call and the rows its buffers hold."""
    tile_rows = (height + TILE_ROWS - 1) // TILE_ROWS
    if n_ranks > 1 and tile_rows >= n_ranks:
        count = (tile_rows - rank + n_ranks - 1) // n_ranks
        return TILE_ROWS * rank, height, n_ranks, TILE_ROWS * count
    r0, r1 = band_rows(height, n_ranks, rank)
    return r0, r1, 1, r1 - r0


def image_rows(height, n_ranks, rank):
    """Image row of every buffer row of rank `rank`, in buffer order; -1 for buffer rows beyond the image (the last tile row's)."""
    r0, r1, stride, rows = split(height, n_ranks, rank)
    if stride == 1:
        return np.arange(r0, r1)
    out = np.concatenate([np.arange(t, t + TILE_ROWS) for t in range(r0, height, TILE_ROWS * stride)])
    assert len(out) == rows
    return np.where(out < height, out, -1)


def band_floats(width, rows):
    """Length of the packed band buffer: sum[3n] | sum2[3n] | count[n] (int32 bits) as float32 words."""
    return 7 * width * rows


def pack_band(s, s2, c):
    """numpy accumulators -> one float32 vector (count is bit-cast, not converted)."""
    return np.concatenate([np.ascontiguousarray(s, np.float32).ravel(), np.ascontiguousarray(s2, np.float32).ravel(),
                           np.ascontiguousarray(c, np.int32).view(np.float32).ravel()])


def unpack_band(buf, width, rows):
    n = width * rows
    buf = np.ascontiguousarray(buf, np.float32)
    return buf[:3 * n].reshape(n, 3), buf[3 * n:6 * n].reshape(n, 3), buf[6 * n:7 * n].view(np.int32)


def max_band_floats(width, height, n_ranks):
    return max(band_floats(width, split(height, n_ranks, r)[3]) for r in range(n_ranks))


def gather_bands(band, width, height, dist, rank, world, dst=0, out=None, async_op=False):
    """The one collective of a frame: every rank sends its packed band (a torch tensor, padded to the largest band)
    to `dst`.  Returns the list of per-rank tensors on `dst` (None elsewhere); with async_op=True returns
    (that list, work handle) and the caller must keep `band` untouched and call work.wait() before using the result.
    `out` lets `dst` reuse its receive buffers from frame to frame."""
    import torch
    n_max = max_band_floats(width, height, world)
    if band.numel() != n_max:
        padded = torch.zeros(n_max, dtype=band.dtype, device=band.device)
        padded[:band.numel()] = band
        band = padded
    if rank == dst and out is None:
        out = [torch.empty_like(band) for _ in range(world)]
    work = dist.gather(band, out if rank == dst else None, dst=dst, async_op=async_op)
    return (out, work) if async_op else out


def assemble(parts, width, height, world):
    """Per-rank packed bands (numpy, possibly padded) -> full-frame (sum, sum2, count)."""
    fs, fs2, fc = np.zeros((height, width, 3), np.float32), np.zeros((height, width, 3), np.float32), np.zeros((height, width), np.int32)
    for r in range(world):
        rows = split(height, world, r)[3]
        s, s2, c = unpack_band(np.asarray(parts[r])[:band_floats(width, rows)], width, rows)
        where = image_rows(height, world, r)
        inside = where >= 0
        fs[where[inside]] = s.reshape(rows, width, 3)[inside]
        fs2[where[inside]] = s2.reshape(rows, width, 3)[inside]
        fc[where[inside]] = c.reshape(rows, width)[inside]
    return fs.reshape(-1, 3), fs2.reshape(-1, 3), fc.ravel()
