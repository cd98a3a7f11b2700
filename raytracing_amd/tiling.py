"""Row-band partition of a frame across ranks and its inverse.

The image is cut into bands of `band_rows` rows; band b belongs to rank
b % nranks.  A rank renders its bands packed top to bottom (rm_set_bands);
`assemble` interleaves the per-rank buffers back into the frame.  Pixels are
independent, so the assembled frame equals the single-rank frame byte for byte.
"""
import numpy as np


def band_of_rank(height, band_rows, rank, nranks):
    """[(y0, y1)] image rows of the bands owned by `rank`, top to bottom."""
    out = []
    b = 0
    for y0 in range(0, height, band_rows):
        if b % nranks == rank:
            out.append((y0, min(height, y0 + band_rows)))
        b += 1
    return out


def local_rows(height, band_rows, rank, nranks):
    return sum(y1 - y0 for y0, y1 in band_of_rank(height, band_rows, rank, nranks))


def max_local_rows(height, band_rows, nranks):
    return max(local_rows(height, band_rows, r, nranks) for r in range(nranks))


def default_band_rows(height, nranks, target_bands_per_rank=8):
    """Bands of a multiple of 16 rows, about `target_bands_per_rank` per rank, so
    the costly middle of the image is dealt evenly."""
    rows = height // max(1, nranks * target_bands_per_rank)
    return max(16, (rows // 16) * 16)


def assemble(parts, height, band_rows):
    """parts[r]: array [>=local_rows(r), width, C] -> frame [height, width, C] (numpy or torch)."""
    nranks = len(parts)
    first = parts[0]
    if hasattr(first, "new_zeros"):
        frame = first.new_zeros((height,) + tuple(first.shape[1:]))
    else:
        frame = np.zeros((height,) + first.shape[1:], dtype=first.dtype)
    for r in range(nranks):
        at = 0
        for y0, y1 in band_of_rank(height, band_rows, r, nranks):
            frame[y0:y1] = parts[r][at:at + (y1 - y0)]
            at += y1 - y0
    return frame
