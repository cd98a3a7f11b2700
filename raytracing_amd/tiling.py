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


def default_band_rows(height, nranks):
    """16 rows (one workgroup row): the finest deal, so the costly middle of the
    image is spread evenly over the ranks."""
    return 16


def bands_per_rank(height, band_rows, nranks):
    nbands = (height + band_rows - 1) // band_rows
    return (nbands + nranks - 1) // nranks


def padded_rows(height, band_rows, nranks):
    """Rows of a frame buffer in which every rank owns the same number of whole
    bands; the rows past `height` are never written by a kernel."""
    return bands_per_rank(height, band_rows, nranks) * nranks * band_rows


def assemble_into(frame_padded, parts, band_rows):
    """One strided device copy: parts [nranks, bands_per_rank*band_rows, W, C]
    (the gather's output on rank 0) -> frame_padded [padded_rows, W, C]."""
    nranks, rows, width, ch = parts.shape
    nb = rows // band_rows
    dst = frame_padded.view(nb, nranks, band_rows, width, ch)
    dst.copy_(parts.view(nranks, nb, band_rows, width, ch).permute(1, 0, 2, 3, 4))
    return frame_padded


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


def expand_luminance(lum):
    """Host mirror of rm_assemble_luminance's pixel: luminance plane [H, W] u8 -> RGBA8 [H, W, 4],
    the material scaling of Source/cuda_kernels.cu:695-699 in fp32 (0.5, 0.7, 1.0; alpha 255)."""
    import numpy as np
    L = np.asarray(lum, dtype=np.uint8).astype(np.float32)
    out = np.empty(L.shape + (4,), dtype=np.uint8)
    out[..., 0] = (L * np.float32(0.5)).astype(np.uint8)          # truncation, as the kernel's cast
    out[..., 1] = (L * np.float32(0.7)).astype(np.uint8)
    out[..., 2] = L.astype(np.uint8)
    out[..., 3] = 255
    return out
