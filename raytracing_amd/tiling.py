"""Row-band partition of a frame across ranks and its inverse.

The image is cut into bands of `band_rows` rows; band b belongs to rank
owner[b % len(owner)] - b % nranks when no pattern is given (rm_set_bands), any
deal of a run of bands with one (rm_set_band_pattern: a rank with other work gets
fewer).  A rank renders its bands packed top to bottom; `assemble` interleaves
the per-rank buffers back into the frame.  Pixels are independent, so the
assembled frame equals the single-rank frame byte for byte.
"""
import numpy as np

MAX_PERIOD = 64          # RM_MAX_BAND_PERIOD (include/rm_types.h)
MAX_OWN = 16             # bands of a run one rank may hold (kernel_params.h RM_MAX_BAND_OWN)


def deal(weights):
    """A run of sum(weights) bands dealt to ranks in proportion to the integer `weights`, each rank's bands
    spread evenly over the run (the costly middle of the image then falls to every rank alike): the k-th of
    rank r's w bands is due at (k + 1/2) / w of the run, and bands are handed out in the order they fall due."""
    weights = [int(w) for w in weights]
    if not weights or min(weights) < 0 or sum(weights) < 1 or sum(weights) > MAX_PERIOD or max(weights) > MAX_OWN:
        raise ValueError("band weights %r: need 1..%d bands in all, at most %d per rank" % (weights, MAX_PERIOD, MAX_OWN))
    due = sorted(((2 * k + 1) / (2.0 * w), r) for r, w in enumerate(weights) for k in range(w))
    return [r for _, r in due]


def lighter_rank0(nranks, rank0_share):
    """Integer weights (for `deal`) that give rank 0 about `rank0_share` of an even share (1.0: an even deal):
    w0 : w with w0 / w nearest to it among the runs that fit MAX_PERIOD and MAX_OWN."""
    if nranks == 1 or rank0_share >= 1.0:
        return [1] * nranks
    best = None
    for w in range(1, MAX_OWN + 1):
        for w0 in range(0, w + 1):
            if w0 + w * (nranks - 1) > MAX_PERIOD:
                continue
            err = abs(w0 / w - rank0_share)
            if best is None or err < best[0] - 1e-12:
                best = (err, w0, w)
    return [best[1]] + [best[2]] * (nranks - 1)


def _owner(nranks, owner):
    return list(range(nranks)) if owner is None else list(owner)


def band_of_rank(height, band_rows, rank, nranks, owner=None):
    """[(y0, y1)] image rows of the bands owned by `rank`, top to bottom."""
    owner = _owner(nranks, owner)
    out = []
    b = 0
    for y0 in range(0, height, band_rows):
        if owner[b % len(owner)] == rank:
            out.append((y0, min(height, y0 + band_rows)))
        b += 1
    return out


def local_rows(height, band_rows, rank, nranks, owner=None):
    return sum(y1 - y0 for y0, y1 in band_of_rank(height, band_rows, rank, nranks, owner))


def max_local_rows(height, band_rows, nranks, owner=None):
    return max(local_rows(height, band_rows, r, nranks, owner) for r in range(nranks))


def default_band_rows(height, nranks):
    """8 rows (one row of tiles): the finest deal, so the costly middle of the image is spread
    evenly over the ranks (chair 4K on 8 ranks, shares rendered one at a time on one GPU: the slowest
    rank is 8 % behind the mean with 16-row bands)."""
    return 8


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


def assemble(parts, height, band_rows, owner=None):
    """parts[r]: array [>=local_rows(r), width, C] -> frame [height, width, C] (numpy or torch)."""
    nranks = len(parts)
    first = parts[0]
    if hasattr(first, "new_zeros"):
        frame = first.new_zeros((height,) + tuple(first.shape[1:]))
    else:
        frame = np.zeros((height,) + first.shape[1:], dtype=first.dtype)
    for r in range(nranks):
        at = 0
        for y0, y1 in band_of_rank(height, band_rows, r, nranks, owner):
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
