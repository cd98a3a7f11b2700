"""Frames of a sequence handed to contexts in launches of several frames (rm_set_batch).

bench.py's render loop, kept apart from it so that the order of C-ABI calls a run makes can be
checked without a GPU: a launch takes its frame count from rm_set_info_batch, so a last launch that
is not full calls nothing that allocates, waits or forgets the tile order (rm_set_batch is called
once per context, before anything is timed).
"""
import numpy as np


class FrameBatcher:
    """step(info) collects camera blocks; every `batch` of them (and flush(), for what is left) becomes
    one launch of the next context in turn: before(j) -> set_info_batch -> process_async -> after(j, n)."""

    def __init__(self, ctxs, batch, before=None, after=None):
        self.ctxs, self.batch = list(ctxs), int(batch)
        self.before, self.after = before, after
        self.ctx = 0                  # the context the next launch goes to
        # camera blocks of the launch being collected, as raw bytes in a buffer made once: numpy takes 40 us
        # per block to concatenate structured arrays, and a launch's host side is 3 us
        self.stage = None
        self.pending = 0
        self.launches = 0
        self.last = None              # (context index, slot of the last frame in its launch)
        self.sizes = []               # launch sizes still to come of a planned run (plan)

    def plan(self, total):
        """A run of `total` frames known in advance goes out in the fewest launches the batch allows, all of
        nearly the same size (20 frames at up to 8 per launch: 7, 7, 6, not 8, 8, 4): a short last launch has
        the same long tiles to wait for as a full one and fewer cheap tiles to run beside them."""
        n = -(-int(total) // self.batch)
        self.sizes = [int(total) // n + (1 if k < int(total) % n else 0) for k in range(n)] if n else []

    def step(self, info):
        info = np.asarray(info)
        if self.stage is None:
            self.stage = np.zeros(self.batch, dtype=info.dtype)
            self.bytes = self.stage.view(np.uint8).reshape(self.batch, info.dtype.itemsize)
        self.bytes[self.pending] = info.reshape(-1)[:1].view(np.uint8)
        self.pending += 1
        if self.pending == (self.sizes[0] if self.sizes else self.batch):
            self.flush()

    def flush(self):
        n = self.pending
        if n == 0:
            return
        j = self.ctx
        if self.before:
            self.before(j)
        rm = self.ctxs[j]
        rm.set_info_batch(self.stage[:n])                     # n <= batch cameras: this launch renders n frames
        rm.process_async()
        self.launches += 1
        if self.sizes:
            self.sizes.pop(0)
        self.last = (j, n - 1)
        self.ctx, self.pending = (j + 1) % len(self.ctxs), 0
        if self.after:
            self.after(j, n)


def run_blocks(run, info_of, steps, repeats, mark, first=0):
    """The measured part of bench.py: one block of `steps` frames that is not timed, then `repeats` timed blocks
    of the same shape straight behind it, every block planned into the same launches (FrameBatcher.plan).
    Nothing here waits for the device, collects garbage or allocates on it: a block's launches are queued
    directly behind the block before, so no block but the first starts on an idle chip (round 2's single timed
    launch followed a drain and read 17 % slow on the driver's box).  `mark()` is called after each block's last
    launch has been queued (it records events on the streams; it must not wait) and its results are returned:
    repeats + 1 marks, block r's time is mark r+1 against mark r.  Returns (marks, next frame number)."""
    marks, k = [], first
    for _ in range(repeats + 1):
        run.plan(steps)
        for _ in range(steps):
            run.step(info_of(k))
            k += 1
        run.flush()
        marks.append(mark())
    return marks, k
