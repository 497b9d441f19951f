"""Multi-GPU sharding of a batch: independent items, contiguous index ranges, no collective.

SURVEY.md §8e: every message / (scalar, point) pair is independent, so rank r of N owns a contiguous
slice of the batch, balanced by BYTES when item sizes differ, and writes its slice of the outputs.
Output order is input order for every N.  The only cross-rank step is the barrier + max-over-ranks
timing in bench.py; no data-path collective exists.
"""


def shard_range(n_items, rank, world):
    """Contiguous, count-balanced slice [lo, hi) of range(n_items) for `rank` of `world`."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return lo, hi


def shard_by_bytes(lengths, world):
    """Contiguous slices balanced by total bytes. Returns a list of (lo, hi) per rank."""
    total = sum(lengths)
    bounds = [0]
    acc = 0
    i = 0
    n = len(lengths)
    for r in range(1, world):
        target = total * r / world
        while i < n and acc + lengths[i] / 2 <= target:
            acc += lengths[i]
            i += 1
        bounds.append(i)
    bounds.append(n)
    return [(bounds[r], bounds[r + 1]) for r in range(world)]


def sharded_map(fn, items, rank, world):
    """Apply the batched operator `fn` to this rank's slice of `items`; returns (lo, hi, results)."""
    lo, hi = shard_range(len(items), rank, world)
    return lo, hi, fn(items[lo:hi])
