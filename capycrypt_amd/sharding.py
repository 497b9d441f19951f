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


def cpu_list_string(cpus):
    """[0, 1, 2, 3, 8, 9] -> "0-3,8-9" (the kernel's cpulist notation)"""
    out, cpus = [], sorted(cpus)
    i = 0
    while i < len(cpus):
        j = i
        while j + 1 < len(cpus) and cpus[j + 1] == cpus[j] + 1:
            j += 1
        out.append(str(cpus[i]) if i == j else "%d-%d" % (cpus[i], cpus[j]))
        i = j + 1
    return ",".join(out)


def device_topology(device):
    """include/capyhip.h: capy_device_topology -- {"device", "pci_bus_id", "numa_node", "cpus" (cpulist string of the CPUs a worker
    of this device pins itself to: the device's local_cpulist within this process's affinity; "" = no pinning), "n_cpus"}"""
    import ctypes as C

    from . import _lib

    bdf = C.create_string_buffer(64)
    numa = C.c_int(-1)
    cap = 4096
    cpus = (C.c_int * cap)()
    n = _lib.lib().capy_device_topology(int(device), bdf, 64, C.byref(numa), cpus, cap)
    if n < 0:
        _lib.check(n)
    ids = list(cpus[:min(n, cap)])
    return {"device": int(device), "pci_bus_id": bdf.value.decode(), "numa_node": int(numa.value),
            "cpus": cpu_list_string(ids), "n_cpus": int(n), "cpu_ids": ids}
