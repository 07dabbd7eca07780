"""CPU placement of one self-play process per GPU.

A rank's search threads, its evaluation threads and the HIP runtime's own threads share the host with the other
ranks' (the reference runs one process with ``threads`` workers, training/self-play/src/self_play.rs:109-137; with
one process per GPU there are up to eight of those).  Left alone, the scheduler migrates them across sockets and
away from the memory their page-locked batch buffers live in.  ``pin_rank`` gives every local rank a DISJOINT set
of the CPUs this job may use, preferring the CPUs of the NUMA node its GPU hangs off (``local_cpulist`` of the
GPU's PCI device in sysfs); ranks whose GPUs share a node split that node's CPUs among themselves.  Threads
created afterwards inherit the mask.  Pure host logic: no GPU call, testable on CPU with a fake sysfs root.
"""

from __future__ import annotations

import os
from pathlib import Path


def parse_cpulist(text: str) -> list[int]:
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11]"""
    out: list[int] = []
    for part in text.strip().split(","):
        if not part:
            continue
        if "-" in part:
            lo, hi = part.split("-")
            out.extend(range(int(lo), int(hi) + 1))
        else:
            out.append(int(part))
    return out


def gpu_local_cpus(pci_bus_ids: list[str], sysfs: str = "/sys/bus/pci/devices") -> list[list[int] | None]:
    """CPUs local to each GPU (by PCI address 'dddd:bb:dd.f'), or None where sysfs does not say."""
    out: list[list[int] | None] = []
    for bdf in pci_bus_ids:
        try:
            cpus = parse_cpulist((Path(sysfs) / bdf.lower() / "local_cpulist").read_text())
            out.append(cpus or None)
        except (OSError, ValueError):
            out.append(None)
    return out


def plan(world_local: int, allowed: list[int], local_cpus: list[list[int] | None]) -> list[list[int]]:
    """CPU set of every local rank: disjoint, covering at most `allowed`, each rank inside its GPU's node where known.

    Ranks are grouped by their GPU's local CPU list (unknown = one group over all allowed CPUs); a group's CPUs
    (its list cut to the allowed ones, minus what closer groups already took) are dealt to its ranks in contiguous,
    equal slices.  A group left without CPUs falls back to an even slice of everything allowed.
    """
    allowed = sorted(allowed)
    groups: dict[tuple, list[int]] = {}
    for r in range(world_local):
        key = tuple(local_cpus[r]) if r < len(local_cpus) and local_cpus[r] else ()
        groups.setdefault(key, []).append(r)
    out: list[list[int] | None] = [None] * world_local
    taken: set[int] = set()
    # specific groups first, the catch-all group last
    for key in sorted(groups, key=lambda k: (len(k) == 0, k)):
        ranks = groups[key]
        pool = [c for c in (key or allowed) if c in set(allowed) and c not in taken]
        per = len(pool) // len(ranks)
        if per == 0:
            continue
        for i, r in enumerate(ranks):
            out[r] = pool[i * per : (i + 1) * per]
            taken.update(out[r])
    for r in range(world_local):  # fallback: an even slice of the allowed CPUs
        if not out[r]:
            per = max(1, len(allowed) // world_local)
            out[r] = allowed[(r * per) % len(allowed) : (r * per) % len(allowed) + per] or allowed
    return out  # type: ignore[return-value]


def pin_rank(local_rank: int, world_local: int, pci_bus_ids: list[str] | None = None) -> list[int]:
    """Restrict this process (and the threads it creates from now on) to its share of the CPUs; returns the share.
    With one rank nothing is changed."""
    allowed = sorted(os.sched_getaffinity(0))
    if world_local <= 1:
        return allowed
    local = gpu_local_cpus(pci_bus_ids) if pci_bus_ids else [None] * world_local
    mine = plan(world_local, allowed, local)[local_rank]
    try:
        os.sched_setaffinity(0, mine)
    except OSError:
        return allowed
    return mine


def torch_pci_bus_ids(n: int) -> list[str] | None:
    """PCI addresses of HIP devices 0..n-1 as torch reports them (no GPU context is created), or None."""
    try:
        import torch

        out = []
        for i in range(n):
            p = torch.cuda.get_device_properties(i)
            out.append(f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0")
        return out
    except Exception:  # noqa: BLE001 - any failure means "unknown topology"
        return None
