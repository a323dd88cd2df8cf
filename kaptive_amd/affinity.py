"""NUMA placement of the per-device processes (``kaptive assembly --devices a,b,...``; ``bench.py --gpus N``).

One process per GPU, no collective: what the ranks share is the host -- reader threads, page-locked
shards, the PCIe root ports.  On a two-socket node a rank whose threads and pinned buffers live on the other socket sends
every upload across the inter-socket link.  So a rank, before it allocates anything large:

  1. asks which NUMA node its device hangs off (``kp_device_numa_node``: the PCI bus id of the HIP device ->
     ``/sys/bus/pci/devices/<id>/numa_node``; asked in a process of its own, ``device_numa_nodes``),
  2. takes its share of the CPUs the process was granted (affinity mask) on that node -- the devices of one node split that
     node's CPUs evenly; devices whose node is unknown or has no granted CPU share what no other device claimed --,
  3. ``sched_setaffinity`` to that set: its reader threads inherit it, and the huge-page blocks of ``kp_host_reserve`` are first
     touched -- hence placed -- on that node.

On a one-node box (the 1-GPU benchmark box) the plan is the whole mask and nothing changes.  ``KAPTIVE_AMD_NUMA=0`` turns
the placement off.  The reference types genomes in one serial loop and has nothing to place
(src/kaptive/serotyping/cli.py:196-208).
"""

from __future__ import annotations

import os
from pathlib import Path


def _parse_cpulist(text: str) -> list[int]:
    out: list[int] = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        out.extend(range(int(lo), int(hi or lo) + 1))
    return out


def cpu_nodes(sys_root: str = "/sys/devices/system/node") -> dict[int, int]:
    """cpu -> NUMA node, from the kernel's node directories (empty when there are none)."""
    out: dict[int, int] = {}
    try:
        for d in sorted(Path(sys_root).glob("node[0-9]*")):
            node = int(d.name[4:])
            for c in _parse_cpulist((d / "cpulist").read_text()):
                out[c] = node
    except (OSError, ValueError):
        return {}
    return out


def device_numa_nodes(devices: list[int]) -> list[int | None]:
    """NUMA node of every HIP device in ``devices`` (None: unknown -- no sysfs entry, node -1, or no such device).  Asked in a
    short-lived process of its own: the answer needs the HIP runtime (device index -> PCI bus id), and the caller may be about
    to fork workers or has not chosen its device yet -- neither goes well with a runtime already initialised here."""
    import json
    import subprocess
    import sys

    lib = Path(__file__).resolve().parent / "libkaptive_amd.so"
    code = ("import ctypes, json, sys; h = ctypes.CDLL(sys.argv[1]); "
            "print(json.dumps([int(h.kp_device_numa_node(int(d))) for d in sys.argv[2:]]))")
    try:
        r = subprocess.run([sys.executable, "-c", code, str(lib), *map(str, devices)], capture_output=True, text=True, timeout=60)
        nodes = json.loads(r.stdout.strip().splitlines()[-1])
        return [n if n >= 0 else None for n in nodes]
    except Exception:  # noqa: BLE001  (placement is an optimisation: never a reason to fail)
        return [None] * len(devices)


def plan(granted: list[int], device_nodes: list[int | None], cpu_node: dict[int, int]) -> list[list[int]]:
    """The CPUs of every device's process: a PARTITION of ``granted`` (every granted CPU belongs to exactly one device, as
    long as there are at least as many CPUs as devices).  Devices of a node split the granted CPUs of that node in order;
    CPUs of nodes without a device, and devices without CPUs of their own, are matched up afterwards, evenly."""
    n_dev = len(device_nodes)
    if n_dev == 0:
        return []
    granted = sorted(set(granted))
    shares: list[list[int]] = [[] for _ in range(n_dev)]
    by_node: dict[int | None, list[int]] = {}
    for c in granted:
        by_node.setdefault(cpu_node.get(c), []).append(c)
    claimed: set[int] = set()
    for node in sorted({n for n in device_nodes if n is not None}):
        devs = [i for i, n in enumerate(device_nodes) if n == node]
        cpus = by_node.get(node, [])
        for k, i in enumerate(devs):  # contiguous slices: hyperthread siblings and shared caches stay together
            lo, hi = len(cpus) * k // len(devs), len(cpus) * (k + 1) // len(devs)
            shares[i] = cpus[lo:hi]
            claimed.update(shares[i])
    left = [c for c in granted if c not in claimed]
    needy = [i for i in range(n_dev) if not shares[i]]
    if needy:  # unknown node, or a node none of whose CPUs was granted: an even share of what nobody claimed ...
        for k, i in enumerate(needy):
            shares[i] = left[len(left) * k // len(needy) : len(left) * (k + 1) // len(needy)]
        left = []
        fair = len(granted) // n_dev
        for i in needy:  # ... topped up from the largest shares when nothing (or too little) was left unclaimed
            while len(shares[i]) < fair:
                donor = max(range(n_dev), key=lambda j: len(shares[j]))
                if len(shares[donor]) <= len(shares[i]) + 1:
                    break
                shares[i].append(shares[donor].pop())
    for k, c in enumerate(left):  # CPUs of nodes without a device: dealt out, so that the plan stays a partition
        shares[k % n_dev].append(c)
    if any(not s for s in shares):  # fewer CPUs than devices: everybody shares everything
        return [list(granted) for _ in range(n_dev)]
    return [sorted(s) for s in shares]


def place(device: int, devices: list[int] | None = None, nodes: list[int | None] | None = None) -> dict:
    """Pins the calling process to its device's share of the granted CPUs (see the module text) and says what it did.
    ``devices``: all devices of the run (the shares depend on who else is there); one device alone keeps the whole mask and
    nothing is asked or changed.  ``nodes``: their NUMA nodes when the caller already knows them (the command line asks once
    for all its workers)."""
    info: dict = {"device": device, "numa_node": None, "cpus": None, "applied": False}
    if os.environ.get("KAPTIVE_AMD_NUMA", "1") == "0" or not hasattr(os, "sched_setaffinity"):
        return info
    devices = list(devices) if devices else [device]
    if device not in devices:
        devices.append(device)
        nodes = None
    if len(devices) < 2:
        return info
    granted = sorted(os.sched_getaffinity(0))
    if nodes is None or len(nodes) != len(devices):
        nodes = device_numa_nodes(devices)
    shares = plan(granted, nodes, cpu_nodes())
    mine = shares[devices.index(device)]
    info.update(numa_node=nodes[devices.index(device)], cpus=mine)
    if mine and set(mine) != set(granted):
        try:
            os.sched_setaffinity(0, mine)
            info["applied"] = True
        except OSError:
            pass
    return info
