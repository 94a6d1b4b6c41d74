"""Where a rank's emulator threads run: the CPUs of the NUMA node its GPU hangs off, split between the ranks that share
the node ("host cores partitioned NUMA-locally").

The reference steps its envs one after the other on the calling thread (``gym.vector.SyncVectorEnv``, reference
atari_env.py:241); here one process per GPU owns a pool of emulator threads, and eight such processes on one node must not
start 8 x 64 unpinned threads on 256 CPUs with their pinned staging wherever the allocator put it.  The policy:

* usable CPUs = scheduler affinity, capped by the cgroup CPU quota (a container that may use 16 of the 256 present);
* a rank's workers = usable // LOCAL_WORLD_SIZE (at least 1, at most 64; ``num_workers`` overrides);
* the rank's CPUs = those of its GPU's NUMA node (sysfs ``numa_node`` of the GPU's PCI address) that are in the affinity
  mask, divided core-wise between the ranks whose GPUs share that node; worker w is pinned to the w-th of them, first
  hardware threads of distinct cores first;
* the pinned staging buffers are allocated while the allocating thread is bound to the same CPUs (first touch on the
  GPU's node), see :func:`bound_to`.

Everything here is a pure function of a topology dictionary, so that tests can hand in a mocked host
(tests/test_hostplan_cpu.py: 8 ranks, 256 CPUs, 2 nodes).
"""
from __future__ import annotations

import contextlib
import os
from typing import Dict, List, Optional, Sequence


def parse_cpulist(text: str) -> List[int]:
    """"0-3,8,10-11" -> [0, 1, 2, 3, 8, 10, 11]"""
    out: List[int] = []
    for part in text.strip().split(","):
        part = part.strip()
        if not part:
            continue
        if "-" in part:
            a, b = part.split("-")
            out.extend(range(int(a), int(b) + 1))
        else:
            out.append(int(part))
    return out


def _read(path: str) -> Optional[str]:
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def cgroup_quota() -> Optional[float]:
    """CPU quota of this process's cgroup in CPUs (None: unlimited)."""
    txt = _read("/sys/fs/cgroup/cpu.max")
    if txt:
        a = txt.split()
        if a and a[0] != "max":
            try:
                return float(a[0]) / float(a[1])
            except (ValueError, IndexError, ZeroDivisionError):
                return None
        return None
    q, per = _read("/sys/fs/cgroup/cpu/cpu.cfs_quota_us"), _read("/sys/fs/cgroup/cpu/cpu.cfs_period_us")
    try:
        if q and per and float(q) > 0:
            return float(q) / float(per)
    except ValueError:
        pass
    return None


_TOPO = None


def read_topology(refresh: bool = False) -> Dict:
    """This host as the planner sees it: {"affinity": [cpu...], "quota": CPUs or None, "nodes": {node: [cpu...]},
    "core_of": {cpu: core key}}.  Read once per process (a few hundred sysfs files on a 256-CPU host)."""
    global _TOPO
    if _TOPO is None or refresh:
        _TOPO = _read_topology()
    return _TOPO


def _read_topology() -> Dict:
    try:
        affinity = sorted(os.sched_getaffinity(0))
    except AttributeError:
        affinity = list(range(os.cpu_count() or 1))
    nodes: Dict[int, List[int]] = {}
    base = "/sys/devices/system/node"
    try:
        for d in sorted(os.listdir(base)):
            if d.startswith("node") and d[4:].isdigit():
                txt = _read(os.path.join(base, d, "cpulist"))
                if txt is not None:
                    nodes[int(d[4:])] = parse_cpulist(txt)
    except OSError:
        pass
    core_of = {}
    for c in affinity:
        pk = _read(f"/sys/devices/system/cpu/cpu{c}/topology/physical_package_id")
        co = _read(f"/sys/devices/system/cpu/cpu{c}/topology/core_id")
        if pk is not None and co is not None:
            core_of[c] = (int(pk), int(co))
    return {"affinity": affinity, "quota": cgroup_quota(), "nodes": nodes, "core_of": core_of}


def usable_cpus(topo: Dict) -> int:
    n = len(topo["affinity"])
    q = topo.get("quota")
    return max(1, n if not q else min(n, int(q + 0.5)))


def gpu_numa_node(device_index: int) -> Optional[int]:
    """NUMA node of HIP device `device_index` (None: unknown / single-node host)."""
    import ctypes as C
    from . import _native as nat
    buf = C.create_string_buffer(32)
    if nat.lib().agx_device_pci_bus_id(int(device_index), buf, 32) != nat.OK:
        return None
    addr = buf.value.decode().lower()
    txt = _read(f"/sys/bus/pci/devices/{addr}/numa_node")
    try:
        node = int(txt) if txt is not None else -1
    except ValueError:
        node = -1
    return node if node >= 0 else None


def _core_order(cpus: Sequence[int], core_of: Dict) -> List[List[int]]:
    """The CPUs grouped by physical core (hardware threads of one core together), cores in ascending order of their first CPU."""
    groups: Dict = {}
    for c in sorted(cpus):
        groups.setdefault(core_of.get(c, ("cpu", c)), []).append(c)
    return sorted(groups.values(), key=lambda g: g[0])


def plan(topo: Dict, local_rank: int, local_world_size: int, gpu_nodes: Optional[Sequence[Optional[int]]] = None,
         workers: Optional[int] = None) -> Dict:
    """Placement of rank `local_rank` of `local_world_size` on this host.  gpu_nodes[r] = NUMA node of rank r's GPU (None:
    unknown).  Returns {"workers", "cpus" (the pin list, one per worker), "domain" (all CPUs of this rank's share),
    "numa_node", "usable", "per_rank"}."""
    lws = max(1, int(local_world_size))
    r = int(local_rank) % lws
    usable = usable_cpus(topo)
    per_rank = max(1, min(64, usable // lws))
    n_workers = per_rank if not workers else max(1, int(workers))
    aff = set(topo["affinity"])
    node = None
    if gpu_nodes is not None and r < len(gpu_nodes):
        node = gpu_nodes[r]
    node_cpus = [c for c in topo.get("nodes", {}).get(node, []) if c in aff] if node is not None else []
    if node_cpus:
        peers = [q for q in range(lws) if q < len(gpu_nodes) and gpu_nodes[q] == node]
    else:                                   # node unknown, or none of its CPUs allowed: all allowed CPUs, shared by every rank
        node, node_cpus, peers = None, sorted(aff), list(range(lws))
    k, m = peers.index(r), len(peers)
    cores = _core_order(node_cpus, topo.get("core_of", {}))
    lo, hi = len(cores) * k // m, len(cores) * (k + 1) // m
    mine = cores[lo:hi] if hi > lo else [cores[min(lo, len(cores) - 1)]]
    # first hardware thread of every core first, then the second threads: the first `workers` CPUs are distinct cores
    depth = max(len(g) for g in mine)
    domain = [g[t] for t in range(depth) for g in mine if t < len(g)]
    cpus = [domain[w % len(domain)] for w in range(n_workers)]
    return {"workers": n_workers, "cpus": cpus, "domain": domain, "numa_node": node, "usable": usable, "per_rank": per_rank}


def infer_peer_nodes(seen: Sequence[Optional[int]], local_rank: int, host_nodes: Sequence[int]) -> List[Optional[int]]:
    """NUMA nodes of all ranks' GPUs from what THIS process can see.  A launcher that masks the devices (one visible GPU per
    process) leaves the peers' entries None: if the ranks are spread over the host's nodes in blocks (ranks 0..3 on node 0, 4..7 on
    node 1 - the usual order) and that guess agrees with this rank's own node, the unknown peers are taken from it, so that a rank
    shares its node's CPUs with the ranks that are really there; otherwise an unknown peer is taken to sit beside this rank
    (disjoint CPU sets either way, at worst smaller than they could be)."""
    lws = len(seen)
    own = seen[local_rank] if 0 <= local_rank < lws else None
    guess = None
    if host_nodes and lws % len(host_nodes) == 0:
        per = lws // len(host_nodes)
        guess = [host_nodes[q // per] for q in range(lws)]
        if own is None or guess[local_rank] != own or any(v is not None and v != g for v, g in zip(seen, guess)):
            guess = None
    return [v if v is not None else (guess[q] if guess is not None else own) for q, v in enumerate(seen)]


def plan_for_process(device_index: int, workers: Optional[int] = None, topo: Optional[Dict] = None) -> Dict:
    """The plan of THIS process: LOCAL_RANK / LOCAL_WORLD_SIZE from the launcher's environment (one process per GPU), the
    GPUs' NUMA nodes from sysfs.  AGX_NO_PIN=1 (or an unreadable topology) yields an unpinned plan (cpus = [])."""
    topo = read_topology() if topo is None else topo
    lws = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1") or 1))
    lr = int(os.environ.get("LOCAL_RANK", "0") or 0) % lws
    # rank r drives device r of the node, except in a one-process job, which may sit on any device (and in rehearsals that let
    # several ranks share one device: a rank whose device ordinal does not exist is taken to sit beside this one)
    own = gpu_numa_node(device_index)
    nodes = infer_peer_nodes([own if (lws == 1 or q == lr) else gpu_numa_node(q) for q in range(lws)], lr, sorted(topo.get("nodes", {})))
    p = plan(topo, lr, lws, nodes, workers)
    if os.environ.get("AGX_NO_PIN") == "1":
        p["cpus"] = []
    return p


@contextlib.contextmanager
def bound_to(cpus: Sequence[int]):
    """Run the body with the calling thread bound to `cpus` (pinned host allocations made inside are first-touched on that
    NUMA node), then restore the previous affinity.  No-op for an empty list or where the call is refused."""
    old = None
    if cpus:
        try:
            old = os.sched_getaffinity(0)
            os.sched_setaffinity(0, set(cpus))
        except (AttributeError, OSError):
            old = None
    try:
        yield
    finally:
        if old is not None:
            try:
                os.sched_setaffinity(0, old)
            except OSError:
                pass
