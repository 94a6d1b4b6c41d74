"""CPU: the per-rank host partition (active_gym/hostplan.py, SURVEY.md 8e "host cores partitioned NUMA-locally") on mocked
topologies - the box the 8-GPU run lands on has 2 sockets x 64 cores x 2 threads and one process per GPU - and the same
default inside libagx_runner.so (agxr_default_threads: usable CPUs // LOCAL_WORLD_SIZE)."""
import importlib.util
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _mi355x_node(quota=None, affinity=None):
    """2 NUMA nodes, 128 cores, SMT2: node 0 = CPUs 0-63 + 128-191, node 1 = 64-127 + 192-255 (what the GPU boxes report)."""
    nodes = {0: list(range(0, 64)) + list(range(128, 192)), 1: list(range(64, 128)) + list(range(192, 256))}
    core_of = {c: (0 if (c % 128) < 64 else 1, c % 128) for c in range(256)}
    return {"affinity": list(range(256)) if affinity is None else list(affinity), "quota": quota, "nodes": nodes, "core_of": core_of}


def test_parse_cpulist():
    from active_gym.hostplan import parse_cpulist
    assert parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    assert parse_cpulist("") == [] and parse_cpulist("5") == [5]


def test_eight_ranks_two_nodes_disjoint_and_node_local():
    from active_gym.hostplan import plan
    topo = _mi355x_node()
    gpu_nodes = [0, 0, 0, 0, 1, 1, 1, 1]
    plans = [plan(topo, r, 8, gpu_nodes) for r in range(8)]
    seen = set()
    for r, p in enumerate(plans):
        assert p["numa_node"] == gpu_nodes[r] and p["usable"] == 256 and p["per_rank"] == 32 and p["workers"] == 32
        assert len(p["cpus"]) == 32 and len(set(p["cpus"])) == 32
        assert set(p["domain"]) <= set(topo["nodes"][gpu_nodes[r]])             # node-local
        assert not (set(p["domain"]) & seen)                                     # disjoint between ranks
        seen |= set(p["domain"])
        # 16 cores x 2 threads per rank; the 32 workers land on 16 first threads, then their 16 siblings
        cores = {topo["core_of"][c] for c in p["cpus"][:16]}
        assert len(cores) == 16 and {topo["core_of"][c] for c in p["cpus"][16:]} == cores
    assert sum(p["workers"] for p in plans) <= 256 and seen == set(range(256))


def test_cgroup_quota_caps_the_workers_and_keeps_them_on_distinct_cores():
    from active_gym.hostplan import plan
    topo = _mi355x_node(quota=16.0)                                              # the one-GPU job's container: 16 of 256
    p = plan(topo, 0, 1, [0])
    assert p["usable"] == 16 and p["workers"] == 16 and p["numa_node"] == 0
    assert len({topo["core_of"][c] for c in p["cpus"]}) == 16 and set(p["cpus"]) <= set(topo["nodes"][0])
    # eight ranks under one 64-CPU quota: 8 workers each, the sum stays inside the quota
    topo = _mi355x_node(quota=64.0)
    plans = [plan(topo, r, 8, [0, 0, 0, 0, 1, 1, 1, 1]) for r in range(8)]
    assert [p["workers"] for p in plans] == [8] * 8 and sum(p["workers"] for p in plans) <= 64
    assert all(len(set(p["cpus"])) == 8 for p in plans)


def test_override_unknown_node_and_restricted_affinity():
    from active_gym.hostplan import plan
    topo = _mi355x_node()
    p = plan(topo, 3, 8, [0, 0, 0, 0, 1, 1, 1, 1], workers=5)                    # num_workers stays an override
    assert p["workers"] == 5 and len(p["cpus"]) == 5
    # unknown GPU node: every allowed CPU, split between all ranks
    plans = [plan(topo, r, 4, [None] * 4) for r in range(4)]
    assert all(p["numa_node"] is None for p in plans)
    doms = [set(p["domain"]) for p in plans]
    assert all(not (doms[i] & doms[j]) for i in range(4) for j in range(i))
    # an affinity mask that excludes the GPU's node altogether falls back to the allowed CPUs
    topo2 = _mi355x_node(affinity=range(64, 72))
    p = plan(topo2, 0, 1, [0])
    assert p["numa_node"] is None and set(p["cpus"]) <= set(range(64, 72)) and p["workers"] == 8
    # more workers than CPUs in the share: wrap around, never leave the share
    p = plan(topo2, 0, 1, [1], workers=20)
    assert len(p["cpus"]) == 20 and set(p["cpus"]) == set(range(64, 72)) and p["numa_node"] == 1
    # more ranks than cores on the node: every rank still gets a CPU of the node
    tiny = {"affinity": [0, 1], "quota": None, "nodes": {0: [0, 1]}, "core_of": {}}
    assert [plan(tiny, r, 4, [0] * 4)["cpus"] for r in range(4)] == [[0], [0], [1], [1]]


def test_peers_behind_a_device_mask_are_placed_by_rank_blocks():
    """One visible GPU per process (a launcher that masks devices): the peers' NUMA nodes are unknown.  With ranks in node blocks
    the guess gives every rank its true share; where the guess contradicts what is known, unknown peers sit beside this rank."""
    from active_gym.hostplan import infer_peer_nodes, plan
    topo = _mi355x_node()
    truth = [0, 0, 0, 0, 1, 1, 1, 1]
    seen_total = set()
    for r in range(8):
        seen = [truth[q] if q == r else None for q in range(8)]
        nodes = infer_peer_nodes(seen, r, [0, 1])
        assert nodes == truth
        p = plan(topo, r, 8, nodes)
        assert len(p["domain"]) == 32 and not (set(p["domain"]) & seen_total)
        seen_total |= set(p["domain"])
    # own node contradicts the block guess (rank 1 on node 1): no guess, peers beside this rank
    assert infer_peer_nodes([None, 1, None, None], 1, [0, 1]) == [1, 1, 1, 1]
    # a visible peer contradicts it
    assert infer_peer_nodes([0, 1, None, None], 0, [0, 1]) == [0, 1, 0, 0]
    # nothing known at all, one rank, more nodes than ranks
    assert infer_peer_nodes([None, None], 0, [0, 1]) == [None, None]
    assert infer_peer_nodes([None], 0, [0, 1]) == [None]
    assert infer_peer_nodes([0, None, None], 0, [0, 1]) == [0, 0, 0]


def test_read_topology_and_bound_to_on_this_host():
    from active_gym import hostplan
    topo = hostplan.read_topology(refresh=True)
    assert topo["affinity"] and hostplan.usable_cpus(topo) >= 1
    before = os.sched_getaffinity(0)
    with hostplan.bound_to([min(before)]):
        assert os.sched_getaffinity(0) == {min(before)}
    assert os.sched_getaffinity(0) == before
    with hostplan.bound_to([]):
        assert os.sched_getaffinity(0) == before


def test_runner_default_threads_follow_quota_and_local_world_size():
    spec = importlib.util.spec_from_file_location("agx_build", os.path.join(REPO, "active-gym_amd", "build.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    m.build_runner()
    code = ("import sys, ctypes; sys.path.insert(0, %r); from active_gym import native_runner as nr; l = nr.lib(); "
            "o = (ctypes.c_int32 * 3)(); l.agxr_host_cpus(ctypes.byref(o)); print(l.agxr_default_threads(), o[0], o[1], o[2])"
            % os.path.join(REPO, "active-gym_amd"))

    def run(lws):
        env = dict(os.environ)
        env.pop("LOCAL_WORLD_SIZE", None)
        if lws:
            env["LOCAL_WORLD_SIZE"] = str(lws)
        return [int(v) for v in subprocess.check_output([sys.executable, "-c", code], env=env, text=True).split()]

    from active_gym import hostplan
    usable = hostplan.usable_cpus(hostplan.read_topology(refresh=True))
    d1, aff, quota, lws = run(None)
    assert lws == 1 and aff == len(os.sched_getaffinity(0)) and d1 == max(1, min(64, usable))
    d4, _, _, lws4 = run(4)
    assert lws4 == 4 and d4 == max(1, min(64, usable // 4))
