"""Host-side placement of a rank next to its GPU (visualslam_amd/cxx/gpu_locality.hpp): the sysfs side
runs without a GPU - `Stream --numa-probe` resolves a PCI address under a FAKE sysfs tree, reports the node's CPUs that this
process may use, and (with --numa-bind) narrows its own affinity to them."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "visualslam_amd", "bin", "Stream")


def cpulist(cpus):
    cpus = sorted(cpus)
    out, i = [], 0
    while i < len(cpus):
        j = i
        while j + 1 < len(cpus) and cpus[j + 1] == cpus[j] + 1:
            j += 1
        out.append(str(cpus[i]) if j == i else f"{cpus[i]}-{cpus[j]}")
        i = j + 1
    return ",".join(out)


@pytest.fixture(scope="module")
def exe():
    if not os.path.exists(EXE):
        r = subprocess.run(["make", "-C", os.path.join(ROOT, "visualslam_amd", "cxx")], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
    return EXE


def fake_tree(tmp_path, nodes, devices):
    for node, cpus in nodes.items():
        d = tmp_path / "devices" / "system" / "node" / f"node{node}"
        d.mkdir(parents=True)
        (d / "cpulist").write_text(cpus + "\n")
    for bdf, node in devices.items():
        d = tmp_path / "bus" / "pci" / "devices" / bdf
        d.mkdir(parents=True)
        (d / "numa_node").write_text(f"{node}\n")
    return str(tmp_path)


def probe(exe, root, bdf, *extra, rank=0):
    env = dict(os.environ, RANK=str(rank), WORLD_SIZE="8", LOCAL_RANK=str(rank))
    r = subprocess.run([exe, "--numa-probe", "--sysfs", root, "--bdf", bdf, *extra], capture_output=True, text=True, timeout=60, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_pci_address_resolves_to_its_node_and_cpus(exe, tmp_path):
    mine = sorted(os.sched_getaffinity(0))
    assert len(mine) >= 2
    half = len(mine) // 2
    node0, node1 = mine[:half], mine[half:]
    # node 1's list also names CPUs this process may not use (a container's cpuset): they must be dropped
    root = fake_tree(tmp_path, {0: cpulist(node0), 1: cpulist(node1) + ",4000-4003"}, {"0000:c1:00.0": 1, "0000:05:00.0": 0, "0000:ff:00.0": -1})
    d = probe(exe, root, "0000:C1:00.0", rank=3)  # HIP prints upper-case hex digits, sysfs directories are lower case
    pl = d["placement"]
    assert pl["rank"] == 3 and pl["gpu"] == 3 and pl["pci_bus_id"] == "0000:c1:00.0" and pl["numa_node"] == 1
    assert pl["cpus"] == cpulist(node1) and pl["n_cpus"] == len(node1) and pl["bound"] is False
    assert d["affinity_after"] == cpulist(mine)  # report only: nothing moved
    d = probe(exe, root, "0000:05:00.0")
    assert d["placement"]["numa_node"] == 0 and d["placement"]["cpus"] == cpulist(node0)


def test_binding_narrows_the_threads_affinity(exe, tmp_path):
    mine = sorted(os.sched_getaffinity(0))
    node1 = mine[len(mine) // 2:]
    root = fake_tree(tmp_path, {0: cpulist(mine[: len(mine) // 2]), 1: cpulist(node1)}, {"0000:c1:00.0": 1})
    d = probe(exe, root, "0000:c1:00.0", "--numa-bind")
    assert d["placement"]["bound"] is True and d["affinity_after"] == cpulist(node1)


def test_unknown_node_or_foreign_cpus_leave_the_rank_alone(exe, tmp_path):
    mine = sorted(os.sched_getaffinity(0))
    root = fake_tree(tmp_path, {0: "4000-4031"}, {"0000:ff:00.0": -1, "0000:05:00.0": 0})
    for bdf in ("0000:ff:00.0", "0000:05:00.0", "0000:77:00.0"):  # node -1; a node whose CPUs are all outside the mask; no such device
        d = probe(exe, root, bdf, "--numa-bind")
        assert d["placement"]["bound"] is False and d["affinity_after"] == cpulist(mine), d
        assert d["placement"]["cpus"] == cpulist(mine) and d["placement"]["note"]
