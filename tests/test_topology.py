"""svh_get_device_topology / svh_bind_host_to_device (include/svh.h): the host side of "one process per GPU" -- a rank
stays on the cores of its GPU's NUMA node.  The sysfs parsing and the binding run here on the CPU against directories
this test builds (svh_topology_from_sysfs, svh_bind_host_to_topology); on a GPU box tests/test_shard_driver.py sees the
real thing through svh_shard's records.  SURVEY 8(e)."""
import ctypes as C
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stereo-vision_amd"))


class Topo(C.Structure):
    _fields_ = [("device", C.c_int32), ("numa_node", C.c_int32), ("n_cpus", C.c_int32), ("r", C.c_int32),
                ("pci", C.c_char * 32), ("cpulist", C.c_char * 256), ("mask", C.c_uint64 * 16)]

    def cpus(self):
        return {c for c in range(1024) if self.mask[c >> 6] >> (c & 63) & 1}


def fake_sysfs(tmp, bus="0000:c5:00.0", node=1, node_cpus="0-3,64-67", local=None, online="0-127"):
    d = tmp / "bus" / "pci" / "devices" / bus
    d.mkdir(parents=True)
    if node is not None:
        (d / "numa_node").write_text("%d\n" % node)
    if local is not None:
        (d / "local_cpulist").write_text(local + "\n")
    if node is not None and node >= 0 and node_cpus is not None:
        n = tmp / "devices" / "system" / "node" / ("node%d" % node)
        n.mkdir(parents=True)
        (n / "cpulist").write_text(node_cpus + "\n")
    c = tmp / "devices" / "system" / "cpu"
    c.mkdir(parents=True)
    (c / "online").write_text(online + "\n")
    return str(tmp).encode()


@pytest.fixture(scope="module")
def lib():
    import svhip as S
    return S.lib()


def test_numa_node_and_its_cpus(lib, tmp_path):
    t = Topo()
    assert lib.svh_topology_from_sysfs(fake_sysfs(tmp_path), b"0000:C5:00.0", C.byref(t)) == 0    # (HIP prints upper case)
    assert t.pci == b"0000:c5:00.0" and t.numa_node == 1 and t.n_cpus == 8
    assert t.cpus() == {0, 1, 2, 3, 64, 65, 66, 67} and t.cpulist == b"0-3,64-67"


def test_machines_without_numa_information_fall_back(lib, tmp_path):
    t = Topo()
    root = fake_sysfs(tmp_path / "a", node=-1, local="8-11")
    assert lib.svh_topology_from_sysfs(root, b"0000:c5:00.0", C.byref(t)) == 0
    assert t.numa_node == -1 and t.cpus() == {8, 9, 10, 11}
    root = fake_sysfs(tmp_path / "b", node=None, online="0-5")
    assert lib.svh_topology_from_sysfs(root, b"0000:c5:00.0", C.byref(t)) == 0
    assert t.numa_node == -1 and t.cpus() == set(range(6))
    assert lib.svh_topology_from_sysfs(str(tmp_path / "none").encode(), b"0000:c5:00.0", C.byref(t)) == 0
    assert t.n_cpus == 0                                                   # nothing known: nothing will be bound


@pytest.mark.parametrize("bad", ["3-1", "a-b", "0-99999", "4-"])
def test_malformed_lists_are_refused(lib, tmp_path, bad):
    t = Topo()
    assert lib.svh_topology_from_sysfs(fake_sysfs(tmp_path, node_cpus=bad), b"0000:c5:00.0", C.byref(t)) == -1


def test_binding_respects_the_quota_and_the_budget(tmp_path):
    """in a fresh process: the mask becomes node CPUs & allowed CPUs, at most max_cpus of them; threads created afterwards
    inherit it; CPUs outside the quota leave the mask alone"""
    allowed = sorted(os.sched_getaffinity(0))
    if len(allowed) < 3:
        pytest.skip("needs three CPUs")
    a, b, c = allowed[0], allowed[1], allowed[-1]
    root = fake_sysfs(tmp_path / "x", node_cpus="%d,%d,%d,900" % (a, b, c)).decode()
    far = fake_sysfs(tmp_path / "y", node_cpus="901-903").decode()
    code = r'''
import ctypes as C, json, os, sys, threading
sys.path.insert(0, %r); sys.path.insert(0, %r)
from test_topology import Topo
import svhip as S
L = S.lib(); t = Topo(); out = {}
L.svh_topology_from_sysfs(%r.encode(), b"0000:c5:00.0", C.byref(t))
out["far"] = L.svh_bind_host_to_topology(C.byref(t), 0), sorted(os.sched_getaffinity(0)) == %r
L.svh_topology_from_sysfs(%r.encode(), b"0000:c5:00.0", C.byref(t))
out["all"] = L.svh_bind_host_to_topology(C.byref(t), 0), sorted(os.sched_getaffinity(0))
seen = []
th = threading.Thread(target=lambda: seen.append(sorted(os.sched_getaffinity(0)))); th.start(); th.join()
out["thread"] = seen[0]
out["two"] = L.svh_bind_host_to_topology(C.byref(t), 2), sorted(os.sched_getaffinity(0))
print(json.dumps(out))
''' % (os.path.join(ROOT, "stereo-vision_amd"), os.path.join(ROOT, "tests"), far, allowed, root)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["far"] == [0, True]
    assert out["all"] == [3, sorted({a, b, c})] and out["thread"] == sorted({a, b, c})
    assert out["two"][0] == 2 and out["two"][1] == sorted({a, b, c})[:2]
