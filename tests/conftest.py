import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

SCENARIOS = os.path.join(ROOT, "smarts_amd", "scenarios")
GOLDEN = os.path.join(ROOT, "tests", "golden")
MAPS = {"loop": "loop", "4lane": "intersections/4lane", "minicity": "minicity"}


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver on the GPU box)")


@pytest.fixture(scope="session")
def nets():
    from smarts_amd.sumo_map import load_net

    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = load_net(os.path.join(SCENARIOS, MAPS[name]))
        return cache[name]

    return get


@pytest.fixture(scope="session")
def compiled_maps(nets):
    from smarts_amd.map_compiler import compile_map

    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = compile_map(nets(name))
        return cache[name]

    return get


@pytest.fixture(scope="session")
def oracle_maps(nets):
    from oracle.road_network import ORoadNetwork

    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = ORoadNetwork(nets(name))
        return cache[name]

    return get
