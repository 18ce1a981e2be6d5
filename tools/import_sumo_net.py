#!/usr/bin/env python3
"""Convert a SUMO ``map.net.xml`` into the compact network description that
``smarts_amd.sumo_map.load_net`` also reads (``map.smxnet.json.gz``).

This is the build's twin of the "scenario build" step for maps:
only what the hot path consumes is kept — lanes (id, index, speed, length, width,
centre-line shape), edges (function, end nodes), node positions and connections.
Coordinates are stored un-shifted; the origin shift is applied at load time.

    python tools/import_sumo_net.py <scenario_dir_or_net.xml> <out_dir>
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smarts_amd.sumo_map import SMX_NET_NAME, load_net  # noqa: E402


def main():
    src, out_dir = sys.argv[1], sys.argv[2]
    net = load_net(src, shift_to_origin=False)
    os.makedirs(out_dir, exist_ok=True)
    out = os.path.join(out_dir, SMX_NET_NAME)
    net.save(out)
    print(f"{src} -> {out}: {len(net.edges)} edges, {len(net.all_lanes())} lanes, {os.path.getsize(out)} bytes")


if __name__ == "__main__":
    main()
