"""Compiled-map cache: the map half of the reference's ``scl scenario build`` (``cli/studio.py:54-84``)
for the accelerated path.

``build_scenario(dir)`` parses the scenario's ``map.net.xml`` (or the shipped ``map.smxnet.json.gz``),
compiles it (lanepoints, grids, packed device records — ``map_compiler``) and stores everything in
``<dir>/map.smxmap.npz`` keyed by a digest of the source map; ``load_compiled_map(dir)`` returns the
tables from that file when the digest still matches and compiles (and refreshes the file when the
directory is writable) otherwise.  minicity: 1.2 s of parse + compile + pack become one ``np.load``.

    python -m smarts_amd.scenario_build scenarios/loop [more dirs ...]
"""
from __future__ import annotations

import dataclasses
import hashlib
import os
import sys
from typing import Dict, Optional, Tuple

import numpy as np

from .map_compiler import CompiledMap, compile_map, pack_tables
from .sumo_map import load_net

CACHE_NAME = "map.smxmap.npz"
FORMAT = 5  # bump when CompiledMap / the packed records change
_SOURCES = ("map.net.xml", "map.smxnet.json.gz")


def _source_digest(scenario_dir: str) -> str:
    for name in _SOURCES:
        path = os.path.join(scenario_dir, name)
        if os.path.exists(path):
            h = hashlib.sha256()
            h.update(name.encode())
            with open(path, "rb") as f:
                for chunk in iter(lambda: f.read(1 << 20), b""):
                    h.update(chunk)
            return h.hexdigest()
    raise FileNotFoundError(f"{scenario_dir}: none of {_SOURCES}")


def _to_arrays(cm: CompiledMap, packed: Dict[str, np.ndarray], digest: str) -> Dict[str, np.ndarray]:
    out: Dict[str, np.ndarray] = {"__format__": np.array(FORMAT), "__digest__": np.array(digest)}
    for f in dataclasses.fields(cm):
        v = getattr(cm, f.name)
        if f.name == "extras":
            continue
        out["cm." + f.name] = np.asarray(v)
    for k, v in packed.items():
        out["packed." + k] = v
    return out


def _from_arrays(z) -> Tuple[CompiledMap, Dict[str, np.ndarray]]:
    kw = {}
    for f in dataclasses.fields(CompiledMap):
        if f.name == "extras":
            continue
        v = z["cm." + f.name]
        if f.name in ("lane_ids", "road_ids"):
            v = [str(x) for x in v]
        elif f.name in ("lpg_cell", "sg_cell", "default_lane_width", "lanepoint_spacing"):
            v = float(v)
        elif f.name == "max_fanout":
            v = int(v)
        elif f.name == "shifted_by":
            v = tuple(float(x) for x in v)
        kw[f.name] = v
    packed = {k[len("packed."):]: z[k] for k in z.files if k.startswith("packed.")}
    return CompiledMap(**kw), packed


def build_scenario(scenario_dir: str, out_path: Optional[str] = None) -> str:
    """Compile the scenario's map and write the cache; returns the cache path."""
    digest = _source_digest(scenario_dir)
    cm = compile_map(load_net(scenario_dir))
    packed = pack_tables(cm)
    out_path = out_path or os.path.join(scenario_dir, CACHE_NAME)
    tmp = out_path + ".tmp.npz"
    np.savez(tmp, **_to_arrays(cm, packed, digest))
    os.replace(tmp, out_path)
    return out_path


def load_compiled_map(scenario_dir: str, use_cache: bool = True) -> CompiledMap:
    """The scenario's ``CompiledMap`` with its packed tables attached (``cm.extras['packed']``, which
    ``engine.map_tables_struct`` uses instead of packing again)."""
    path = os.path.join(scenario_dir, CACHE_NAME)
    digest = _source_digest(scenario_dir)
    if use_cache and os.path.exists(path):
        try:
            with np.load(path, allow_pickle=False) as z:
                if int(z["__format__"]) == FORMAT and str(z["__digest__"]) == digest:
                    cm, packed = _from_arrays(z)
                    cm.extras["packed"] = packed
                    cm.extras["from_cache"] = True
                    return cm
        except (OSError, KeyError, ValueError):
            pass  # unreadable or stale: rebuild below
    cm = compile_map(load_net(scenario_dir))
    cm.extras["packed"] = pack_tables(cm)
    cm.extras["from_cache"] = False
    if use_cache and os.access(scenario_dir, os.W_OK):
        try:
            tmp = path + ".tmp.npz"
            np.savez(tmp, **_to_arrays(cm, cm.extras["packed"], digest))
            os.replace(tmp, path)
        except OSError:
            pass
    return cm


if __name__ == "__main__":
    for d in sys.argv[1:]:
        from .env.core import resolve_scenario

        print(build_scenario(resolve_scenario(d)))
