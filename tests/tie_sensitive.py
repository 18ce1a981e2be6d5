"""Golden poses on which the order among EXACTLY equidistant lanepoints decides the result.

The reference leaves that order to scipy's KD-tree heap (lanepoints.py:581-590: k = 10 nearest, then
``sorted`` by dist^2 + |heading difference|, a stable sort over the tree's output order).  The device and the
oracle's ``tie_rule = "index"`` order equal distances by global lanepoint index (deviation 1).
Found by running the oracle with both rules over every golden pose (tests/test_oracle_golden.py asserts the
lists are exact): on every other pose the two rules give the same bits."""

# waypoints_<map>_<route kind>_<lookahead>.npz: pose indices
WAYPOINTS = {
    ("loop", "empty_route", 16): [49, 224, 294],
    ("loop", "empty_route", 32): [49, 224, 294],
    ("loop", "none", 32): [49, 98, 170, 224, 294],
    ("4lane", "empty_route", 16): [84],
    ("4lane", "empty_route", 32): [84],
    ("4lane", "none", 32): [84],
    ("minicity", "empty_route", 16): [],
    ("minicity", "empty_route", 32): [],
    ("minicity", "none", 32): [],
}

# missions_<map>.npz, waypoint paths along a route (wp<lookahead>_*): pose indices
ROUTE_WAYPOINTS = {
    ("loop", 16): [], ("loop", 32): [],
    ("4lane", 16): [], ("4lane", 32): [],
    ("minicity", 16): [], ("minicity", 32): [],
}

# road_waypoints_<map>.npz: pose indices
ROAD_WAYPOINTS = {"loop": [], "4lane": [], "minicity": []}

# controller_<map>.npz: row indices (the controller asks waypoint_paths with lookahead 16 at the vehicle pose)
CONTROLLER = {"loop": [], "4lane": [179], "minicity": []}
