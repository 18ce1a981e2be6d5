"""CPU oracle for the SMARTS hot path — TEST INFRASTRUCTURE, NOT PRODUCT.

This package is a plain Python/numpy restatement of the reference's per-tick
vehicle step + sensor/observation path (fahmyadan/SMARTS v0.6.1rc1), written
per agent and sequentially, the way the reference runs it.  Every function
cites the reference ``file:line`` it follows.

Who may import it: ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` — as the checker / the timed CPU port,
never as part of the shipped path.  ``smarts_amd`` must not import ``oracle``.

Pinning status (see DESIGN.md "Oracle"):

* in-tree arithmetic — math, coordinates, lanepoints, waypoint paths, nearest
  lanes, the lane-following and PD trajectory-tracking controllers, the
  accelerometer / driven-path / trip-meter sensors, the wrong-way test, lidar
  ray generation, ``lane_ttc`` / ``FormatObs``: pinned against outputs of the
  reference's own modules run in the build container
  (``tests/golden/gen_golden.py`` → ``tests/golden/*.npz``) and against the
  reference's known-answer tests (``tests/test_oracle_kat.py``).
* third-party arithmetic that is absent from the reference tree (pybullet
  dynamics/contacts/ray casts, Panda3D OGM raster, sumolib+rtree map queries):
  restated from the reference's call sites; map queries are pinned by the
  ``test_map.py`` known answers; **vehicle pose trajectories are parity
  unpinned** (the reference holds no golden poses and pybullet cannot run
  here) — the planar model in ``oracle/dynamics.py`` is a documented
  substitution.  Likewise **unpinned**: the OGM raster beyond the reference's
  +-2 px check, the drivable-area grid map likewise (lane bands instead of the rendered road
  mesh), lidar hits (ray / box instead of Bullet), and the scripted
  social-traffic models (constant speed, IDM car following) that stand in for SUMO
  (``oracle/sim.py::SocialBody``).
"""
