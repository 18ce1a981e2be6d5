"""Host-side mirror of the reference's environment surface for the accelerated path
(``smarts.env:hiway-v0``, ``AgentSpec`` / ``AgentInterface``, ``ParallelEnv``, ``FormatObs``)."""
from .agent import Agent, AgentSpec  # noqa: F401
from .agent_interface import (  # noqa: F401
    OGM, RGB, Accelerometer, ActionSpaceType, AgentInterface, AgentType, DoneCriteria, DrivableAreaGridMap,
    EventConfiguration, Lidar, NeighborhoodVehicles, RoadWaypoints, Waypoints,
)
from .core import SMARTSDestroyedError, SMARTSNotSetupError  # noqa: F401
from .format_obs import FormatObs, StdObs  # noqa: F401
from .hiway_env import HiWayEnv  # noqa: F401
from .observations import (  # noqa: F401
    Collision, Dimensions, EgoVehicleObservation, Events, GridMapMetadata, Heading, Observation, OccupancyGridMap,
    VehicleObservation, ViaPoint, Vias, Waypoint,
)
from ..vias import Via  # noqa: F401
from .parallel_env import ParallelEnv  # noqa: F401


def make(env_id: str, **kwargs) -> HiWayEnv:
    """``gym.make`` for the one id the reference registers, ``"smarts.env:hiway-v0"``
    (smarts/env/__init__.py:22-25); the module prefix is optional, as with gym."""
    name = env_id.split(":")[-1]
    if name != "hiway-v0":
        raise ValueError(f"unknown environment id {env_id!r}: the accelerated path provides 'hiway-v0'")
    return HiWayEnv(**kwargs)


try:  # with gym installed, gym.make("smarts_amd.env:hiway-v0", ...) works like the reference's id
    from gym.envs.registration import register as _register

    _register(id="hiway-v0", entry_point="smarts_amd.env:HiWayEnv")
except Exception:  # gym is not part of this image; ``make`` above does not need it
    pass
