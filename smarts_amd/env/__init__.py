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
