"""Agent interface mirror (reference ``smarts/core/agent_interface.py``, ``smarts/core/controllers/__init__.py``).

Same names, fields, defaults and presets as the reference so that agent specs written for
``hiway-v0`` construct unchanged.  What the accelerated path does not implement fails loudly when an
environment is built from the interface (``validate_for_device``), never silently.
"""
from __future__ import annotations

from dataclasses import dataclass, field, replace
from enum import Enum, IntEnum
from typing import List, Optional, Union

from ..lidar import BasicLidar, SensorParams as LidarSensorParams


class ActionSpaceType(Enum):
    """controllers/__init__.py:42-57."""

    Continuous = 0
    Lane = 1
    ActuatorDynamic = 2
    LaneWithContinuousSpeed = 3
    TargetPose = 4
    Trajectory = 5
    MultiTargetPose = 6  # for boid control
    MPC = 7
    TrajectoryWithTime = 8  # for pure interpolation provider
    Imitation = 9


# action spaces with a device controller (include/smx.h SMX_ACTION_SPACE_*)
DEVICE_ACTION_SPACES = (ActionSpaceType.Lane, ActionSpaceType.Continuous, ActionSpaceType.ActuatorDynamic,
                        ActionSpaceType.LaneWithContinuousSpeed, ActionSpaceType.Trajectory)


@dataclass
class DrivableAreaGridMap:
    """agent_interface.py:29-38."""

    width: int = 256
    height: int = 256
    resolution: float = 50 / 256


@dataclass
class OGM:
    """agent_interface.py:41-51."""

    width: int = 256
    height: int = 256
    resolution: float = 50 / 256


@dataclass
class RGB:
    """agent_interface.py:54-64."""

    width: int = 256
    height: int = 256
    resolution: float = 50 / 256


@dataclass
class Lidar:
    """agent_interface.py:67-71."""

    sensor_params: LidarSensorParams = BasicLidar


@dataclass
class Waypoints:
    """agent_interface.py:74-80."""

    lookahead: int = 32


@dataclass
class RoadWaypoints:
    """agent_interface.py:83-93."""

    horizon: int = 20


@dataclass
class NeighborhoodVehicles:
    """agent_interface.py:96-101."""

    radius: Optional[float] = None


@dataclass
class Accelerometer:
    """agent_interface.py:104-108."""


class AgentType(IntEnum):
    """agent_interface.py:111-142."""

    Buddha = 0
    Full = 1
    Standard = 2
    Laner = 3
    Loner = 4
    Tagger = 5
    StandardWithAbsoluteSteering = 6
    LanerWithSpeed = 7
    Tracker = 8
    Boid = 9
    MPCTracker = 10
    TrajectoryInterpolator = 11
    Imitation = 12


@dataclass(frozen=True)
class AgentsListAlive:
    """agent_interface.py:145-152."""

    agents_list: List[str]
    minimum_agents_alive_in_list: int


@dataclass(frozen=True)
class AgentsAliveDoneCriteria:
    """agent_interface.py:155-176."""

    minimum_ego_agents_alive: Optional[int] = None
    minimum_total_agents_alive: Optional[int] = None
    agent_lists_alive: Optional[List[AgentsListAlive]] = None


@dataclass(frozen=True)
class EventConfiguration:
    """agent_interface.py:179-186."""

    not_moving_time: float = 60
    not_moving_distance: float = 1


@dataclass(frozen=True)
class DoneCriteria:
    """agent_interface.py:189-211."""

    collision: bool = True
    off_road: bool = True
    off_route: bool = True
    on_shoulder: bool = False
    wrong_way: bool = False
    not_moving: bool = False
    agents_alive: Optional[AgentsAliveDoneCriteria] = None


@dataclass
class AgentInterface:
    """agent_interface.py:214-297."""

    debug: bool = False
    event_configuration: EventConfiguration = EventConfiguration()
    done_criteria: DoneCriteria = field(default_factory=lambda: DoneCriteria())
    max_episode_steps: Optional[int] = None
    neighborhood_vehicles: Union[NeighborhoodVehicles, bool] = False
    waypoints: Union[Waypoints, bool] = False
    road_waypoints: Union[RoadWaypoints, bool] = False
    drivable_area_grid_map: Union[DrivableAreaGridMap, bool] = False
    ogm: Union[OGM, bool] = False
    rgb: Union[RGB, bool] = False
    lidar: Union[Lidar, bool] = False
    action: Optional[ActionSpaceType] = None
    vehicle_type: str = "sedan"
    accelerometer: Union[Accelerometer, bool] = True

    def __post_init__(self):
        self.neighborhood_vehicles = AgentInterface._resolve_config(self.neighborhood_vehicles, NeighborhoodVehicles)
        self.waypoints = AgentInterface._resolve_config(self.waypoints, Waypoints)
        self.road_waypoints = AgentInterface._resolve_config(self.road_waypoints, RoadWaypoints)
        self.drivable_area_grid_map = AgentInterface._resolve_config(self.drivable_area_grid_map, DrivableAreaGridMap)
        self.ogm = AgentInterface._resolve_config(self.ogm, OGM)
        self.rgb = AgentInterface._resolve_config(self.rgb, RGB)
        self.lidar = AgentInterface._resolve_config(self.lidar, Lidar)
        self.accelerometer = AgentInterface._resolve_config(self.accelerometer, Accelerometer)
        assert self.vehicle_type in {"sedan", "bus"}

    @staticmethod
    def from_type(requested_type: AgentType, **kwargs) -> "AgentInterface":
        """agent_interface.py:299-396: the same presets."""
        A = ActionSpaceType
        presets = {
            AgentType.Buddha: dict(),
            AgentType.Full: dict(neighborhood_vehicles=True, waypoints=True, drivable_area_grid_map=True, ogm=True,
                                 rgb=True, lidar=True, action=A.Continuous),
            AgentType.StandardWithAbsoluteSteering: dict(waypoints=True, neighborhood_vehicles=True, action=A.Continuous),
            AgentType.Standard: dict(waypoints=True, neighborhood_vehicles=True, action=A.ActuatorDynamic),
            AgentType.Laner: dict(waypoints=True, action=A.Lane),
            AgentType.LanerWithSpeed: dict(waypoints=True, action=A.LaneWithContinuousSpeed),
            AgentType.Tracker: dict(waypoints=True, action=A.Trajectory),
            AgentType.TrajectoryInterpolator: dict(action=A.TrajectoryWithTime),
            AgentType.MPCTracker: dict(waypoints=True, action=A.MPC),
            AgentType.Boid: dict(waypoints=True, neighborhood_vehicles=True, action=A.MultiTargetPose),
            AgentType.Loner: dict(waypoints=True, action=A.Continuous),
            AgentType.Tagger: dict(waypoints=True, neighborhood_vehicles=True, action=A.Continuous),
            AgentType.Imitation: dict(neighborhood_vehicles=True, action=A.Imitation),
        }
        if requested_type not in presets:
            raise Exception("Unsupported agent type %s" % requested_type)
        return AgentInterface(**presets[requested_type]).replace(**kwargs)

    def replace(self, **kwargs) -> "AgentInterface":
        """agent_interface.py:398-406."""
        return replace(self, **kwargs)

    @property
    def action_space(self):
        """Deprecated alias of ``action`` (agent_interface.py:408-412)."""
        return self.action

    @staticmethod
    def _resolve_config(config, type_):
        if config is True:
            return type_()
        elif isinstance(config, type_):
            return config
        else:
            return False

    # ------------------------------------------------------------------ device support
    def validate_for_device(self):
        """Raise for anything the MI355X path does not implement (the Lane,
        Continuous, ActuatorDynamic, LaneWithContinuousSpeed and Trajectory action spaces and the waypoints /
        neighbourhood / accelerometer / OGM / drivable-area grid map / lidar / road-waypoints sensors)."""
        # action=None (AgentType.Buddha) needs no controller: the vehicle is never given a command
        if self.action is not None and self.action not in DEVICE_ACTION_SPACES:
            raise NotImplementedError(
                f"action space {self.action} is not on the accelerated path "
                f"(supported: {[a.name for a in DEVICE_ACTION_SPACES]})")
        if self.rgb:
            raise NotImplementedError("AgentInterface.rgb is not on the accelerated path")
        if self.road_waypoints and not 1 <= self.road_waypoints.horizon <= 64:
            raise NotImplementedError("RoadWaypoints.horizon must be 1..64 on the accelerated path (include/smx.h)")
        if self.vehicle_type != "sedan":
            raise NotImplementedError("only the sedan chassis is modelled")
        alive = self.done_criteria.agents_alive
        if alive is not None and alive.agent_lists_alive and len(alive.agent_lists_alive) > 4:
            raise NotImplementedError("DoneCriteria.agents_alive: at most four agent lists on the accelerated path")
