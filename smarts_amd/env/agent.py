"""``Agent`` / ``AgentSpec`` mirror (reference ``smarts/core/agent.py:33-67``, ``smarts/zoo/agent_spec.py:35-118``)."""
from __future__ import annotations

import inspect
from dataclasses import dataclass
from typing import Any, Callable, Optional

from .agent_interface import AgentInterface


class Agent:
    """The base class for agents (agent.py:33-67)."""

    @classmethod
    def from_function(cls, agent_function: Callable[[Any], Any]) -> "Agent":
        assert callable(agent_function)

        class FunctionAgent(Agent):
            def act(self, obs):
                return agent_function(obs)

        return FunctionAgent()

    def act(self, obs, **configs):
        raise NotImplementedError


@dataclass
class AgentSpec:
    """agent_spec.py:35-118 (cloudpickle self-check omitted: nothing here crosses a process)."""

    interface: Optional[AgentInterface] = None
    agent_builder: Optional[Callable[..., Agent]] = None
    agent_params: Optional[Any] = None
    observation_adapter: Callable = lambda obs: obs
    action_adapter: Callable = lambda act: act
    reward_adapter: Callable = lambda obs, reward: reward
    info_adapter: Callable = lambda obs, reward, info: info

    def build_agent(self) -> Agent:
        if self.agent_builder is None:
            raise ValueError("Can't build agent, no agent builder was supplied")
        if not callable(self.agent_builder):
            raise ValueError(f"agent_builder: {self.agent_builder} is not callable")
        if self.agent_params is None:
            return self.agent_builder()
        elif isinstance(self.agent_params, (list, tuple)):
            return self.agent_builder(*self.agent_params)
        elif isinstance(self.agent_params, dict):
            fas = inspect.getfullargspec(self.agent_builder)
            if fas[2] is not None:
                return self.agent_builder(**self.agent_params)
            return self.agent_builder(**{k: self.agent_params[k] for k in self.agent_params.keys() & set(fas[0])})
        return self.agent_builder(self.agent_params)
