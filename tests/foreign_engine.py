"""Test-only engine plugin: the KinematicEngine hidden behind the plain BaseEntity getters/setters (no hot_state()),
i.e. what a third-party add-gym engine looks like to the agent.  Exercises the agent's slow (pack/unpack) path."""
from add_gym_amd.engine.kinematic_engine import KinematicEngine, KinematicEntity, KinematicScene


class OpaqueEntity:
    def __init__(self, inner: KinematicEntity):
        object.__setattr__(self, "_inner", inner)

    def __getattr__(self, name):
        if name == "hot_state":
            raise AttributeError(name)
        return getattr(object.__getattribute__(self, "_inner"), name)


class OpaqueScene(KinematicScene):
    def add_entity(self, morph_type, **kw):
        e = super().add_entity(morph_type, **kw)
        return e if morph_type == "plane" else OpaqueEntity(e)


class ForeignEngine(KinematicEngine):
    def create_scene(self, show_viewer, sim_options, rigid_options, vis_options=None, viewer_options=None):
        return OpaqueScene(sim_options["dt"], self._device, self._lag)
