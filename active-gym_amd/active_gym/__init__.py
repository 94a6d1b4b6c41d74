"""active_gym — MI355X-native drop-in for the Atari active-vision path of
elicassion/active-gym.  The export list mirrors the reference's
(active_gym/__init__.py:3-18,58-64) for the Atari family and the DMC pixel family;
the robosuite / RLBench families are out of scope and are not exported.  The observation pipeline
runs as hand-written HIP kernels in libagx.so; there is no CPU fallback."""
from . import _native  # noqa: F401  (ctypes binding; loading the .so is deferred to first use)
from .pipeline import ObsPipeline  # noqa: F401
from .atari_env import (  # noqa: F401
    AtariBaseEnv,
    AtariFixedFovealEnv,
    AtariFlexibleFovealEnv,
    AtariFixedFovealPeripheralEnv,
    AtariEnvArgs,
    AtariEnv,
)
from .dmc_env import (  # noqa: F401
    DMCBaseEnv,
    DMCFixedFovealEnv,
    DMCFlexibleFovealEnv,
    DMCFixedFovealPeripheralEnv,
    DMCEnvArgs,
    DMCEnv,
    DMCVecEnv,
)
from .fov_env import (  # noqa: F401
    RecordWrapper,
    FixedFovealEnv,
    FlexibleFovealEnv,
    FlexibleFovealEnvActionType,
    FixedFovealPeripheralEnv,
)
from .vector import AtariVecEnv  # noqa: F401
from .sharding import shard_bounds, shard_game, ShardedAtariVecEnv, make_vec_env  # noqa: F401

__version__ = "0.1.0"
