"""alphazero_amd: MI355X-native self-play hot path (env step -> PUCT select ->
ResNet leaf evaluation -> expand -> backup) behind the reference's
Env / MCTS / Evaluator / PVNetwork interfaces.  The compute lives in
libazhip.so (hand-written HIP for gfx950, C ABI in include/az_hip.h)."""
from . import capi, synth  # noqa: F401
from .capi import AzError, Engine  # noqa: F401
