"""MI355X-native framewise speech feature engine (CtuCopy-compatible hot path)."""
from .engine import CtuError, Engine, Plan, config_dims, config_table, load_library  # noqa: F401
