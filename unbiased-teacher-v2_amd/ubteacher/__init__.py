"""MI355X-native Unbiased Teacher v2 training step (host side).

Mirrors the reference's public surface (ubteacher/__init__.py:2 -> add_ubteacher_config;
ubteacher.engine trainers; ubteacher.modeling registries) on top of a C-ABI library of
hand-written HIP kernels for gfx950 (see include/utv2.h, csrc/).
"""
from .config import add_ubteacher_config  # noqa: F401,E402
