"""Minimal stand-in for the `gym` package (absent from this image, no network).

TEST INFRASTRUCTURE ONLY.  Lets `/root/reference/gym_go` import in THIS container so the
oracle can be pinned against the real reference and golden vectors generated.  It provides
exactly what gym_go/__init__.py:1-10 and gym_go/envs/go_env.py:3,19,35-37 touch.
"""
import importlib

from . import spaces  # noqa: F401
from .envs import registration  # noqa: F401


class Env:
    metadata = {}


def make(spec, **kwargs):
    mod, _, env_id = spec.partition(':')
    if not env_id:
        mod, env_id = None, spec
    if mod:
        importlib.import_module(mod)
    entry = registration.registry[env_id]
    mod_name, _, cls_name = entry.partition(':')
    cls = getattr(importlib.import_module(mod_name), cls_name)
    return cls(**kwargs)
