from .. import _bootstrap  # noqa: F401
