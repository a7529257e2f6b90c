"""`smart_logger` resolution: the real package when installed, else the in-tree shim."""
try:  # pragma: no cover - depends on the environment
    import smart_logger  # noqa: F401
    from smart_logger import Logger
    from smart_logger.parameter.ParameterTemplate import ParameterTemplate
except Exception:  # noqa: BLE001
    from . import smart_logger_shim as smart_logger
    from .smart_logger_shim import Logger, ParameterTemplate
