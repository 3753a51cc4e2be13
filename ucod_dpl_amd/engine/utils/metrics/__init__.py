from .metric import statistics, compute_cod_metric  # noqa: F401
