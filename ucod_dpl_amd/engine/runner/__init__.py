from .runner import StandardRunner, LocalRefineRunner, RunnerFactory, create_runner, get_available_runner_types  # noqa: F401
from .loop_UCOD_DPL import TrainLoop  # noqa: F401
from .pipeline import FeaturePipeline  # noqa: F401
from .loop_CORAL import LocalRefineValidationLoop, WindowFeatures  # noqa: F401
