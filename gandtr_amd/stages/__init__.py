"""Stage (operator) layer: ``stage(params: dict, data: tuple) -> (metadata: dict, *output_columns)``
(contract: mdir/examples/perform_scenario.py:129).  Only ``infer`` is on the hot path's boundary."""
from . import infer as _infer_module
from .infer import infer

FUNCTIONS = {"mdir.stages.infer.infer": infer}
