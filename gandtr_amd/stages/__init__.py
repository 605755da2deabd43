"""Stage (operator) layer: ``stage(params: dict, data: tuple) -> (metadata: dict, *output_columns)``
(contract: mdir/examples/perform_scenario.py:129).  ``infer`` and ``validate`` (descriptor extraction + ranking, the retrieval
evaluation's caller of the path) are on the hot path's boundary; ``whiten`` / ``learn_lw_whitening`` belong to the "next" rows of
SURVEY.md section 8f."""
from . import infer as _infer_module
from .infer import infer
from .validate import rank_images, validate
from .whiten import learn_lw_whitening, whiten

FUNCTIONS = {"mdir.stages.infer.infer": infer, "mdir.stages.validate.validate": validate, "mdir.stages.whiten.whiten": whiten,
             "mdir.stages.whiten.learn_lw_whitening": learn_lw_whitening,
             # this build's own stage (no reference counterpart): the validate arithmetic with ranks / scores as output columns
             "gandtr_amd.stages.validate.rank_images": rank_images}
