"""gandtr_amd -- MI355X-native implementation of the gandtr inference hot path.

Drop-in surface (mirrors the reference's own modules):
  hubconf.py / gandtr_amd.hub.model      cyclegan, hedngan, gem_vgg16_*, gem_resnet101_*
  gandtr_amd.components.model.network    MODEL_LABELS / initialize_model
  gandtr_amd.components.data.wrapper     WRAPPERS_LABELS / initialize_wrappers
  gandtr_amd.learning.network            NETWORKS / initialize_network / SingleNetwork / CirSequentialNetwork
  gandtr_amd.stages                      infer, whiten, learn_lw_whitening (stage ABI)
"Next" rows of SURVEY.md section 8f (device paths beside the hot path):
  gandtr_amd.clahe / ingest / retrieval / whiten_learn
Compute: gandtr_amd/csrc (HIP kernels for gfx950) behind the C ABI in include/gandtr_hip.h.
"""
__version__ = "0.1.0"
