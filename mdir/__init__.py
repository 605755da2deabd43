"""Drop-in alias: ``import mdir...`` resolves to the MI355X host mirror in ``gandtr_amd`` for the modules on the hot path.

    mdir.hub.model                         -> gandtr_amd.hub.model
    mdir.components.model.network          -> gandtr_amd.components.model.network   (MODEL_LABELS, initialize_model)
    mdir.components.data.wrapper           -> gandtr_amd.components.data.wrapper    (WRAPPERS_LABELS, initialize_wrappers)
    mdir.components.data.transform         -> gandtr_amd.components.data.transform  (initialize_transforms)
    mdir.learning / mdir.learning.network  -> gandtr_amd.learning(.network)         (NETWORKS, initialize_network, load_network)
    mdir.stages.infer                      -> gandtr_amd.stages.infer               (infer(params, data))
    mdir.tools.tensors                     -> gandtr_amd.tools.tensors

Unlike the reference's ``mdir/__init__.py`` this does NOT call ``torch.set_num_threads(3)`` (mdir/stages/infer.py:12-15)
and does not touch ``sys.path``; set ``GANDTR_REFERENCE_THREADS=1`` to reproduce the thread cap.
"""
import importlib
import os
import sys

_ALIASES = {
    "mdir.hub": "gandtr_amd.hub",
    "mdir.hub.model": "gandtr_amd.hub.model",
    "mdir.components": "gandtr_amd.components",
    "mdir.components.model": "gandtr_amd.components.model",
    "mdir.components.model.network": "gandtr_amd.components.model.network",
    "mdir.components.model.network.p2p_networks": "gandtr_amd.components.model.network.p2p_networks",
    "mdir.components.model.network.cirnet": "gandtr_amd.components.model.network.cirnet",
    "mdir.components.model.network.hed": "gandtr_amd.components.model.network.hed",
    "mdir.components.model.weight_initialization": "gandtr_amd.components.model.weight_initialization",
    "mdir.components.data": "gandtr_amd.components.data",
    "mdir.components.data.wrapper": "gandtr_amd.components.data.wrapper",
    "mdir.components.data.transform": "gandtr_amd.components.data.transform",
    "mdir.learning": "gandtr_amd.learning",
    "mdir.learning.network": "gandtr_amd.learning.network",
    "mdir.learning.checkpoints": "gandtr_amd.learning.checkpoints",
    "mdir.stages": "gandtr_amd.stages",
    "mdir.stages.infer": "gandtr_amd.stages.infer",
    "mdir.stages.whiten": "gandtr_amd.stages.whiten",
    "mdir.stages.validate": "gandtr_amd.stages.validate",
    "mdir.tools": "gandtr_amd.tools",
    "mdir.tools.tensors": "gandtr_amd.tools.tensors",
    "mdir.tools.utils": "gandtr_amd.tools.utils",
}

for _alias, _target in _ALIASES.items():
    _mod = importlib.import_module(_target)
    sys.modules[_alias] = _mod
    _parent, _, _leaf = _alias.rpartition(".")
    setattr(sys.modules[_parent], _leaf, _mod)

if os.environ.get("GANDTR_REFERENCE_THREADS") == "1":
    import torch
    torch.set_num_threads(3)
