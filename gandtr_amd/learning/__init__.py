"""mirror of mdir/learning/__init__.py:9-13 (load_network)"""
from .checkpoints import Checkpoints
from .network import initialize_network


def load_network(params, device):
    state = Checkpoints.load_network(params["path"])
    if state is not None:
        return initialize_network(None, device, state, params["runtime"])
    return initialize_network(params, device)
