"""Checkpoint reading for the hub path -- mirror of Checkpoints.load_network, mdir/learning/checkpoints.py:209-220.
File format: {"type","frozen","network_params":{"model","runtime"},"model_state"} (mdir/learning/network.py:212-220)."""
import torch

from .network import load_restricted


class Checkpoints:
    @classmethod
    def load_network(cls, directory):
        if directory is None:
            return None
        # the payload is dicts / lists / strings / numbers / tensors: the restricted loader reads it (no arbitrary unpickling of a
        # file that may have come over plain http, hub/model.py BASE_URL)
        checkpoint = load_restricted(directory)
        assert "net" not in checkpoint.get("_networks_included", {})
        return {"net": checkpoint, **checkpoint.pop("_networks_included", {})}
