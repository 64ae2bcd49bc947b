"""Model hyper-parameter defaults (the model part of the reference's RunConfig, config.py:10-17).
Training options are out of scope for this inference engine."""
from dataclasses import dataclass


@dataclass
class ModelConfig:
    cutoff: float = 5.0
    threebody_cutoff: float = 4.0
    l_max: int = 3
    n_max: int = 3
    num_types: int = 95
    embedding_dim: int = 64
    num_blocks: int = 3
    energy_scale: float = 1.0
    length_scale: float = 1.0

    def build(self, elemental_energies=None, device=None):
        from .model.build import build_model

        return build_model(self.cutoff, self.threebody_cutoff, self.l_max, self.n_max, self.num_types, self.embedding_dim,
                           self.num_blocks, elemental_energies=elemental_energies, energy_scale=self.energy_scale,
                           length_scale=self.length_scale, device=device)
