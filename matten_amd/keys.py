"""Keys of the data dict passed between layers (the reference's `DataKey`,
eigenn/data/_key.py:14-47).  Same strings, so dicts are interchangeable."""
from typing import Dict

import torch

Type = Dict[str, torch.Tensor]

POSITIONS = "pos"
NODE_ATTRS = "node_attrs"
NODE_FEATURES = "node_features"
EDGE_INDEX = "edge_index"
EDGE_CELL_SHIFT = "edge_cell_shift"
EDGE_VECTORS = "edge_vectors"
EDGE_LENGTH = "edge_lengths"
EDGE_ATTRS = "edge_attrs"
EDGE_EMBEDDING = "edge_embedding"
EDGE_MESSAGE = "edge_message"
CELL = "cell"
ATOMIC_NUMBERS = "atomic_numbers"
SPECIES_INDEX = "species_index"
PER_ATOM_ENERGY = "atomic_energy"
TOTAL_ENERGY = "total_energy"
BATCH = "batch"

# Extension (no counterpart in eigenn/data/_key.py): optional int64 [N] class index of the node
# attributes, equal class => identical NODE_ATTRS row (for a species embedding: the species index).
# When present the self connection takes its low-rank route without having to discover the classes
# (which costs host synchronisations); when absent they are discovered from NODE_ATTRS itself.
# Supplying it is a promise that NODE_ATTRS = table[classes] for some row table: the gradient the
# self connection returns for NODE_ATTRS is then the class mean of the per-node gradient (its
# projection onto tensors of that form), which hands `table` exactly the reference's gradient at
# a fraction of the cost; per-node attribute gradients need the key left out.
NODE_ATTRS_CLASSES = "node_attrs_classes"
