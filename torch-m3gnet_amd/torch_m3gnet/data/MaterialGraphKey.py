"""String keys of the tensors carried by a MaterialGraph.

Same names as the reference's key module (data/MaterialGraphKey.py:1-37) -- they are the wire
format of the `forward(graph) -> graph` protocol, so they cannot differ.  Grouped here by role.
"""
_INPUT = dict(
    POS="pos", ATOM_TYPES="atom_types", NUM_TRIPLET_I="num_triplet_i",
    EDGE_INDEX="edge_index", EDGE_CELL_SHIFT="edge_cell_shift", NUM_TRIPLET_IJ="num_triplet_ij",
    TRIPLET_EDGE_INDEX="triplet_edge_index", LATTICE="lattice", BATCH="batch",
)
_COUNTS = dict(NUM_NODES="num_nodes", NUM_EDGES="num_edges", NUM_TRIPLETS="num_triplets")
_DERIVED = dict(
    SCALED_POS="scaled_pos", SCALED_LATTICE="scaled_lattice", EDGE_DISTANCES="edge_distances",
    EDGE_WEIGHTS="edge_weights", TRIPLET_ANGLES="triplet_angles", ELEMENTAL_ENERGIES="elemental_energies",
    NODE_FEATURES="x", EDGE_ATTR="edge_attr",
)
_TARGETS = dict(
    SCALED_ATOMIC_ENERGIES="scaled_atomic_energies", SCALED_TOTAL_ENERGY="scaled_total_energy",
    TOTAL_ENERGY="total_energy", FORCES="forces", STRESSES="stresses",
)
# engine extras (not in the reference): three-body aggregate of every block, [num_blocks, E, l_max*n_max]
_EXTRA = dict(MID_EDGE_FEATURES="mid_edge_features")

ALL_KEYS = {**_INPUT, **_COUNTS, **_DERIVED, **_TARGETS, **_EXTRA}
globals().update(ALL_KEYS)
INPUT_KEYS = tuple(_INPUT.values())
