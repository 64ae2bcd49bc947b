"""Stand-in for the absent `pymatgen` package (test infrastructure only)."""
