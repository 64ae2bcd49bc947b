"""Stand-in for the absent `torch_geometric` package (test infrastructure only)."""
