class Structure:
    pass


class Lattice:
    pass
