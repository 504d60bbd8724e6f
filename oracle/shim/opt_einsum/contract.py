"""``opt_einsum.contract`` sub-module stand-in (see package docstring)."""
import numpy as np


class ContractExpression:
    """Callable with a pre-computed contraction path, like opt_einsum's."""

    def __init__(self, subscripts, shapes, optimize):
        self.subscripts = subscripts
        if optimize in (False, None):
            self.path = False
        elif isinstance(optimize, list):
            self.path = optimize
        else:
            dummies = [np.empty(s) for s in shapes]
            self.path = np.einsum_path(subscripts, *dummies, optimize=optimize)[0]

    def __call__(self, *operands, out=None, **_):
        return np.einsum(self.subscripts, *operands, optimize=self.path, out=out)
