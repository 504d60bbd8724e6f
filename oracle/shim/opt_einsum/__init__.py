"""Minimal stand-in for the ``opt_einsum`` package (absent from this image).

TEST INFRASTRUCTURE ONLY.  Used solely by ``oracle/make_golden.py`` to import
the upstream reference from /root/reference in the build container (the
reference does ``import opt_einsum as oe`` at module scope).
It maps the two entry points the reference uses onto ``numpy.einsum``.
Never imported by the product package.
"""
import numpy as np

from . import contract as _contract_module  # noqa: F401  (sub-module must exist)
from .contract import ContractExpression

__all__ = ['contract', 'contract_expression', 'ContractExpression']


def _dense(op):
    return op.todense() if hasattr(op, 'todense') else np.asarray(op)


def _path(optimize):
    if optimize is True or optimize is None or optimize == 'auto':
        return 'optimal'
    if optimize is False:
        return False
    if isinstance(optimize, (list, tuple)):
        return ['einsum_path', *[tuple(p) for p in optimize]]
    return optimize


def contract(subscripts, *operands, optimize=True, backend=None, out=None, **_):
    ops = [_dense(o) for o in operands]
    res = np.einsum(subscripts, *ops, optimize=_path(optimize), out=out)
    if backend == 'sparse':
        import sparse
        return sparse.COO.from_numpy(res)
    return res


def contract_expression(subscripts, *shapes, optimize=True, **_):
    return ContractExpression(subscripts, shapes, _path(optimize))
