"""Closed-form fidelity filter functions of the standard pure-dephasing dynamical-decoupling
sequences, as functions of ``z = omega*tau`` with *tau* the total duration (reference
filter_functions/analytic.py:59-88; Cywinski et al., PRB 77, 174509 (2008)).

Everything here is ``F(omega)*omega**2`` for ideal pi pulses: ``FID`` free induction decay, ``SE``
spin echo, ``PDD`` periodic and ``CPMG`` Carr-Purcell-Meiboom-Gill decoupling with *n* pulses,
``CDD`` concatenated decoupling of order *g*, ``UDD`` Uhrig decoupling with *n* pulses.  Host-side
formulas (a handful of elementary functions per frequency); the tests use them as an independent
check of the device filter functions.
"""
import numpy as np

__all__ = ['FID', 'SE', 'PDD', 'CPMG', 'CDD', 'UDD']


def FID(z):
    z = np.asarray(z, dtype=float)
    return 2*np.sin(z/2)**2


def SE(z):
    z = np.asarray(z, dtype=float)
    return 8*np.sin(z/4)**4


def PDD(z, n):
    z = np.asarray(z, dtype=float)
    envelope = np.cos(z/2)**2 if n % 2 == 0 else np.sin(z/2)**2
    return 2*np.tan(z/(2*n + 2))**2*envelope


def CPMG(z, n):
    z = np.asarray(z, dtype=float)
    envelope = np.sin(z/2)**2 if n % 2 == 0 else np.cos(z/2)**2
    return 8*np.sin(z/(4*n))**4*envelope/np.cos(z/(2*n))**2


def CDD(z, g):
    z = np.asarray(z, dtype=float)
    out = 2.0**(2*g + 1)*np.sin(z/2**(g + 1))**2
    for k in range(1, g + 1):
        out = out*np.sin(z/2**(k + 1))**2
    return out


def UDD(z, n):
    z = np.asarray(z, dtype=float)
    k = np.arange(-n - 1, n + 1)
    terms = (-1.0)**k[:, None]*np.exp(0.5j*z[None]*np.cos(np.pi*k/(n + 1))[:, None]) \
        if z.ndim else (-1.0)**k*np.exp(0.5j*z*np.cos(np.pi*k/(n + 1)))
    return np.abs(terms.sum(axis=0))**2/2
