from .polynomial import B2A, Polynomial
