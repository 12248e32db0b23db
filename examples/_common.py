"""Shared by the example scripts: the Nitinol rod of the reference's examples as parameter tables.

Values as in /root/reference/examples/example_utilities.py:25-34 (material constants) and :37-73 (one row per
element, FIXED at node 0): element length 0.25 m, E = 75 GPa, r = 5 mm, rho = 6450 kg/m^3, C_d = 0.82.
"""
import os
import sys

import numpy as np
import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "continuum-robot_amd"))

ELEMENT_LENGTH, MODULUS, RADIUS, DENSITY, DRAG_COEF = 0.25, 75e9, 0.005, 6450.0, 0.82


def rod(n_elements: int, kind) -> pd.DataFrame:
    """`kind`: "linear", "nonlinear", "mixed" (alternating, linear first) or a list of per-element kinds."""
    if kind == "mixed":
        kinds = ["linear" if i % 2 == 0 else "nonlinear" for i in range(n_elements)]
    elif isinstance(kind, str):
        kinds = [kind] * n_elements
    else:
        kinds = list(kind)
    area, inertia = np.pi * RADIUS**2, np.pi * RADIUS**4 / 4
    return pd.DataFrame({
        "length": [ELEMENT_LENGTH] * n_elements,
        "elastic_modulus": [MODULUS] * n_elements,
        "moment_inertia": [inertia] * n_elements,
        "density": [DENSITY] * n_elements,
        "cross_area": [area] * n_elements,
        "type": kinds,
        "boundary_condition": ["FIXED"] + ["NONE"] * (n_elements - 1),
        "wetted_area": [2 * np.pi * RADIUS * ELEMENT_LENGTH] * n_elements,
        "drag_coef": [DRAG_COEF] * n_elements,
    })
