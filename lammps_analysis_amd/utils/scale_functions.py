"""Memory scale functions of the calculators
(mdsuite/utils/scale_functions.py:30-117 of the reference): how many bytes one
loaded configuration costs a calculator, as a function of its stored size."""
import numpy as np


def linear_scale_function(memory_usage, scale_factor=1):
    return memory_usage * scale_factor


def linearithmic_scale_function(memory_usage, scale_factor=1):
    return scale_factor * memory_usage * np.log(memory_usage)


def quadratic_scale_function(memory_usage, inner_scale_factor=1, outer_scale_factor=1):
    return outer_scale_factor * (memory_usage * inner_scale_factor) ** 2


def polynomial_scale_function(memory_usage, inner_scale_factor=1, outer_scale_factor=1,
                              order=3):
    return outer_scale_factor * (memory_usage * inner_scale_factor) ** order
