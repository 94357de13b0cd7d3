"""TEST INFRASTRUCTURE - CPU restatement of the reference's per-atom coordinate
transformations in NumPy float64, operation by operation.  Only tests, the smoke check
and the benchmark's CPU baseline may import this module; the product never does.

Pinned by the reference's own unit tests
(CI/unit_tests/transformations/test_transformations.py:103-260), restated in
tests/test_oracle_reference_pins.py.
"""
import numpy as np


def unwrap_coordinates(pos, box_l, carryover=None):
    """mdsuite/transformations/unwrap_coordinates.py:51-81.  ``tf.math.round`` rounds
    half to even, as ``np.rint``."""
    pos = np.asarray(pos, dtype=np.float64)
    box_l = np.asarray(box_l, dtype=np.float64).reshape(1, 1, 3)
    if carryover is None:
        last_pos = pos[:, 0, :]
        last_image_box = np.zeros_like(last_pos)
    else:
        last_pos = np.asarray(carryover["last_pos"], dtype=np.float64)
        last_image_box = np.asarray(carryover["last_image_box"], dtype=np.float64)
    image_box = np.concatenate([last_pos[:, None, :], pos], axis=1)
    image_box = np.diff(image_box, axis=1)
    image_box = np.rint(image_box / box_l)
    image_box = -np.cumsum(image_box, axis=1)
    image_box = image_box + last_image_box[:, None, :]
    unwrapped = pos + image_box * box_l
    carry = {"last_pos": pos[:, -1, :].copy(), "last_image_box": image_box[:, -1, :].copy()}
    return unwrapped, carry


def unwrap_via_indices(pos, box_im, box_l):
    """mdsuite/transformations/unwrap_via_indices.py:48-57."""
    box_l = np.asarray(box_l, dtype=np.float64).reshape(1, 1, 3)
    return np.asarray(pos, np.float64) + np.asarray(box_im, np.float64) * box_l


def wrap_coordinates(unwrap_pos, box_l, center_box=True):
    """mdsuite/transformations/wrap_coordinates.py:60-74."""
    unwrap_pos = np.asarray(unwrap_pos, dtype=np.float64)
    box_l = np.asarray(box_l, dtype=np.float64).reshape(1, 1, 3)
    if center_box:
        unwrap_pos = unwrap_pos + box_l / 2.0
    pos = unwrap_pos - np.floor(unwrap_pos / box_l) * box_l
    if center_box:
        pos = pos - box_l / 2.0
    return pos


def scale_coordinates(pos, box_l):
    """mdsuite/transformations/scale_coordinates.py:47-54."""
    return np.asarray(pos, np.float64) * np.asarray(box_l, np.float64).reshape(1, 1, 3)


def velocity_from_positions(pos, time_step, sample_rate):
    """mdsuite/transformations/velocity_from_positions.py:42-58."""
    pos = np.asarray(pos, dtype=np.float64)
    dt = float(time_step) * float(sample_rate)
    vel = (np.roll(pos, shift=-1, axis=1) - pos) / dt
    vel = vel[:, :-1, :]
    return np.concatenate((vel, vel[:, -1:, :]), axis=1)
