"""goal_distance / scale_gym_data / unscale_gym_data with the reference's exact numerics
(reference pybullet_robot_envs/envs/utils.py:11-14, 78-107): limits are float32 arrays, so
`high - low` is rounded in float32 and then promoted when combined with float64 data.  Batched
inputs ([N, dim]) broadcast against the space's [dim] limits."""
import numpy as np


def goal_distance(a, b):
    a, b = np.asarray(a), np.asarray(b)
    if not a.shape == b.shape:
        raise AssertionError("goal_distance(): shape of points mismatch")
    return np.linalg.norm(a - b, axis=-1)


def scale_gym_data(data_space, data):
    """Rescale from [low, high] to [-1, 1]."""
    data = np.asarray(data)
    assert data.shape[-len(data_space.shape):] == data_space.shape
    low, high = data_space.low, data_space.high
    return 2.0 * ((data - low) / (high - low)) - 1.0


def unscale_gym_data(data_space, scaled_data):
    """Rescale from [-1, 1] to [low, high]."""
    scaled_data = np.asarray(scaled_data)
    assert scaled_data.shape[-len(data_space.shape):] == data_space.shape
    low, high = data_space.low, data_space.high
    return low + (0.5 * (scaled_data + 1.0) * (high - low))
