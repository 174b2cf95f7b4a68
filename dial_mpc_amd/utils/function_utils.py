"""Host-side (NumPy) copies of the reference's env helpers (dial_mpc/utils/function_utils.py:7-43).
The in-kernel versions live in csrc/rollout_body.h; these exist for API parity and for tests."""
import numpy as np


def _rotate(vec, quat):
    s, u = quat[0], np.asarray(quat[1:])
    vec = np.asarray(vec)
    r = 2 * (np.dot(u, vec) * u) + (s * s - np.dot(u, u)) * vec
    return r + 2 * s * np.cross(u, vec)


def global_to_body_velocity(v, q):
    """Transforms global velocity to body velocity (rotate by the inverse of q)."""
    q = np.asarray(q)
    return _rotate(v, q * np.array([1, -1, -1, -1]))


def body_to_global_velocity(v, q):
    """Transforms body velocity to global velocity."""
    return _rotate(v, np.asarray(q))


def get_foot_step(duty_ratio, cadence, amplitude, phases, time):
    """Desired foot heights from the gait clock; same arguments as the reference."""
    phases = np.asarray(phases, dtype=np.float64)
    t = time * 2 * np.pi * cadence + np.pi
    footphase = 2 * np.pi * phases
    angle = (t + np.pi - footphase) % (2 * np.pi) - np.pi
    if duty_ratio < 1:
        angle = angle * 0.5 / (1 - duty_ratio)
    clipped = np.clip(angle, -np.pi / 2, np.pi / 2)
    value = np.cos(clipped) if duty_ratio < 1 else np.zeros_like(clipped)
    final = np.where(np.abs(value) >= 1e-6, np.abs(value), 0.0)
    return amplitude * final
