"""Path / config helpers with the reference's semantics (dial_mpc/utils/io_utils.py:5-24)."""
import os

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def get_model_path(robot_name, model_name):
    return os.path.join(_PKG, "models", robot_name, model_name)


def get_example_path(example_name):
    return os.path.join(_PKG, "examples", example_name)


def load_dataclass_from_dict(dataclass, data_dict, convert_list_to_array=False):
    """Keep only the keys that are fields of ``dataclass`` (silently ignoring the rest); with
    ``convert_list_to_array`` lists become float64 NumPy arrays (the reference makes jnp arrays)."""
    keys = dataclass.__dataclass_fields__.keys() & data_dict.keys()
    kwargs = {key: data_dict[key] for key in keys}
    if convert_list_to_array:
        import numpy as np

        for key, value in kwargs.items():
            if isinstance(value, list):
                kwargs[key] = np.array(value, dtype=np.float64)
    return dataclass(**kwargs)
