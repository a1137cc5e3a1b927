"""ctypes mirror of ``include/drone_vec.h`` (the C-ABI structs and constants).

Pure data definitions: importing this module loads no native code. The
reference-side binding these replace cannot be cited — the reference snapshot
has no binding source (``/root/reference/.gitmodules:1-3`` names an empty
submodule).
"""
import ctypes as C

OBS_DIM = 20
OBS_DIM_MAX = 24
ACT_DIM = 4
GATHER_ID_BYTES = 128
PEER_TOKEN_BYTES = 288
TASK_HOVER = 0
TASK_WAYPOINT = 1
TASK_SWARM = 2
TASK_RACE = 3


def obs_dim(task):
    return OBS_DIM_MAX if task in (TASK_SWARM, TASK_RACE) else OBS_DIM

BUFFERS_HOST = 0
BUFFERS_DEVICE = 1
LAYOUT_AUTO = 0            # DroneConfig.state_layout: by footprint (hover / swarm from ~2^19 envs on drop the target plane)
LAYOUT_TARGET_PLANE = 1
LAYOUT_DERIVED_TARGET = 2

_F = C.c_float


class DroneConfig(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("task", C.c_int32),
        ("buffer_kind", C.c_int32),
        ("device", C.c_int32),
        ("env_offset", C.c_uint32),
        ("horizon", C.c_int32),
        ("substeps", C.c_int32),
        ("compact_done", C.c_int32),
        ("agents_per_env", C.c_int32),
        ("dt", _F),
        ("mass", _F), ("arm", _F), ("ixx", _F), ("iyy", _F), ("izz", _F),
        ("k_thrust", _F), ("k_torque", _F), ("k_drag", _F), ("k_ang_damp", _F), ("gravity", _F),
        ("max_rpm", _F), ("motor_tau", _F), ("max_vel", _F), ("max_omega", _F),
        ("bound", _F), ("spawn_extent", _F), ("target_extent", _F), ("tilt_init", _F),
        ("hover_radius", _F), ("waypoint_radius", _F),
        ("wind_theta", _F), ("wind_sigma", _F), ("wind_max", _F),
        ("c_omega", _F), ("c_action", _F), ("crash_penalty", _F), ("progress_scale", _F), ("waypoint_bonus", _F),
        ("collision_radius", _F), ("proximity_radius", _F), ("c_proximity", _F),
        ("gate_radius", _F),
        ("host_pages_exclusive", C.c_int32),
        ("state_layout", C.c_int32),
    ]

    def as_dict(self):
        return {name: getattr(self, name) for name, _ in self._fields_}


class DroneLog(C.Structure):
    _fields_ = [("perf", _F), ("score", _F), ("episode_return", _F), ("episode_length", _F), ("oob", _F), ("n", _F)]

    def as_dict(self):
        return {name: getattr(self, name) for name, _ in self._fields_}


class DroneStateRow(C.Structure):
    _fields_ = [
        ("pos", _F * 3), ("vel", _F * 3), ("quat", _F * 4), ("omega", _F * 3), ("rpm", _F * 4),
        ("target", _F * 3), ("wind", _F * 3),
        ("ep_return", _F),
        ("tick", C.c_uint32), ("episode", C.c_uint32), ("score_count", C.c_uint32),
        ("perf_sum", _F), ("score_sum", _F), ("ret_sum", _F), ("len_sum", _F), ("n_sum", _F), ("oob_sum", _F),
    ]


# numpy structured dtype with the same layout as DroneStateRow (33 words).
def state_row_dtype():
    import numpy as np

    return np.dtype([
        ("pos", "<f4", 3), ("vel", "<f4", 3), ("quat", "<f4", 4), ("omega", "<f4", 3), ("rpm", "<f4", 4),
        ("target", "<f4", 3), ("wind", "<f4", 3),
        ("ep_return", "<f4"),
        ("tick", "<u4"), ("episode", "<u4"), ("score_count", "<u4"),
        ("perf_sum", "<f4"), ("score_sum", "<f4"), ("ret_sum", "<f4"), ("len_sum", "<f4"), ("n_sum", "<f4"), ("oob_sum", "<f4"),
    ])


# Every symbol include/drone_vec.h declares: name -> (restype, argtypes).
_P = C.c_void_p
SYMBOLS = {
    "drone_config_default": (None, [C.POINTER(DroneConfig), C.c_int]),
    "drone_obs_dim": (C.c_int, [C.c_int]),
    "drone_vec_host_transport": (C.c_int, [_P]),
    "drone_vec_variant": (C.c_char_p, [_P]),
    "drone_vec_bytes_per_env_step": (C.c_int, [_P]),
    "drone_device_count": (C.c_int, []),
    "drone_vec_init": (_P, [_P, _P, _P, _P, _P, C.c_int, C.c_uint64, C.POINTER(DroneConfig)]),
    "drone_vec_reset": (None, [_P, C.c_uint64]),
    "drone_vec_step": (None, [_P]),
    "drone_vec_step_send": (None, [_P]),
    "drone_vec_step_recv": (None, [_P]),
    "drone_vec_rollout": (None, [_P, C.c_int]),
    "drone_vec_step_many": (None, [_P, C.c_int, _P, _P, _P, _P, _P]),
    "drone_vec_step_repeat": (None, [_P, C.c_int, _P, _P, _P, _P, _P]),
    "drone_vec_host_pin": (C.c_int, [_P, _P, C.c_size_t, C.c_int]),
    "drone_vec_host_unpin": (C.c_int, [_P, _P]),
    "drone_vec_log": (None, [_P, C.POINTER(DroneLog)]),
    "drone_vec_close": (None, [_P]),
    "drone_vec_set_stream": (C.c_int, [_P, _P]),
    "drone_vec_sync": (C.c_int, [_P]),
    "drone_vec_bind_actions": (C.c_int, [_P, _P]),
    "drone_vec_bind_outputs": (C.c_int, [_P, _P, _P, _P, _P]),
    "drone_vec_fill_random_actions": (C.c_int, [_P, _P, C.c_uint32]),
    "drone_vec_gstep": (C.c_uint32, [_P]),
    "drone_vec_num_envs": (C.c_int, [_P]),
    "drone_vec_buffers": (C.c_int, [_P, _P, _P, _P, _P, _P]),
    "drone_vec_device": (C.c_int, [_P]),
    "drone_vec_set_gstep": (C.c_int, [_P, C.c_uint32]),
    "drone_vec_enable_graph_capture": (C.c_int, [_P, C.c_int]),
    "drone_vec_status": (C.c_int, [_P]),
    "drone_vec_status_message": (C.c_char_p, [_P]),
    "drone_vec_clear_status": (None, [_P]),
    "drone_gather_unique_id": (C.c_int, [_P]),
    "drone_vec_gather_init": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, _P, _P, _P, _P]),
    "drone_vec_gather_init_root": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, C.c_int, _P, _P, _P, _P]),
    "drone_vec_gather_peer_export": (C.c_int, [_P, _P, _P, _P, _P, _P]),
    "drone_vec_gather_init_peer": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, _P, C.c_int]),
    "drone_vec_gather": (C.c_int, [_P]),
    "drone_vec_gather_close": (None, [_P]),
    "drone_vec_get_state": (C.c_int, [_P, _P, C.c_int, C.c_int]),
    "drone_vec_set_state": (C.c_int, [_P, _P, C.c_int, C.c_int]),
    "drone_vec_done_list": (C.c_int, [_P, _P, C.c_int]),
    "drone_vec_done_list_at": (C.c_int, [_P, C.c_int, _P, C.c_int]),
    "drone_device_malloc": (_P, [C.c_int, C.c_size_t]),
    "drone_device_free": (None, [C.c_int, _P]),
    "drone_vec_copy_to_host": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "drone_vec_timer_start": (C.c_int, [_P]),
    "drone_vec_timer_stop": (C.c_int, [_P, C.POINTER(C.c_float)]),
    "drone_last_error": (C.c_char_p, []),
}


def bind(lib, prefix_from="drone_", prefix_to="drone_", names=None):
    """Attach restype/argtypes for the ABI symbols on a loaded CDLL.

    ``prefix_to`` lets the test-only oracle library (same signatures under the
    ``oracle_`` prefix) reuse this table.
    """
    out = {}
    for name, (res, args) in SYMBOLS.items():
        if names is not None and name not in names:
            continue
        sym = prefix_to + name[len(prefix_from):]
        fn = getattr(lib, sym)
        fn.restype = res
        fn.argtypes = args
        out[name] = fn
    return out
