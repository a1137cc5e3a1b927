"""The env's device buffers as objects of the DLPack protocol (``__dlpack__`` / ``__dlpack_device__``), over the
compiled binding's ``vec_dlpack`` capsules — what ``torch.from_dlpack``, ``cupy.from_dlpack``, ``jax.dlpack.from_dlpack``
and ``numpy``-style consumers of the array-API protocol take (newer consumers no longer accept bare capsules).

    from drone_amd import drone_binding as binding, dlpack
    env = binding.vec_init(None, None, None, None, None, 65536, 0)        # library-owned HBM buffers
    obs = torch.from_dlpack(dlpack.buffer(env, "observations"))           # zero-copy, [N][O] float32 on the env's GPU

No torch import here; nothing is copied. SURVEY.md §8 f3.
"""
K_DL_ROCM = 10  # DLDeviceType::kDLROCM

NAMES = ("observations", "actions", "rewards", "terminals", "truncations")


class Buffer:
    """One of the env's five buffers, exportable any number of times. Each export holds the env alive until the
    consumer's tensor is gone (the binding's managed-tensor deleter drops the reference)."""

    def __init__(self, handle, name):
        if name not in NAMES:
            raise ValueError(f"unknown buffer {name!r}; one of {NAMES}")
        self._handle, self._name = handle, name

    def __dlpack_device__(self):
        from . import drone_binding

        return (K_DL_ROCM, drone_binding.vec_device(self._handle))

    def __dlpack__(self, stream=None, **_newer_protocol_arguments):
        """``stream``: the consumer's stream (protocol: the data must be safe to use on it). The env writes its buffers
        on ITS stream (``vec_set_stream``), so unless the consumer said it does not care (-1) the env's stream is drained
        first — a no-op for the usual arrangement where env and consumer share one stream and nothing is in flight.
        ``max_version`` / ``dl_device`` / ``copy`` of newer protocol versions are accepted and ignored: the capsule is the
        unversioned ``dltensor`` form, which every consumer still takes, in place, on the env's own device."""
        from . import drone_binding

        if stream != -1:
            drone_binding.vec_sync(self._handle)
        return drone_binding.vec_dlpack(self._handle, self._name)


def buffer(handle, name):
    return Buffer(handle, name)


def buffers(handle):
    """(observations, actions, rewards, terminals, truncations) as protocol objects."""
    return tuple(Buffer(handle, n) for n in NAMES)
