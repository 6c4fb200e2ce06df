import numpy as np
import pytest

F = np.float32


def gpu_or_skip():
    from longbow_amd import _lib
    lib = _lib.load()
    if lib.lb_gpu_device_count() <= 0:
        pytest.skip("no MI355X visible")
    return lib


def new_index(dim, metric=0, order=0, device=0):
    from longbow_amd import gpu
    idx = gpu.NewIndexWithConfig(gpu.GPUConfig(DeviceID=device, Dimension=dim, Metric=metric))
    idx.set_order(order)
    return idx


def assert_same(gpu_labels, gpu_dist, oi, od, ctx=""):
    assert np.array_equal(gpu_labels, oi), f"index sets/order differ {ctx}: {np.argwhere(gpu_labels != oi)[:5]}"
    assert np.array_equal(gpu_dist, od), f"distances differ {ctx}: max abs {np.abs(gpu_dist - od).max()}"
