"""Device-memory plumbing: torch tensors are used only to own HBM buffers and pick streams."""
import numpy as np
import torch

_NP2T = {np.dtype(np.float64): torch.float64, np.dtype(np.float32): torch.float32,
         np.dtype(np.int32): torch.int32, np.dtype(np.int64): torch.int64,
         np.dtype(np.uint8): torch.uint8, np.dtype(np.bool_): torch.bool}


def device():
    if not torch.cuda.is_available():
        raise RuntimeError("pose_estimation_amd needs a HIP device (torch.cuda.is_available() is False); "
                           "there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def stream_ptr():
    return torch.cuda.current_stream().cuda_stream


def is_tensor(x):
    return isinstance(x, torch.Tensor)


def to_dev(x, dtype):
    """numpy array or torch tensor -> contiguous device tensor of `dtype` (no copy when possible)."""
    if is_tensor(x):
        t = x.to(device=device(), dtype=dtype)
    else:
        a = np.ascontiguousarray(x)
        if a.dtype == np.bool_ and dtype == torch.uint8:
            a = a.view(np.uint8)
        t = torch.from_numpy(a).to(device=device(), dtype=dtype)
    return t.contiguous()


def rows(x, cols, dtype):
    """Small parameter table (crop boxes, ...) given as list, numpy array or tensor on any device -> contiguous
    (n, cols) device tensor of `dtype`."""
    if is_tensor(x):
        return to_dev(x.reshape(-1, cols), dtype)
    return to_dev(np.asarray(x).reshape(-1, cols), dtype)


def empty(shape, dtype):
    return torch.empty(shape, dtype=dtype, device=device())


def ptr(t):
    return None if t is None else t.data_ptr()


def back(t, like_tensor):
    """Return a result in the caller's world: torch tensor stays on device, otherwise numpy."""
    return t if like_tensor else t.cpu().numpy()
