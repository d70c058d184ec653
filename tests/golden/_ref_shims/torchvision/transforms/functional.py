import numpy as np
import torch


def to_tensor(pil):
    a = np.asarray(pil, dtype=np.uint8)
    if a.ndim == 2:
        a = a[:, :, None]
    t = torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1)))
    return t.to(torch.float32).div(255)


def normalize(t, mean, std):
    m = torch.tensor(mean, dtype=t.dtype).view(-1, 1, 1)
    s = torch.tensor(std, dtype=t.dtype).view(-1, 1, 1)
    return (t - m) / s
